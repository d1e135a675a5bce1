#!/bin/bash
# Round profile of the bench command (run on the GPU box from the repo root):
#   1. rocprofv3 --kernel-trace --stats  -> per-kernel time of `python3 bench.py --no-cpu --no-extra` (only the headline launches:
#      the average duration of filter_kernel_dense<10,false,1024> is directly comparable with bench.py's HIP-event figure)
#   2. the same of the full default `python3 bench.py` (rot+scale side measurement, matcher kernels, CPU baseline)
#   3. rocprofv3 --pmc FETCH_SIZE        -> HBM-side read traffic   (own pass: TCC slots)
#   4. rocprofv3 --pmc WRITE_SIZE        -> HBM-side write traffic  (own pass)
# Counter passes use --pmc only (no tracing domains). Summaries land in gpurun_out/prof/; tools/make_profiles.py copies what is to be
# judged into profiles/.
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu --no-extra > "$OUT/bench_under_trace.json" 2> "$OUT/trace.log" || { tail -5 "$OUT/trace.log"; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_full" -- python3 "$GRAFT_REPO_ROOT/bench.py" > "$OUT/bench_full_under_trace.json" 2> "$OUT/trace_full.log" || { tail -5 "$OUT/trace_full.log"; exit 1; }
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$C" -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu --no-extra --steps 5 --warmup 2 > "$OUT/bench_under_$C.json" 2> "$OUT/pmc_$C.log" || { tail -5 "$OUT/pmc_$C.log"; exit 1; }
done
python3 - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
for tag in ("trace", "trace_full"):
    stats = glob.glob(out + f"/{tag}/**/*kernel_stats.csv", recursive=True)
    rows = list(csv.DictReader(open(stats[0]))) if stats else []
    with open(out + f"/kernel_stats_{tag}.csv", "w") as f:
        if rows:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
    for r in rows:
        if "gms::" in r["Name"]:
            print(tag, r["Name"].split("(")[0][-60:], r["Calls"], r["AverageNs"])
summary = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for fn in glob.glob(out + f"/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if "filter_kernel_dense<" in r.get("Kernel_Name", "") and r["Counter_Name"] == c:
                vals.append(float(r["Counter_Value"]))
    summary[c] = {"mean_per_launch": sum(vals) / len(vals) if vals else None, "launches": len(vals)}
json.dump(summary, open(out + "/pmc_traffic_raw.json", "w"), indent=1)
print(json.dumps(summary))
PY
