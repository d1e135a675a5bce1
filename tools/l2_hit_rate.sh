#!/bin/bash
# Diagnostic: L2 hits / misses of the headline launches with and without the touch-ahead (GMS_PREFETCH=3 / -1); counters only.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/l2hit; rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp
for pf in 3 -1; do
  GMS_PREFETCH=$pf rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$OUT/pf$pf" -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu --no-extra --steps 5 --warmup 2 > "$OUT/pf$pf.json" 2> "$OUT/pf$pf.log" || tail -3 "$OUT/pf$pf.log"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
for pf in ("3", "-1"):
    agg = collections.defaultdict(list)
    for fn in glob.glob(sys.argv[1] + f"/pf{pf}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if "filter_kernel_dense<" in r.get("Kernel_Name", ""):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("GMS_PREFETCH", pf, {k: round(sum(v) / len(v)) for k, v in agg.items()})
PY
