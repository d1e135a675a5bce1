#!/bin/bash
# Per-kernel times of the large-pair paths (BASELINE config 4: 50k matches per 4K pair) under rocprofv3 --kernel-trace --stats:
#   gpurun_out/prof4/<tag>/ ...   tags: rot_scale_64 (FeatureMatchUtil.cpp:69 flags, 64 pairs), default_256 (default flags, 256 pairs)
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof4
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/rot_scale_64" -- python3 "$GRAFT_REPO_ROOT/tools/measure_misc.py" batch 50000 64 1 1 > "$OUT/rot_scale_64.json" 2> "$OUT/rot_scale_64.log" || { tail -5 "$OUT/rot_scale_64.log"; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/default_256" -- python3 "$GRAFT_REPO_ROOT/tools/measure_misc.py" batch 50000 256 0 0 > "$OUT/default_256.json" 2> "$OUT/default_256.log" || { tail -5 "$OUT/default_256.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
for tag in ("rot_scale_64", "default_256"):
    stats = glob.glob(out + f"/{tag}/**/*kernel_stats.csv", recursive=True)
    rows = list(csv.DictReader(open(stats[0]))) if stats else []
    with open(out + f"/kernel_stats_{tag}.csv", "w") as f:
        if rows:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
    print(tag, open(out + f"/{tag}.json").read().strip())
    for r in rows[:14]:
        print("  ", r["Name"].split("(")[0][-70:], r["Calls"], r["AverageNs"], r["Percentage"])
PY
