#!/bin/bash
# SQ counters + kernel durations of the brute-force matcher kernels (tools/bf_bench.py KIND 256), one rocprofv3 pass: counters and the
# kernel trace only. Run on the GPU box from the repo root: bash tools/bf_pmc.sh orb|sift ; results under gpurun_out/bf_pmc_KIND/.
set -e
export TMPDIR=/tmp
KIND=${1:-orb}
OUT=$PWD/gpurun_out/bf_pmc_$KIND
rm -rf "$OUT"; mkdir -p "$OUT"
SCRIPT=$GRAFT_REPO_ROOT/tools/bf_bench.py
cd /tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d "$OUT/p1" -- python3 $SCRIPT $KIND 256 > "$OUT/p1.log" 2>&1 || { tail -5 "$OUT/p1.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
dur = []
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "bf_mfma_kernel" in row.get("Kernel_Name", ""):
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
for f in glob.glob(out + "/p*/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "bf_mfma_kernel" in row.get("Kernel_Name", ""):
            dur.append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
lines = ["counter,mean_per_dispatch,n"] + [f"{k},{sum(v)/len(v):.0f},{len(v)}" for k, v in sorted(agg.items())]
if dur:
    d = sum(dur) / len(dur)
    lines.append(f"kernel_duration_ns,{d:.0f},{len(dur)}")
    if "GRBM_GUI_ACTIVE" in agg:
        lines.append(f"shader_clock_GHz,{sum(agg['GRBM_GUI_ACTIVE'])/len(agg['GRBM_GUI_ACTIVE'])/d:.3f},1")
open(out + "/summary.csv", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
