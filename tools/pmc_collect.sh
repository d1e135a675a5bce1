#!/bin/bash
# Collects SQ counters for gms::filter_kernel in separate rocprofv3 passes (8 SQ slots each); counters only,
# no tracing domains. Run on the GPU box from the repo root; results under gpurun_out/pmc/.
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc${PMC_TAG}
rm -rf "$OUT"; mkdir -p "$OUT"
ARGS="$GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-extra --pairs 1024 $PMC_EXTRA_ARGS"
cd /tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU"
P3="SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_ATOMIC_RETURN SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM"
i=1
for P in "$P1" "$P2" "$P3"; do
  rocprofv3 --pmc $P --output-format csv -d "$OUT/p$i" -- python3 $ARGS > "$OUT/p$i.log" 2>&1 || { tail -5 "$OUT/p$i.log"; exit 1; }
  i=$((i+1))
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "filter_kernel" in row.get("Kernel_Name", ""):
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
open(out + "/../pmc_summary" + __import__("os").environ.get("PMC_TAG", "") + ".csv", "w").write("counter,mean_per_dispatch,n\n" + "".join(f"{k},{sum(agg[k])/len(agg[k]):.0f},{len(agg[k])}\n" for k in sorted(agg)))
print("counter,mean_per_dispatch,n")
for k in sorted(agg):
    v = agg[k]
    print(f"{k},{sum(v)/len(v):.0f},{len(v)}")
PY
