#!/bin/bash
# Diagnostic: the headline bench at several first-round staggers (GMS_STAGGER_US, read once per process). bash tools/stagger_sweep.sh 26 22 24 ...
for us in "$@"; do
  GMS_STAGGER_US=$us python3 bench.py --no-cpu --no-extra > /tmp/stagger_$us.json 2>/dev/null
  python3 - "$us" <<'PY'
import json, sys
d = json.loads(open(f"/tmp/stagger_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("stagger_us", sys.argv[1], "kernel_ms", round(d["roofline"]["kernel_ms_per_launch"], 4), "pairs_per_s", round(d["value"]))
PY
done
