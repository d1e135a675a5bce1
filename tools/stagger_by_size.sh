#!/bin/bash
# Diagnostic: the headline bench at smaller pairs (--features = matches per pair) with the first-round stagger as the library sets it and off.
for f in 3000 4000 6000 8000; do for s in default 0; do
if [ $s = default ]; then unset GMS_STAGGER_US; else export GMS_STAGGER_US=0; fi
python3 bench.py --no-cpu --no-extra --features $f 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('features $f stagger $s', round(d['roofline']['kernel_ms_per_launch'],4), round(d['value']), d['parity']['bit_exact'], d['variant']['first_round_stagger_us'])"
done; done
