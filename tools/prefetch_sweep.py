#!/usr/bin/env python3
"""Diagnostic: the headline bench under several GMS_PREFETCH settings (grid type before which a workgroup touches the records of the
pair its CU's next workgroup will take [, how many pairs ahead]) in one session on one device. python tools/prefetch_sweep.py [settings...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
settings = sys.argv[1:] or ["-1", "3", "2", "1", "0", "3,128", "3,512", "-1", "3"]
for s in settings:
    env = dict(os.environ, GMS_PREFETCH=s)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu", "--no-extra"], capture_output=True, text=True, env=env)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print("GMS_PREFETCH=%-6s" % s, round(d["roofline"]["kernel_ms_per_launch"], 4), "ms/launch", round(d["value"]), "pairs/s", d["parity"]["bit_exact"], flush=True)
    except Exception:
        print("GMS_PREFETCH=%s failed" % s, r.stderr[-300:], flush=True)
