#!/usr/bin/env python3
"""Diagnostic A/B of matchGMS(true, true) at 10k matches per pair: library builds x environment settings in one session on one
device. Every leg is a fresh process (the switches are read once per process):

    python tools/ab_scales.py [--pairs 2048] [--zoom] leg [leg ...]       leg = name:lib[:VAR=val[,VAR=val...]]
    e.g.  base:libgms_hip_base.so  new:libgms_hip.so  new_unsorted:libgms_hip.so:GMS_SCALES_SORTED=0

Per leg: pairs/s with the scale probe forced on and off (and, with --zoom, bench.py's zooming sequence), parity of every 64th pair
against the oracle. Two rounds, so that drift of the box shows."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import argparse, importlib, json, sys, time
import numpy as np
sys.path.insert(0, {root!r})
capi = importlib.import_module('sfm-gms_amd.capi'); capi.library_path = lambda: {lib!r}
import torch, bench
pkg = importlib.import_module('sfm-gms_amd')
dev = torch.device('cuda', 0)
ctx = pkg.GmsContext(0)
stream = torch.cuda.Stream(device=dev)
N = {pairs}
args = argparse.Namespace(pairs=N, frames=200, features={features}, inlier_frac=0.5, warmup=2, steps=8, max_resident=10)
wl = bench.Workload(args, 0, 1, dev, pkg, ctx)
ctx.set_stream(stream.cuda_stream)
out = {{}}
for name, val in (('on', 1), ('off', 0)):
    ctx.set_option(2, val)
    w, k = bench.timed_steps(ctx, wl, stream, 8, 2, True, True, None)
    c, b = bench.check_parity(wl, pkg, range(len(wl.chunks)), True, True, sample={{i: list(range(0, N, 64)) for i in range(len(wl.chunks))}})
    out[name] = dict(pairs_per_s=N * 8 / w, ms=k, checked=c, bad=b)
ctx.set_option(2, -1)
if {zoom}:
    z = bench.zoom_leg(ctx, pkg, stream, dev)
    out['zoom'] = dict(auto=z['auto']['pairs_per_s'], off=z['off']['pairs_per_s'], on=z['on']['pairs_per_s'], bad=z['parity']['mismatches'], auto_mask=z['auto']['probe_mask'])
print(json.dumps(out))
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=2048)
    ap.add_argument("--features", type=int, default=10000)
    ap.add_argument("--zoom", action="store_true")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("legs", nargs="+")
    a = ap.parse_args()
    for rnd in range(a.rounds):
        for leg in a.legs:
            parts = leg.split(":")
            name, lib = parts[0], parts[1]
            env = dict(os.environ)
            if len(parts) > 2:
                for kv in parts[2].split(","):
                    k, v = kv.split("=")
                    env[k] = v
            code = CHILD.format(root=ROOT, lib=os.path.join(ROOT, "sfm-gms_amd", "csrc", lib), pairs=a.pairs, features=a.features, zoom=bool(a.zoom))
            r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
            try:
                d = json.loads(r.stdout.strip().splitlines()[-1])
                line = {k: (round(v["pairs_per_s"]), v["bad"]) for k, v in d.items() if k != "zoom"}
                if "zoom" in d:
                    line["zoom"] = {k: round(v) for k, v in d["zoom"].items()}
                print(f"round {rnd} {name}: {line}", flush=True)
            except Exception:
                print(f"round {rnd} {name}: failed", r.stderr[-600:], flush=True)


if __name__ == "__main__":
    main()
