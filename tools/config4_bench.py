#!/usr/bin/env python3
"""Throughput of gms_filter_device on large pairs (BASELINE config 4: 3840 x 2160, 50k matches per pair) for every flag combination,
on the streamed byte-matrix kernels (default) and, in a child process with GMS_STREAM=0, on the band / tile kernels."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
CASES = [(50000, 64, 1, 1), (50000, 256, 1, 1), (50000, 256, 0, 0), (50000, 64, 0, 0), (50000, 64, 1, 0), (50000, 256, 1, 0), (50000, 64, 0, 1), (20000, 256, 1, 1), (20000, 256, 0, 0)]


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        import measure_misc as mm
        ctx = mm.pkg.GmsContext(0)
        out = {}
        for m, n, rot, scale in CASES:
            out[f"batch{n}_{m}_rot{rot}_scale{scale}"] = mm.device_batch(ctx, m, n, bool(rot), bool(scale), reps=3)
        print(json.dumps(out))
        return
    res = {}
    for tag, env in (("stream", {}), ("band_tile", {"GMS_STREAM": "0"})):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], capture_output=True, text=True, env=dict(os.environ, **env))
        res[tag] = json.loads(r.stdout.strip().splitlines()[-1])
    for k in res["stream"]:
        print(k, "stream %.0f" % res["stream"][k]["pairs_per_s"], "band/tile %.0f" % res["band_tile"][k]["pairs_per_s"])
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "config4_bench.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
