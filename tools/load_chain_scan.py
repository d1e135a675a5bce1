#!/usr/bin/env python3
"""Diagnostic (CPU only: hipcc cross-compiles): for every kernel of the filter sources, how many global loads are waited for at once
-- a `global_load` with `s_waitcnt vmcnt(0)` within the next two instructions. A value loaded under a condition (`c ? p[i] : d`), a
v_readfirstlane straight behind its load, a load that only a conditional store uses: each compiles to a round trip of its own, and a
run of them is a chain of round trips where the source reads like loads in flight together.
python tools/load_chain_scan.py [csrc directory]   (default: sfm-gms_amd/csrc)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "sfm-gms_amd", "csrc")
FILES = ["gms_kernels.hip", "gms_kernel_stream.hip", "gms_kernel_band.hip", "gms_kernel_big.hip", "bf_kernels.hip", "consumer_kernels.hip",
         "twoview_kernels.hip", "detect_kernels.hip"]


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return {n: (d.replace("(anonymous namespace)::", "").split("(")[0] if d else n) for n, d in zip(names, out)}
    except OSError:
        return {n: n for n in names}


def main():
    print(f"# {SRC}")
    print("# kernel: global loads, of which waited for at once")
    for f in FILES:
        path = os.path.join(SRC, f)
        if not os.path.exists(path):
            continue
        with tempfile.TemporaryDirectory() as td:
            asm = os.path.join(td, "k.s")
            flags = ["-mllvm", "-amdgpu-mfma-vgpr-form"] if f == "bf_kernels.hip" else ["-mllvm", "-disable-machine-licm"] if f.startswith("gms_kernel") else []
            r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", f"-I{ROOT}/include", f"-I{SRC}",
                                *flags, "--cuda-device-only", "-S", path, "-o", asm], capture_output=True, text=True)
            if r.returncode != 0:
                print(f"{f}: does not compile here", r.stderr[-300:])
                continue
            cur, res, k, last = None, {}, 0, -99
            for line in open(asm):
                m = re.match(r"^(_Z\S+):", line)
                if m:
                    cur, k, last = m.group(1), 0, -99
                    res[cur] = [0, 0]
                    continue
                t = line.strip()
                if cur is None or not t or t.startswith((".", ";")):
                    continue
                k += 1
                if t.startswith(("global_load", "buffer_load")):
                    res[cur][0] += 1
                    last = k
                if t.startswith("s_waitcnt vmcnt(0)") and k - last <= 2:
                    res[cur][1] += 1
            names = demangle(list(res))
            for kn, (nl, nw) in res.items():
                if nl:
                    print(f"{f}: {names[kn]}: {nl}, {nw}")


if __name__ == "__main__":
    main()
