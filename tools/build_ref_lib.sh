#!/bin/bash
# Diagnostic: build the library with gms_kernels.hip taken from a git revision (for tools/ab_bench.py), the other objects from the current build.
# tools/build_ref_lib.sh <rev> <name>  ->  sfm-gms_amd/csrc/libgms_hip_<name>.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); C=$ROOT/sfm-gms_amd/csrc; B=$C/build; T=$(mktemp -d)
git -C "$ROOT" show "$1:sfm-gms_amd/csrc/gms_kernels.hip" > "$C/.ref_tmp.hip"
mkdir -p "$T/inc"; for h in gms_kernels.h gms_device_common.h; do git -C "$ROOT" show "$1:sfm-gms_amd/csrc/$h" > "$T/inc/$h"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I"$T/inc" -I"$ROOT/include" -I"$C" -pthread -mllvm -disable-machine-licm -c -x hip "$C/.ref_tmp.hip" -o "$T/k.o"
rm -f "$C/.ref_tmp.hip"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -pthread -o "$C/libgms_hip_$2.so" "$T/k.o" $B/gms_kernel_big.hip.o $B/gms_kernel_band.hip.o $B/gms_kernel_stream.hip.o $B/bf_kernels.hip.o $B/consumer_kernels.hip.o $B/twoview_kernels.hip.o $B/detect_kernels.hip.o $B/gms_capi.cpp.o $B/gms_io.cpp.o
rm -rf "$T"; echo "$C/libgms_hip_$2.so"
