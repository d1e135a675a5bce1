#!/usr/bin/env python3
"""Rewrites the measured-number tables of DESIGN.md (section 6, up to 'How the round got here') and BASELINE.md from the
committed profiles/r01_* files, so that the documents quote what the profiles hold."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
b = json.load(open('profiles/r01_bench.json'))
t = json.load(open('profiles/traffic.json'))
m = json.load(open('profiles/r01_side_measurements.json'))
ph = json.load(open('profiles/r01_phase_cycles.json'))
sq = {}
for ln in open('profiles/r01_sq_counters.csv'):
    p = ln.strip().split(',')
    if len(p) == 3 and p[0].startswith('SQ_'):
        sq[p[0]] = float(p[1])
ks = [l for l in open('profiles/r01_kernel_stats.csv') if 'filter_kernel_dense<10, false' in l][0].split('",')[1].split(',')
calls, avg_ns = int(ks[0]), float(ks[2])
pairs = 1024
rs = b['rot_scale']
cs = m['crowded_scene_4096x10k_default_flags']
P = ph['phases']
c = lambda k: P[k]['cycles']

head = f"""## 6. Measurements (round 1, MI355X, 1 GPU; `profiles/r01_*`)

Headline (`python bench.py`, defaults: 4096 pairs per step, 20 steps, 3 warm-up; `profiles/r01_bench.json`):

| Quantity | Value |
|---|---|
| GMS-filtered image pairs / s, flags (false, false, 6.0), 10k matches per 1080p pair | **{b['value']/1e6:.2f} M pairs/s** ({b['ms_per_step']:.3f} ms per 4096-pair step) |
| kernel `gms::filter_kernel_dense<10,false,1024>` per launch | {b['roofline']['kernel_ms_per_launch']:.3f} ms by HIP events in bench.py; {avg_ns/1e6:.3f} ms average over {calls} calls in `rocprofv3 --kernel-trace --stats` (`r01_kernel_stats.csv`) |
| algorithmic bytes per launch (32·M + 16·K per pair, K = 4970) | {b['roofline']['algorithmic_bytes_per_launch']/1e9:.3f} GB |
| `roofline.achieved` / peak / frac | {b['roofline']['achieved']:.0f} GB/s / 8000 GB/s / **{b['roofline']['frac']:.3f}** |
| PMC traffic per launch (`r01_pmc_traffic.json`; FETCH_SIZE ×2 per the gfx950 calibration, + WRITE_SIZE) | {t['hbm_bytes_per_launch']/1e9:.2f} GB = {t['hbm_bytes_per_launch']/t['algorithmic_bytes_per_launch']:.2f} × algorithmic (reads {t['hbm_read_bytes_per_launch']/1e9:.2f} GB, writes {t['hbm_write_bytes_per_launch']/1e9:.3f} GB = 16·K·P + results) |
| CPU baseline, oracle port on the GPU box's host (EPYC 9575F), 16 threads, one pair per thread | {b['cpu_baseline']['value']:.0f} pairs/s ({b['cpu_baseline']['value_1thread']:.0f} pairs/s on 1 thread) → GPU/CPU(16 thr) = {b['gpu_vs_cpu']:.0f}× |
| parity inside the bench | 256 sampled pairs bit-exact vs the oracle |
| same pairs with flags (true, true, 6.0) — 8 rot × 5 scale × 4 grids (`FeatureMatchUtil.cpp:69`): scales 0..3 on the byte matrix, scale 4 hashed (§4.3) | {rs['value']/1e3:.0f} k pairs/s; CPU 16 threads {rs['cpu_baseline']['value']:.0f} pairs/s → {rs['value']/rs['cpu_baseline']['value']:.0f}× |

`FETCH_SIZE` counts requests on the L2's memory side, Infinity Cache hits included (MI355X_MICROARCH.md), so the traffic
figure is an upper bound on HBM bytes; the calibration kernel (`tools/ubench/fetch_calib.hip`, 2 GiB buffer) shows the
counter reporting exactly half of the bytes for this kernel's read shapes, hence the ×2. Traffic is now *below* the
algorithmic figure: the `DMatch` records are read once (they stay in registers until copy-out; the re-read that cost
655 MB per launch is gone) and most of the two 8-byte keypoint gathers per match are served by L2 (frame A) and by the
LDS copy of frame B.

SQ counters per pair (`r01_sq_counters.csv`, 1024-pair launches): {sq['SQ_INSTS_VALU']/pairs/1e3:.0f} k VALU (the hashed path: 93 k), {sq['SQ_INSTS_SALU']/pairs/1e3:.0f} k SALU,
{sq['SQ_INSTS_LDS']/pairs/1e3:.1f} k LDS wave-instructions ({sq['SQ_INSTS_LDS_ATOMIC']/pairs/1e3:.1f} k of them atomics); a wave is issuing {100*sq['SQ_ACTIVE_INST_ANY']/sq['SQ_WAVE_CYCLES']:.0f} % of its
life, waits at `s_waitcnt`/barriers {100*sq['SQ_WAIT_ANY']/sq['SQ_WAVE_CYCLES']:.0f} %, is stalled at issue {100*sq['SQ_WAIT_INST_ANY']/sq['SQ_WAVE_CYCLES']:.0f} %.
In-kernel phase stamps (diagnostic build, `tools/phase_timing.py`, `r01_phase_cycles.json`; shader cycles per pair seen by
wave 0, {ph['total_cycles']/1e3:.0f} k total — read the shares, wave 0 is the oldest wave and runs ahead of the others inside a
phase): records + frame B arriving {c('bin_qt_loads')/1e3:.0f} k, gathers + code words + histogram {(c('bin_gathers')+c('bin_compute')+c('bin_barrier'))/1e3:.0f} k, clearing what
frame B occupied {c('clear')/1e3:.1f} k; summed over the four grid types: binning {(c('insert_first')+c('insert_barrier'))/1e3:.0f} k, verify {c('verify')/1e3:.0f} k, mark + undo {c('mark')/1e3:.0f} k;
scan {c('out_scan')/1e3:.0f} k, copy-out {c('copy_out')/1e3:.0f} k. A workgroup alone on the chip gets its 240 KB in 8 k cycles (≈30 B/cycle, the CU's
vector-memory issue rate); with every CU streaming, the same loads take 11–17 k (`tools/phase_timing.py --starts`
records each workgroup's start and landing time).

Crowded scenes (`tools/crowded_bench.py`: the same workload with the keypoints squeezed into the central 28 % × 28 % of the
image, ≈277 matches per populated cell): every pair trips the first `nLeft` check and is redone in the kernel's crowded mode —
the same byte matrix (an entry only wraps when ONE (left cell, right cell) pair collects more than 255 matches; every returned
count is checked, and then the hashed path takes over), `nLeft` counted per grid type into 16-bit counters with one more LDS
atomic per match — {cs['crowded']['pairs_per_s']/1e6:.2f} M pairs/s ({cs['uniform']['pairs_per_s']/1e6:.2f} M in the same run for the uniform sequence; 3.88 M before the crowded mode
existed, when such pairs went to the hashed path), bit-exact. With rotation + scale the dense-scales kernel has the same mode:
{cs['crowded']['rot_scale_pairs_per_s']/1e3:.0f} k pairs/s on the crowded sequence ({cs['uniform']['rot_scale_pairs_per_s']/1e3:.0f} k on the uniform one in the same run: fewer populated cells, fewer verify waves).

"""
side = f"""Side measurements (`r01_side_measurements.json`): the one-shot host-pointer call `gms_match` on a 10k-match pair takes
{m['one_shot_10k_rot0_scale0']['gpu_call_ms_incl_pcie']:.2f} ms end to end including the PCIe copies ({m['one_shot_10k_rot0_scale0']['gpu_pairs_per_s']:.0f} pairs/s; the oracle needs {m['one_shot_10k_rot0_scale0']['cpu_oracle_ms_1thread']:.2f} ms on one core) and {m['one_shot_10k_rot1_scale1']['gpu_call_ms_incl_pcie']:.2f} ms with
rotation + scale (oracle {m['one_shot_10k_rot1_scale1']['cpu_oracle_ms_1thread']:.0f} ms) — PCIe-inclusive rates, never the headline `value`. BASELINE config 1
(640×480, 500 matches): {m['config1_500_rot0_scale0']['gpu_call_ms_incl_pcie']:.3f} / {m['config1_500_rot1_scale1']['gpu_call_ms_incl_pcie']:.3f} ms per call without / with rotation + scale (oracle {m['config1_500_rot0_scale0']['cpu_oracle_ms_1thread']:.3f} / {m['config1_500_rot1_scale1']['cpu_oracle_ms_1thread']:.2f} ms;
at this size a call is launch + copy latency). BASELINE config 4 (3840×2160,
50k matches): {m['config4_50k_rot0_scale0']['gpu_call_ms_incl_pcie']:.2f} ms per call with the default flags (band kernels; oracle {m['config4_50k_rot0_scale0']['cpu_oracle_ms_1thread']:.1f} ms), {m['config4_50k_rot1_scale1']['gpu_call_ms_incl_pcie']:.2f} ms with rotation + scale
(tile kernels; oracle {m['config4_50k_rot1_scale1']['cpu_oracle_ms_1thread']:.0f} ms; {m['batch64_50000_rot_scale_band1']['pairs_per_s']/1e3:.0f} k pairs/s with 64 pairs resident, the slab kernel alone {m['batch64_50000_rot_scale_band0']['pairs_per_s']/1e3:.1f} k). Device-resident batches of 256 large pairs, default flags: {m['batch256_50000_default_flags_band1']['pairs_per_s']/1e3:.0f} k pairs/s at 50k matches
({m['batch256_50000_default_flags_band1']['gmatches_per_s']:.1f} G matches/s; slab kernel alone {m['batch256_50000_default_flags_band0']['pairs_per_s']/1e3:.0f} k), {m['batch256_168750_default_flags_band1']['pairs_per_s']/1e3:.0f} k pairs/s at 168 750 matches (slab kernel alone {m['batch256_168750_default_flags_band0']['pairs_per_s']/1e3:.0f} k).

"""
s = open('DESIGN.md').read()
a, z = s.index('## 6. Measurements (round 1'), s.index('**How the round got here**')
s = s[:a] + head + s[z:]
a, z = s.index('Side measurements (`r01_side_measurements.json`)'), s.index('## 8. What comes next')
s = s[:a] + side + s[z:]
open('DESIGN.md', 'w').write(s)

s = open('BASELINE.md').read()
a = s.index("| 3-shape batch: 1080p, 10k matches/pair")
z = s.index("multi-GPU scaling is measured by the driver")
new = f"""| 3-shape batch: 1080p, 10k matches/pair, 4096 pairs resident per step | (0,0,6.0) | {b['cpu_baseline']['value_1thread']:.0f} pairs/s (1 thread), {b['cpu_baseline']['value']:.0f} pairs/s (16 threads, one pair per thread) | {b['value']:.0f} pairs/s, kernel {b['roofline']['kernel_ms_per_launch']:.3f} ms per launch | {b['gpu_vs_cpu']:.0f}× vs 16 threads | 256 sampled pairs bit-exact |
| same pairs | (1,1,6.0) | {rs['cpu_baseline']['value_1thread']:.0f} / {rs['cpu_baseline']['value']:.0f} pairs/s | {rs['value']:.0f} pairs/s | {rs['value']/rs['cpu_baseline']['value']:.0f}× | 32 sampled pairs bit-exact |

Roofline (algorithmic 32·M + 16·K bytes per pair): {b['roofline']['achieved']:.0f} GB/s of 8000 GB/s = {b['roofline']['frac']:.3f}; PMC-measured HBM-side traffic {t['hbm_bytes_per_launch']/1e9:.2f} GB
per launch against {t['algorithmic_bytes_per_launch']/1e9:.2f} GB algorithmic. Targets of BASELINE.json: ≥ 10× CPU — met ({b['gpu_vs_cpu']:.0f}× vs 16 host threads); ≥ 40 % of HBM roofline —
met at {100*b['roofline']['frac']:.0f} % with the byte-matrix kernel for the default flags (the hashed path, still used for scale hypothesis 4 and for crowded cells, sits at 19 % on the same pairs; see DESIGN.md §4, §6);
"""
s = s[:a] + new + s[z:]
open('BASELINE.md', 'w').write(s)
r = open('README.md').read()
import re
r = re.sub(r"Measured on one MI355X \(`profiles/r01_\*`, DESIGN.md §6\): [0-9.]+ M filtered pairs/s at 10k matches per pair with the\nreference's default flags \([0-9.]+ of the HBM roofline, [0-9]+× a 16-thread host run of the oracle\), bit-exact\.",
           f"Measured on one MI355X (`profiles/r01_*`, DESIGN.md §6): {b['value']/1e6:.2f} M filtered pairs/s at 10k matches per pair with the\nreference's default flags ({b['roofline']['frac']:.2f} of the HBM roofline, {b['gpu_vs_cpu']:.0f}× a 16-thread host run of the oracle), bit-exact.", r)
open('README.md', 'w').write(r)

# INTEGRATION.md section 4
s = open('INTEGRATION.md').read()
a = s.index("## 4. What to expect")
tab = f"""## 4. What to expect (one MI355X, `profiles/r01_*`)

| Call pattern | Flags | Measured |
|---|---|---|
| `gms_match`, host pointers, 640×480 / 500 matches (BASELINE config 1) | default / rot+scale | {m['config1_500_rot0_scale0']['gpu_call_ms_incl_pcie']:.2f} / {m['config1_500_rot1_scale1']['gpu_call_ms_incl_pcie']:.2f} ms per call (copies included) |
| `gms_match`, 1080p / 10k matches (config 2) | default / rot+scale | {m['one_shot_10k_rot0_scale0']['gpu_call_ms_incl_pcie']:.2f} / {m['one_shot_10k_rot1_scale1']['gpu_call_ms_incl_pcie']:.2f} ms per call |
| `gms_match`, 4K / 50k matches (config 4) | default / rot+scale | {m['config4_50k_rot0_scale0']['gpu_call_ms_incl_pcie']:.2f} / {m['config4_50k_rot1_scale1']['gpu_call_ms_incl_pcie']:.2f} ms per call |
| `gms_filter_device`, 4096 pairs × 10k matches resident (config 3 shape) | default / rot+scale | {b['value']/1e6:.2f} M / {b['rot_scale']['value']/1e6:.2f} M pairs per second |
| the same, crowded scene (every populated cell above 255 matches) | default / rot+scale | {cs['crowded']['pairs_per_s']/1e6:.2f} M / {cs['crowded']['rot_scale_pairs_per_s']/1e6:.2f} M pairs per second |
| `gms_filter_device`, 256 pairs × 50k matches resident | default | {m['batch256_50000_default_flags_band1']['pairs_per_s']/1e3:.0f} k pairs per second |
| `gms_filter_device`, 64 pairs × 50k matches resident | rot+scale | {m['batch64_50000_rot_scale_band1']['pairs_per_s']/1e3:.0f} k pairs per second |

Environment knobs exist for diagnostics only (`GMS_DENSE`, `GMS_BAND`, `GMS_STAGGER_US`, …; README.md); none is needed in production.
"""
open('INTEGRATION.md', 'w').write(s[:a] + tab)
print("docs updated:", b['value'], b['roofline']['frac'])
