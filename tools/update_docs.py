#!/usr/bin/env python3
"""Rewrites the measured-number passages of DESIGN.md (section 6) and BASELINE.md from the committed profiles/r01_* files."""
import json
import os
import re

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
ks = [l for l in open('profiles/r01_kernel_stats.csv') if 'filter_kernel<10, false' in l][0].split('",')[1].split(',')
calls, avg_ns = int(ks[0]), float(ks[2])
P = ph['phases']
c = lambda k: P[k]['cycles']
bin_c = c('bin_qt_loads') + c('bin_gathers') + c('bin_compute') + c('bin_barrier')
ins_c = c('insert_first') + c('insert_leftover') + c('insert_barrier')
pairs = 1024
rs = b['rot_scale']

sec6 = f"""## 6. Measurements (round 1, MI355X, 1 GPU; `profiles/r01_*`)

Headline (`python bench.py`, defaults: 4096 pairs per step, 20 steps, 3 warm-up; `profiles/r01_bench.json`):

| Quantity | Value |
|---|---|
| GMS-filtered image pairs / s, flags (false, false, 6.0), 10k matches per 1080p pair | **{b['value']/1e6:.2f} M pairs/s** ({b['ms_per_step']:.3f} ms per 4096-pair step) |
| kernel `gms::filter_kernel<10,false,1024>` per launch | {b['roofline']['kernel_ms_per_launch']:.3f} ms by HIP events in bench.py; {avg_ns/1e6:.3f} ms average over {calls} calls in `rocprofv3 --kernel-trace --stats` (`r01_kernel_stats.csv`) |
| algorithmic bytes per launch (32·M + 16·K per pair, K = 4970) | {b['roofline']['algorithmic_bytes_per_launch']/1e9:.3f} GB |
| `roofline.achieved` / peak / frac | {b['roofline']['achieved']:.0f} GB/s / 8000 GB/s / **{b['roofline']['frac']:.3f}** |
| PMC traffic per launch (`r01_pmc_traffic.json`; FETCH_SIZE ×2 per the gfx950 calibration, + WRITE_SIZE) | {t['hbm_bytes_per_launch']/1e9:.2f} GB = {t['hbm_bytes_per_launch']/t['algorithmic_bytes_per_launch']:.2f} × algorithmic (reads {t['hbm_read_bytes_per_launch']/1e9:.2f} GB, writes {t['hbm_write_bytes_per_launch']/1e9:.3f} GB = exactly 16·K·P) |
| CPU baseline, oracle port on the GPU box's host (EPYC 9575F), 16 threads, one pair per thread | {b['cpu_baseline']['value']:.0f} pairs/s ({b['cpu_baseline']['value_1thread']:.0f} pairs/s on 1 thread) → GPU/CPU(16 thr) = {b['gpu_vs_cpu']:.0f}× |
| parity inside the bench | 256 sampled pairs bit-exact vs the oracle |
| same pairs with flags (true, true, 6.0) — 8 rot × 5 scale × 4 grids (`FeatureMatchUtil.cpp:69`) | {rs['value']/1e3:.0f} k pairs/s; CPU 16 threads {rs['cpu_baseline']['value']:.0f} pairs/s → {rs['value']/rs['cpu_baseline']['value']:.0f}× |

`FETCH_SIZE` counts requests on the L2's memory side, Infinity Cache hits included (MI355X_MICROARCH.md), so the traffic
figure is an upper bound on HBM bytes; the calibration kernel (`tools/ubench/fetch_calib.hip`, 2 GiB buffer) shows the
counter reporting exactly half of the bytes for both of this kernel's read shapes, hence the ×2. The read traffic above the
algorithmic figure is (a) the second read of the `DMatch` records at copy-out (they are not kept on chip between binning
and copy-out: 655 MB per launch, the first thing to remove) and (b) keypoint-table reads served by L2 / Infinity Cache
(the 80 KB frame-B table is copied into LDS once per pair).

SQ counters per pair (`r01_sq_counters.csv`, 1024-pair launches): {sq['SQ_INSTS_VALU']/pairs/1e3:.0f} k VALU, {sq['SQ_INSTS_SALU']/pairs/1e3:.0f} k SALU,
{sq['SQ_INSTS_LDS']/pairs/1e3:.1f} k LDS wave-instructions ({sq['SQ_INSTS_LDS_ATOMIC']/pairs/1e3:.1f} k of them atomics); a wave is issuing {100*sq['SQ_ACTIVE_INST_ANY']/sq['SQ_WAVE_CYCLES']:.0f} % of its
life, waits at `s_waitcnt`/barriers {100*sq['SQ_WAIT_ANY']/sq['SQ_WAVE_CYCLES']:.0f} %, is stalled at issue {100*sq['SQ_WAIT_INST_ANY']/sq['SQ_WAVE_CYCLES']:.0f} %; bank conflicts are {100*sq['SQ_LDS_BANK_CONFLICT']/sq['SQ_LDS_IDX_ACTIVE']:.0f} % of the LDS's
active cycles (random-address atomics and `ds_read_b128`).
**What bounds the kernel today is not HBM** but the CU itself: one 16-wave workgroup per CU (LDS-bound, 147 KB per pair),
chains of dependent LDS operations, and an instruction stream of ≈{(sq['SQ_INSTS_VALU']+sq['SQ_INSTS_SALU']+sq['SQ_INSTS_LDS'])/pairs/1e3:.0f} k wave-instructions per pair of which no single
unit is saturated (SIMD VALU ≈ {100*sq['SQ_ACTIVE_INST_VALU']*4/sq['SQ_WAVE_CYCLES']:.0f} % busy). In-kernel phase stamps (diagnostic build, `tools/phase_timing.py`,
`r01_phase_cycles.json`; shader cycles per pair seen by wave 0, {ph['total_cycles']/1e3:.0f} k total): binning {bin_c/1e3:.0f} k (match loads + staging
frame B {c('bin_qt_loads')/1e3:.0f} k), region tables {c('region_tables')/1e3:.0f} k; summed over the four grid types: table clear {c('clear')/1e3:.0f} k, insert {ins_c/1e3:.0f} k (of which
{c('insert_barrier')/1e3:.0f} k is wave 0 waiting for the other waves), verify {c('verify')/1e3:.0f} k, mark {c('mark')/1e3:.0f} k; select {c('count_select')/1e3:.0f} k, scan {c('out_scan')/1e3:.0f} k, copy-out {c('copy_out')/1e3:.0f} k.
The round's speed-ups came from instruction count, not from memory: 2.2 M → 3.8 M pairs/s by keeping match state in
registers, per-cell bucket regions read with one `ds_read_b128`, a branch-free staged insert with per-match flags held
as lane masks, half-cell descriptor tables, region headers instead of an arg-max scan, and staging frame B in LDS.
Measured and rejected this round: 512-thread workgroups with twice the matches in flight (−27 %), 1.5× / 1.25× table
regions instead of 2× (−4 % / −8 %), an L2 touch-ahead of the next round's match array (−4 %),
a persistent-workgroup variant that issues the next pair's loads in front of the copy-out (−17 %: `vmcnt` returns in order,
so the copy-out's own loads queue behind the prefetch), static `s_setprio` by wave age (±0 %), explicit address-space-3
pointers for the atomics (fewer adds, more `s_nop` hazard fillers: ±0 %), and — kept in the tree
behind `GMS_OCC2=1`, covered by a test — `gms_kernel_occ2.hip`, a variant cut down to 78 KB of LDS and 64 VGPRs so that two
workgroups (32 waves) share a CU: bit-exact, but −13 % (1.25× regions, an arg-max scan instead of region headers, and
register spills cost more than the second workgroup hides), which also says the 16-wave kernel is closer to the CU's
issue/LDS throughput than its wait-dominated wave timeline suggests.

Side measurements (`r01_side_measurements.json`): the one-shot host-pointer call `gms_match` on a 10k-match pair takes
{m['one_shot_10k_rot0_scale0']['gpu_call_ms_incl_pcie']:.2f} ms end to end including the PCIe copies ({m['one_shot_10k_rot0_scale0']['gpu_pairs_per_s']:.0f} pairs/s; the oracle needs {m['one_shot_10k_rot0_scale0']['cpu_oracle_ms_1thread']:.2f} ms on one core) and {m['one_shot_10k_rot1_scale1']['gpu_call_ms_incl_pcie']:.2f} ms with
rotation + scale (oracle {m['one_shot_10k_rot1_scale1']['cpu_oracle_ms_1thread']:.0f} ms) — PCIe-inclusive rates, never the headline `value`. BASELINE config 4 (3840×2160,
50k matches, large-pair kernel): {m['config4_50k_rot0_scale0']['gpu_call_ms_incl_pcie']:.1f} ms / {m['config4_50k_rot1_scale1']['gpu_call_ms_incl_pcie']:.1f} ms per call without / with rotation + scale (oracle {m['config4_50k_rot0_scale0']['cpu_oracle_ms_1thread']:.1f} / {m['config4_50k_rot1_scale1']['cpu_oracle_ms_1thread']:.0f} ms).

"""
s = open('DESIGN.md').read()
a, z = s.index('## 6. Measurements (round 1'), s.index('## 8. What comes next')
s = s[:a] + sec6 + s[z:]
s = re.sub(r"PMC traffic is [0-9.]+ GB per launch against", f"PMC traffic is {t['hbm_bytes_per_launch']/1e9:.2f} GB per launch against", s)
open('DESIGN.md', 'w').write(s)

bm = open('BASELINE.md').read()
a = bm.index('## Measured numbers')
bm = bm[:a] + f"""## Measured numbers (round 1; full detail in DESIGN.md §6 and `profiles/r01_*`)

Same run, same inputs (`python bench.py` on one MI355X box; host CPU AMD EPYC 9575F, 16 threads used):

| config | flags | CPU restatement | 1 × MI355X | ratio | parity |
|---|---|---|---|---|---|
| 3-shape batch: 1080p, 10k matches/pair, 4096 pairs resident per step | (0,0,6.0) | {b['cpu_baseline']['value_1thread']:.0f} pairs/s (1 thread), {b['cpu_baseline']['value']:.0f} pairs/s (16 threads, one pair per thread) | {b['value']:.0f} pairs/s, kernel {b['roofline']['kernel_ms_per_launch']:.3f} ms per launch | {b['gpu_vs_cpu']:.0f}× vs 16 threads | 256 sampled pairs bit-exact |
| same pairs | (1,1,6.0) | {rs['cpu_baseline']['value_1thread']:.0f} / {rs['cpu_baseline']['value']:.0f} pairs/s | {rs['value']:.0f} pairs/s | {rs['value']/rs['cpu_baseline']['value']:.0f}× | 32 sampled pairs bit-exact |

Roofline (algorithmic 32·M + 16·K bytes per pair): {b['roofline']['achieved']:.0f} GB/s of 8000 GB/s = {b['roofline']['frac']:.3f}; PMC-measured HBM-side traffic {t['hbm_bytes_per_launch']/1e9:.2f} GB
per launch against 1.64 GB algorithmic. Targets of BASELINE.json: ≥ 10× CPU — met ({b['gpu_vs_cpu']:.0f}× vs 16 host threads); ≥ 40 % of HBM roofline —
not met ({100*b['roofline']['frac']:.0f} %: the kernel is bound by the CU — LDS chains and instruction issue at one workgroup per CU — not by HBM, see DESIGN.md §6);
multi-GPU scaling is measured by the driver (`bench.py --gpus N`, pairs sharded, no collective).
"""
open('BASELINE.md', 'w').write(bm)
print("docs updated:", b['value'], b['roofline']['frac'])
