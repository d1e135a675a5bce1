#!/usr/bin/env python3
"""bench.py -- GMS-filtered image pairs per second on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (gms_filter_device, one launch) over one batch of synthetic image
pairs whose keypoint tables and putative matches are already resident in HBM. Workload = BASELINE
config 3 ("1080p sequence, 10k features per frame, all pairs sharded across GPUs") at the flags the
reference's DisparityUtil.cpp:149,299 call sites use (withRotation=false, withScale=false, 6.0); the
(true, true) flags of FeatureMatchUtil.cpp:69 are reported beside it under "rot_scale".

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Pairs are independent: each rank owns a contiguous shard of the global pair list and
there is no collective on the data path ("scaling": "weak": pairs per GPU are fixed). torch is used for
device memory, streams, events and the rendezvous only.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "sfm-gms_amd"
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s achievable by a copy


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=4096, help="image pairs per step per GPU")
    ap.add_argument("--frames", type=int, default=128, help="frames of the synthetic sequence per GPU shard")
    ap.add_argument("--features", type=int, default=10000)
    ap.add_argument("--inlier-frac", type=float, default=0.5)
    ap.add_argument("--cpu-pairs", type=int, default=256, help="sample size of the CPU baseline / parity check")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg (profiling runs)")
    ap.add_argument("--no-extra", action="store_true", help="skip the rot+scale side measurement")
    return ap.parse_args()


def build_workload(args, rank, world, dev, pkg, synth, ctx):
    """Frames of one synthetic 1080p sequence + this rank's shard of the pair list, all on the GPU."""
    size = (1920, 1080)
    n_kp = args.features
    frames = synth.make_sequence(1000 + rank, args.frames, size=size, n_kp=n_kp)
    batch = importlib.import_module(PKG + ".batch")
    table = batch.FrameTable(ctx, frames, [size] * args.frames, device=dev)

    # global pair list of the whole job = world * pairs; this rank takes its contiguous block
    n_total = args.pairs * world
    lo, hi = pkg.shard_range(n_total, rank, world)
    n_pairs = hi - lo
    all_pairs = pkg.all_pairs_count(args.frames)
    pairs = np.zeros(n_pairs, dtype=pkg.PAIR_DTYPE)
    for i in range(n_pairs):
        a, b = pkg.pair_from_index((i * 7919 + rank) % all_pairs, args.frames)
        pairs[i] = (a, b, n_kp, 0, i * n_kp)

    # putative matches generated on the device (M = N1, queryIdx = i; a fraction are true correspondences
    # i -> i, the rest uniformly random), like BFMatcher output without cross-check
    g = torch.Generator(device=dev)
    g.manual_seed(0x5F3759DF ^ (77 + rank))
    total_m = n_pairs * n_kp
    q = torch.arange(n_kp, device=dev, dtype=torch.int32).repeat(n_pairs)
    is_in = torch.rand(total_m, device=dev, generator=g) < args.inlier_frac
    rnd = torch.randint(0, n_kp, (total_m,), device=dev, generator=g, dtype=torch.int32)
    t = torch.where(is_in, q, rnd)
    dist = torch.rand(total_m, device=dev, generator=g) * 256.0
    d_matches = torch.empty((total_m, 4), dtype=torch.int32, device=dev)
    d_matches[:, 0] = q
    d_matches[:, 1] = t
    d_matches[:, 2] = 0
    d_matches[:, 3] = dist.view(torch.int32)
    del q, is_in, rnd, t, dist
    d_pairs = torch.from_numpy(pairs.view(np.uint8).reshape(-1)).to(dev)
    d_out = torch.zeros((total_m, 4), dtype=torch.int32, device=dev)
    d_res = torch.zeros((n_pairs, 4), dtype=torch.int32, device=dev)
    return dict(size=size, frames=frames, table=table, pairs=pairs, d_pairs=d_pairs, d_matches=d_matches,
                d_out=d_out, d_res=d_res, n_pairs=n_pairs, n_kp=n_kp)


def launch(ctx, wl, rot, scale):
    t = wl["table"]
    ctx.filter_device(t.d_pts.data_ptr(), t.d_frame_off.data_ptr(), t.n_frames, wl["d_pairs"].data_ptr(),
                      wl["n_pairs"], wl["n_kp"], wl["d_matches"].data_ptr(), wl["d_out"].data_ptr(),
                      wl["d_res"].data_ptr(), None, rot, scale, 6.0)


def timed_steps(ctx, wl, stream, steps, warmup, rot, scale, dist):
    """W warm-up steps, then exactly K steps between barrier + synchronize brackets. Returns
    (wall seconds of the K steps, mean kernel ms per launch from HIP events on the launch stream)."""
    with torch.cuda.stream(stream):
        for _ in range(warmup):
            launch(ctx, wl, rot, scale)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        for s in range(steps):
            ev[s][0].record(stream)
            launch(ctx, wl, rot, scale)
            ev[s][1].record(stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    return wall, kern_ms


def cpu_leg(args, wl, pkg, rot, scale, n_sample, threads):
    """Oracle (CPU restatement of the reference's matchGMS) on a bounded sample of the same pairs:
    returns (pairs/s all threads, pairs/s one thread, parity ok, sample description)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gms_oracle
    n = min(n_sample, wl["n_pairs"])
    n_kp = wl["n_kp"]
    pairs = wl["pairs"][:n].copy()
    matches = wl["d_matches"][: n * n_kp].cpu().numpy().view(pkg.DMATCH_DTYPE).reshape(-1)
    kp_all = np.concatenate(wl["frames"])
    foff = wl["table"].frame_off_host
    wh = np.array([wl["size"]] * len(wl["frames"]), dtype=np.int32).reshape(-1)
    # parity of the GPU output on the sample (d_out/d_res hold the last launch with these flags)
    gpu_out = wl["d_out"][: n * n_kp].cpu().numpy().view(pkg.DMATCH_DTYPE).reshape(-1)
    gpu_res = wl["d_res"][:n].cpu().numpy().view(pkg.RESULT_DTYPE).reshape(-1)

    def run(sel, nthreads, reps):
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            failed, out, res, _ = gms_oracle.batch(kp_all, foff, wh, sel, matches, rot, scale, 6.0, nthreads)
            dt = time.perf_counter() - t0
            assert failed == 0
            best = dt if best is None else min(best, dt)
        return len(sel) / best, out, res

    rate1, _, _ = run(pairs[: max(8, n // 8)], 1, 2)
    rate_mt, out, res = run(pairs, threads, 3)
    ok = bool(np.array_equal(res["n_inliers"], gpu_res["n_inliers"]) and
              np.array_equal(res["best_scale"], gpu_res["best_scale"]) and
              np.array_equal(res["best_rot"], gpu_res["best_rot"]))
    if ok:
        for i in range(n):
            k = int(res["n_inliers"][i])
            o = int(pairs["match_off"][i])
            if out[o:o + k].tobytes() != gpu_out[o:o + k].tobytes():
                ok = False
                break
    return rate_mt, rate1, ok, n


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # GMS_BENCH_ONE_DEVICE=1 is a rehearsal mode for a one-GPU box: every rank uses cuda:0 and the rendezvous
    # runs over gloo (RCCL refuses two ranks on one device). The driver's multi-GPU runs do not set it.
    one_device = os.environ.get("GMS_BENCH_ONE_DEVICE", "0") == "1"
    dev_index = 0 if one_device else local_rank
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(dev_index)
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)

    pkg = importlib.import_module(PKG)
    synth = importlib.import_module(PKG + ".synth")
    ctx = pkg.GmsContext(dev_index)  # raises if the HIP extension is missing: no fallback
    stream = torch.cuda.Stream(device=dev)
    ctx.set_stream(stream.cuda_stream)

    wl = build_workload(args, rank, world, dev, pkg, synth, ctx)
    torch.cuda.synchronize()

    wall, kern_ms = timed_steps(ctx, wl, stream, args.steps, args.warmup, False, False, dist)
    # whole-job aggregate: every rank ran the same number of pairs; time = max over ranks
    t = torch.tensor([wall], dtype=torch.float64, device="cpu" if one_device else dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall_max = float(t.item())
    pairs_total = args.pairs * world * args.steps
    value = pairs_total / wall_max

    if rank == 0:
        n_kp = wl["n_kp"]
        res = wl["d_res"].cpu().numpy().view(pkg.RESULT_DTYPE).reshape(-1)
        assert (res["status"] == 0).all()
        kept = int(res["n_inliers"].astype(np.int64).sum())
        # algorithmic bytes per launch: 32*M + 16*K per pair
        alg_bytes = 32.0 * n_kp * wl["n_pairs"] + 16.0 * kept
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "gms_filtered_image_pairs_per_sec", "value": value, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i32", "data": "synthetic",
            "config": {"workload": "config3 1080p sequence: 10k keypoints/frame, M=10k putative matches/pair, "
                                   "matchGMS(withRotation=false, withScale=false, thresholdFactor=6.0)",
                       "image_size": [1920, 1080], "features": n_kp, "pairs_per_step_per_gpu": args.pairs,
                       "frames_per_gpu": args.frames, "inlier_frac": args.inlier_frac,
                       "mean_kept_per_pair": kept / wl["n_pairs"], "sharding": f"pairs x{world}, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "kernel": "gms::filter_kernel_dense<10, false, 1024>",
                         "kernel_ms_per_launch": kern_ms, "algorithmic_bytes_per_launch": alg_bytes},
        }
        run_cpu = not args.no_cpu and world == 1  # the CPU baseline is timed on rank 0 of the 1-GPU run only
        if run_cpu:
            rate_mt, rate1, ok, n = cpu_leg(args, wl, pkg, False, False, args.cpu_pairs, args.cpu_threads)
            line["cpu_baseline"] = {"value": rate_mt, "unit": "pairs/s", "cores": args.cpu_threads, "kind": "port",
                                    "sample": f"first {n} pairs of the step's batch, oracle/gms_ref.c, one pair "
                                              f"per thread, best of 3", "value_1thread": rate1,
                                    "host_cpus": os.cpu_count()}
            line["parity"] = {"pairs_checked": n, "bit_exact": ok}
            line["gpu_vs_cpu"] = value / rate_mt
        if not args.no_extra and world == 1:
            sub = dict(wl)
            n_sub = min(wl["n_pairs"], 512)
            sub["n_pairs"] = n_sub
            w2, k2 = timed_steps(ctx, sub, stream, 8, 2, True, True, None)
            extra = {"workload": "same pairs, matchGMS(withRotation=true, withScale=true, 6.0) "
                                 "(FeatureMatchUtil.cpp:69 flags), 8 rot x 5 scale x 4 grids",
                     "pairs_per_step": n_sub, "value": n_sub * 8 / w2, "unit": "pairs/s", "kernel_ms_per_launch": k2}
            if run_cpu:
                rate_mt2, rate12, ok2, n2 = cpu_leg(args, sub, pkg, True, True, min(32, args.cpu_pairs),
                                                    args.cpu_threads)
                extra["cpu_baseline"] = {"value": rate_mt2, "cores": args.cpu_threads, "value_1thread": rate12,
                                         "sample": f"first {n2} pairs"}
                extra["parity"] = {"pairs_checked": n2, "bit_exact": ok2}
            line["rot_scale"] = extra
        print(json.dumps(line))
    elif not args.no_extra:
        pass
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
