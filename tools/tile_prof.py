import sys, json, importlib, argparse, torch
sys.path.insert(0, ".")
import bench
pkg = importlib.import_module("sfm-gms_amd"); synth = importlib.import_module("sfm-gms_amd.synth")
ctx = pkg.GmsContext(0); dev = torch.device("cuda", 0); stream = torch.cuda.Stream(device=dev); ctx.set_stream(stream.cuda_stream)
args = argparse.Namespace(pairs=64, frames=16, features=50000, inlier_frac=0.5)
wl = bench.build_workload(args, 0, 1, dev, pkg, synth, ctx)
w, k = bench.timed_steps(ctx, wl, stream, 3, 1, True, True, None)
print(json.dumps({"pairs_per_s": 64 * 3 / w, "ms_per_64_pairs": w / 3 * 1e3}))
