import importlib, sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle')
pkg = importlib.import_module("sfm-gms_amd"); import cases, gms_oracle
ctx = pkg.GmsContext(0)
def timeit(fn, reps):
    fn(); t=[]
    for _ in range(reps):
        t0=time.perf_counter(); fn(); t.append(time.perf_counter()-t0)
    return float(np.median(t))*1e3
for name, c in (("500", cases.random_pair(11, n=500, size1=(640, 480), inlier_frac=0.6)), ("10k", cases.random_pair(100, n=10000, inlier_frac=0.5)), ("16k", cases.random_pair(101, n=16000, inlier_frac=0.5))):
    for flags in ((False, False), (True, True)):
        got = ctx.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], *flags)
        want = gms_oracle.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], *flags)[1]
        assert got.tobytes() == want.tobytes()
        print(name, flags, round(timeit(lambda: ctx.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], *flags), 40), 4), "ms")
