"""Runs the oracle (C restatements) under AddressSanitizer + UBSan on the CPU: make -C oracle asan, then
  LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 python tools/oracle_asan_run.py"""
import ctypes, sys, os, numpy as np, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
import gms_oracle
gms_oracle._lib = None
real_cdll = ctypes.CDLL
gms_oracle.C.CDLL = lambda path: real_cdll(os.path.join(os.path.dirname(path), "libgms_oracle_asan.so"))
import cases
n = 0
for name, c in list(cases.adversarial_cases().items()) + [("rand", cases.random_pair(1, n=3000)), ("big", cases.random_pair(2, n=40000, size1=(3840, 2160)))]:
    for rot, scale in cases.FLAGS:
        rc, out, mask, res = gms_oracle.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], rot, scale, 6.0)
        assert rc == 0; n += 1
for name, c in cases.domain_error_cases().items():
    rc, *_ = gms_oracle.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], True, True, 6.0)
    assert rc != 0; n += 1
synth = importlib.import_module("sfm-gms_amd.synth")
d = synth.sequence_descriptors(5, 2, 400, "orb"); gms_oracle.bf_match(d[0], d[1], True)
d = synth.sequence_descriptors(5, 2, 400, "sift"); gms_oracle.bf_match(d[0], d[1], False)
# the multi-threaded driver with its per-thread scratch (the cpu_baseline leg of bench.py) and the keypoint source's statement
frames = synth.make_sequence(7, 6, size=(1280, 720), n_kp=1500)
dist = importlib.import_module("sfm-gms_amd.dist")
pairs = dist.pair_table(6, 0, 12, 1500)
matches = np.concatenate([dist.synth_matches_host(k, 1500, 0.5) for k in range(12)])
foff = np.arange(7, dtype=np.int64) * 1500
wh = np.array([(1280, 720)] * 6, dtype=np.int32).reshape(-1)
for rot, scale in ((False, False), (True, True)):
    failed, out, res, _ = gms_oracle.batch(np.concatenate(frames), foff, wh, pairs, matches, rot, scale, 6.0, 4)
    assert failed == 0; n += 12
imgs = synth.make_textured_images(3, 1, size=(320, 200))
kp, rows = gms_oracle.detect(imgs[0], 15, 400)
assert len(kp) > 50 and gms_oracle.describe(imgs[0], kp)[2].tobytes() == rows.tobytes()
kp2, _ = gms_oracle.detect(imgs[0], 15, 40)
assert len(kp2) == 40
print("asan/ubsan clean:", n, "filter calls (one-shot and 4-thread batch) + matcher calls + detector calls")
