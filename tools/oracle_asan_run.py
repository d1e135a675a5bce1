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
print("asan/ubsan clean:", n, "filter calls + matcher calls")
