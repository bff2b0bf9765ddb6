#!/usr/bin/env python3
"""Batched MatPerm::calc (reference src/MatrixPermanent.cpp:41-112) micro-benchmark, n = 8..20: rfsgpu_mat_perm on a
batch of random matrices (one matrix per wavefront; the call includes the H2D / D2H copies of the batch) beside the oracle's
single-thread restatement on a few matrices, scaled.   python tools/matperm_bench.py [batch]"""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
from oracle import binding as ob  # noqa: E402
pkg = load_package()
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rng = np.random.default_rng(7)
pkg.mat_perm(rng.uniform(0, 1, (4, 8, 8)))      # first call: context
import ctypes as C
lib = pkg.load_library()
lib.rfsgpu_mat_perm_last_kernel_ms.restype = C.c_double
print("| n | batch | device ms (call, incl. copies) | kernel ms (HIP events) | TFLOP/s-equivalent (2^(n-1) x 2 n flop per matrix / kernel) | matrices/s (call) | oracle, 1 thread, matrices/s | ratio | max rel diff |")
print("|---|---|---|---|---|---|---|---|---|")
for n in range(8, 21):
    A = rng.uniform(0.0, 1.0, (batch, n, n))
    best = None
    kms = None
    for _ in range(3):
        t0 = time.perf_counter()
        out = pkg.mat_perm(A)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
        k_ = lib.rfsgpu_mat_perm_last_kernel_ms()
        kms = k_ if kms is None or k_ < kms else kms
    tflops = batch * (2.0 ** (n - 1)) * 2 * n / (kms * 1e-3) / 1e12
    k = max(1, min(batch, 2 ** max(0, 22 - n) // 16))
    t0 = time.perf_counter()
    ref = ob.mat_perm(A[:k])
    dtc = (time.perf_counter() - t0) / k
    rel = float(np.max(np.abs(out[:k] - ref) / np.abs(ref)))
    print("| %d | %d | %.3f | %.3f | %.2f | %.0f | %.1f | %.0fx | %.1e |" % (n, batch, best * 1e3, kms, tflops, batch / best, 1.0 / dtc, (batch / best) * dtc, rel))
