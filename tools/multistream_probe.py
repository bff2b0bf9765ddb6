#!/usr/bin/env python3
"""Tuning probe: does splitting one GPU's particles over K handles (= K HIP streams) overlap the kernel tails?"""
import sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
sc = pkg.scenarios
N, NM, NZ, CAP = 2000, 200, 30, 384
scen = sc.make_scenario(N, NM, NZ, seed=12345)
for K in (1, 2, 4, 8):
    hs = []
    for k in range(K):
        lo, hi = k * N // K, (k + 1) * N // K
        sub = dict(scen)
        sub.update(n=hi - lo, poses=scen["poses"][lo:hi], w=scen["w"][lo:hi], mean=scen["mean"][lo:hi], cov=scen["cov"][lo:hi],
                   particle_w=scen["particle_w"][lo:hi])
        f = pkg.RBPHDFilter(hi - lo, gm_capacity=CAP)
        sc.load_scenario(f, sub)
        f.save_state()
        hs.append(f)
    def step():
        for f in hs:
            f.restore_state()
            f.update_async(scen["Z"])
        for f in hs:
            f.weight_sums_async()
    for _ in range(20):
        step()
    for f in hs:
        f.synchronize()
    t0 = time.perf_counter()
    S = 200
    for _ in range(S):
        step()
    for f in hs:
        f.synchronize()
    dt = time.perf_counter() - t0
    print("K=%d handles: %.4f ms/step  (%.1f steps/s)" % (K, dt / S * 1e3, S / dt), flush=True)
    for f in hs:
        f.close()
