#!/usr/bin/env python3
"""Tuning aid: what a timed region of K stream-ordered steps costs beside K x (one step) -- the intercept of wall time over K.
GPU box.   python tools/fixed_cost_probe.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402
import torch  # noqa: E402

pkg = g.load_package()
sc = pkg.scenarios
scen = sc.make_scenario(2000, 200, 30, seed=12345)
f = pkg.RBPHDFilter(2000, gm_capacity=384)
sc.load_scenario(f, scen)
f.save_state()
Z = scen["Z"]
f.set_step_timing_stride(1 << 20)
for _ in range(300):
    f.restore_state(); f.step_async(Z, True)
f.synchronize()
for sync_name, sync in (("torch.cuda.synchronize", torch.cuda.synchronize), ("rfsgpu_synchronize (hipStreamSynchronize)", f.synchronize)):
    rows = []
    for K in (1, 2, 5, 10, 20, 50, 100):
        best = 1e9
        for rep in range(7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(K):
                f.restore_state(); f.step_async(Z, True)
            sync()
            best = min(best, time.perf_counter() - t0)
        rows.append((K, best * 1e6))
    ks = np.array([r[0] for r in rows], float); ts = np.array([r[1] for r in rows])
    slope, icpt = np.polyfit(ks[2:], ts[2:], 1)
    print(sync_name, " ".join("K=%d: %.1f us" % r for r in rows), "| per step %.2f us, intercept %.1f us" % (slope, icpt))
t = []
for _ in range(200):
    t0 = time.perf_counter(); torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
print("torch.cuda.synchronize() on an idle device: median %.1f us" % (np.median(t) * 1e6))
