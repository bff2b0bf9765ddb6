#!/usr/bin/env python3
"""The FIRST step of a filter that queues Murty partitions: until the pinned `hostSeen` flag flips the post
kernel is the light instance on a small grid (csrc/murty.h, murty_launch).  Prints the wall time of the first, second and later
updates of a fresh C5-shaped filter (1000 x 200 x 50, 10-sigma gate), and -- for the price of that grid -- the C2a step time (no
Murty work ever) with the same setting.   RFSGPU_MURTY_FIRST_BLOCKS=<n> python tools/murty_first_step.py"""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
if os.environ.get("RFS_LIB"):
    pkg.engine.LIB = os.environ["RFS_LIB"]      # a tools/variant_bench.py --build variant
sc = pkg.scenarios
# clocks up: a C2a filter stepping for a while; its steady step time is the empty-queue cost figure
s2 = sc.make_scenario(2000, 200, 30, seed=12345)
g = pkg.RBPHDFilter(2000, gm_capacity=384)
sc.load_scenario(g, s2)
g.save_state()
g.set_step_timing_stride(1 << 20)
for _ in range(2500):
    g.restore_state(); g.step_async(s2["Z"])
g.synchronize()
best = 1e9
for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(400):
        g.restore_state(); g.step_async(s2["Z"])
    g.synchronize()
    best = min(best, (time.perf_counter() - t0) / 400)
scen = sc.make_scenario(1000, 200, 50, seed=555, n_clutter=10, n_eval=40, weighting_md=10.0, weights=(0.8, 1.0))
times = []
for trial in range(3):
    f = pkg.RBPHDFilter(1000, gm_capacity=448)
    sc.load_scenario(f, scen)
    f.save_state()
    row = []
    for k in range(4):
        f.restore_state(); f.synchronize()
        t0 = time.perf_counter()
        f.step_async(scen["Z"]); f.synchronize()
        row.append((time.perf_counter() - t0) * 1e3)
    times.append(row)
    f.close()
t = np.array(times)
print("first blocks %s: C2a step %.2f us | C5 update #1 %.2f ms, #2 %.2f, #3 %.2f, #4 %.2f (median of 3 fresh filters)" %
      (os.environ.get("RFSGPU_MURTY_FIRST_BLOCKS", "default"), best * 1e6, *np.median(t, 0)))
