#!/usr/bin/env python3
"""The synchronous drop-in call (rfsgpu_update: what the reference-side binding calls once per RBPHDFilter::update) at C2a's shape:
wall time per call against the stream-ordered step's.   python tools/sync_update_bench.py [n_particles]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
sc = pkg.scenarios
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
scen = sc.make_scenario(n, 200, 30, seed=12345)
f = pkg.RBPHDFilter(n, gm_capacity=384)
sc.load_scenario(f, scen)
f.save_state()
Z = scen["Z"]
for _ in range(600):          # (a fresh process spends its first ~100 ms at low clocks: a 200-particle loop measured 260 us per call there, 131 afterwards)
    f.restore_state(); f.update(Z)
S = 300
t0 = time.perf_counter()
for _ in range(S):
    f.restore_state(); f.update(Z)
dt = (time.perf_counter() - t0) / S
ns = f.last_kernel_ns()
t0 = time.perf_counter()
for _ in range(S):
    f.restore_state(); f.step_async(Z, True)
f.synchronize()
dta = (time.perf_counter() - t0) / S
print("%d particles: restore + rfsgpu_update (synchronous) %.1f us per call; restore + rfsgpu_step_async %.1f us per step; kernels of the last sync call (us): %s" % (n, dt * 1e6, dta * 1e6, [round(x / 1e3, 1) for x in ns]))
