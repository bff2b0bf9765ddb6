#!/usr/bin/env python3
"""configs[4] (C5: 1000 particles x 200 Gaussians x 50 measurements, 40 evaluation points, 10-sigma weighting gate ->
partitions with extended dimension > 8 -> the Murty-200 path) update rate -- a measurement for DESIGN.md, not the headline.
State re-seeded every step.   python tools/c5_bench.py [--cpu]"""
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
N = int(os.environ.get("C5_N", 1000))
scen = sc.make_scenario(N, 200, 50, seed=555, n_clutter=10, n_eval=40, weighting_md=10.0, weights=(0.8, 1.0))
f = pkg.RBPHDFilter(N, gm_capacity=448)
sc.load_scenario(f, scen)
if "--exact" in sys.argv:      # RFSGPU_PARTITION_EXACT: untruncated partition sums instead of Murty-200 (opt-in; SURVEY 8(d) asks for both timings)
    f.set_partition_mode(True)
f.save_state()
f.set_phase_timing(True)   # (per-phase kernel times below)
for _ in range(2):
    f.restore_state(); f.update(scen["Z"])
S = int(os.environ.get("C5_STEPS", 10))
t0 = time.perf_counter()
for _ in range(S):
    f.restore_state()
    f.update(scen["Z"])
dt = time.perf_counter() - t0
ns = f.last_kernel_ns()
t = f.getTimingInfo()
print("C5 RB-PHD update%s, %d particles: %.3f ms/update (%.2f updates/s); kernels us: update_map %.1f, weighting (incl. Murty jobs) %.1f, merge+prune %.1f" %
      (" (exact partition mode)" if "--exact" in sys.argv else "", N, dt / S * 1e3, S / dt, ns[0] / 1e3, ns[1] / 1e3, ns[2] / 1e3))
if "--cpu" in sys.argv:
    import importlib
    ob = importlib.import_module("oracle.binding")
    n = 256
    sub = dict(scen); sub.update(n=n, poses=scen["poses"][:n], w=scen["w"][:n], mean=scen["mean"][:n], cov=scen["cov"][:n], particle_w=scen["particle_w"][:n])
    ow = ob.OracleFilter(n)                     # warm-up instance: the first OpenMP region pays for starting the thread team
    sc.load_scenario(ow, sub)
    ow.update(scen["Z"])
    o = ob.OracleFilter(n)
    sc.load_scenario(o, sub)
    t0 = time.perf_counter()
    o.update(scen["Z"])
    dt = time.perf_counter() - t0
    print("oracle (OpenMP, %d host threads): %.1f ms for %d particles -> %.3f updates/s at %d particles; Murty calls %d" % (os.cpu_count(), dt * 1e3, n, 1.0 / (dt * N / n), N, o.murty_calls()))
