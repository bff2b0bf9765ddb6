#!/usr/bin/env python3
"""RB-PHD update rate at an arbitrary shape (not the headline metric: bench.py) -- e.g. configs[2]'s shard, 2500 particles x
500 Gaussians x 30 measurements with a 5 m range limit:
    SB_N=2500 SB_NM=500 SB_NZ=30 SB_RMAX=5 SB_CAP=640 python tools/shape_bench.py [--cpu]
State re-seeded from a device snapshot every step; fused step (update_async) and the three stand-alone kernels."""
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
N, NM, NZ, CAP = [int(os.environ.get(k, d)) for k, d in (("SB_N", 2500), ("SB_NM", 500), ("SB_NZ", 30), ("SB_CAP", 640))]
RMAX = float(os.environ.get("SB_RMAX", 5.0))
scen = sc.make_scenario(N, NM, NZ, seed=777, rmax=RMAX)
f = pkg.RBPHDFilter(N, gm_capacity=CAP)
sc.load_scenario(f, scen)
f.save_state()
f.set_phase_timing(True)   # (per-phase kernel times below)
for _ in range(3):
    f.restore_state(); f.update(scen["Z"])
ns = f.last_kernel_ns()
S = int(os.environ.get("SB_STEPS", 50))
f.synchronize()
t0 = time.perf_counter()
for _ in range(S):
    f.restore_state()
    f.update_async(scen["Z"])
f.synchronize()
dt = time.perf_counter() - t0
avg, n = f.kernel_time_stats()
print("RB-PHD update %d particles x %d Gaussians x %d measurements (cap %d): %.3f ms/step (%.1f steps/s); fused kernel %.1f us; stand-alone kernels us: update_map %.1f, weighting %.1f, merge+prune %.1f; map size after %d" %
      (N, NM, NZ, CAP, dt / S * 1e3, S / dt, avg[0] * 1e-3, ns[0] / 1e3, ns[1] / 1e3, ns[2] / 1e3, int(f.gm_sizes().mean())))
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
    print("oracle (OpenMP, %d host threads): %.1f ms for %d particles -> %.3f steps/s at %d particles" % (os.cpu_count(), dt * 1e3, n, 1.0 / (dt * N / n), N))
