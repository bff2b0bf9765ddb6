#!/usr/bin/env python3
"""OpenMP scaling of the CPU baseline (oracle, -O3 -march=native build) on this host, per phase (state the
CPU model, thread count and binding).  Not part of the product.  usage: tools/cpu_scaling.py [particles] [nm] [nz]"""
import os
import sys
import time
os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_PLACES", "cores")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
import ctypes as C
from oracle import binding as ob  # noqa: E402
pkg = load_package()
sc = pkg.scenarios
n, nm, nz = [int(x) for x in (sys.argv[1:4] + [4096, 200, 30][len(sys.argv[1:4]):])]
so, flags = ob.build_fast()
lib = C.CDLL(so)
model, logical, physical = ob.cpu_info()
print("CPU: %s; %d logical CPUs, %d physical cores; build flags %s; OMP_PROC_BIND=%s OMP_PLACES=%s" %
      (model, logical, physical, " ".join(flags), os.environ["OMP_PROC_BIND"], os.environ["OMP_PLACES"]))
scen = sc.make_scenario(n, nm, nz, seed=12345)
base = None
for thr in [t for t in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512) if t <= logical]:
    n_s = min(n, max(64, 32 * thr))
    sub = dict(scen); sub.update(n=n_s, poses=scen["poses"][:n_s], w=scen["w"][:n_s], mean=scen["mean"][:n_s], cov=scen["cov"][:n_s], particle_w=scen["particle_w"][:n_s])
    ob.set_threads(thr)
    lib.rfsor_set_threads(C.c_int(thr))
    best = None
    for rep in range(4):
        o = ob.OracleFilter(n_s, stable_sort=False, lib=lib)
        sc.load_scenario(o, sub)
        o.reset_timing()
        t0 = time.perf_counter()
        o.update(scen["Z"])
        dt = time.perf_counter() - t0
        tm = o.getTimingInfo()
        o.close()
        if best is None or dt < best[0]:
            best = (dt, tm)
    per_particle_us = best[0] / n_s * 1e6
    if base is None:
        base = per_particle_us
    tm = best[1]
    print("threads %4d  particles %5d  %.1f us/particle  speed-up %.1fx  efficiency %.2f  | phases ms: map %.2f weight %.2f merge %.2f prune %.2f" %
          (thr, n_s, per_particle_us, base / per_particle_us, base / per_particle_us / thr,
           tm.mapUpdate_wall / 1e6, tm.particleWeighting_wall / 1e6, tm.mapMerge_wall / 1e6, tm.mapPrune_wall / 1e6))
