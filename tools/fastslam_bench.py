#!/usr/bin/env python3
"""FastSLAM 1.0 update rate -- not the headline metric (bench.py), a measurement for DESIGN.md.
2000 particles, 200 landmarks over a 25 m disc, 30 measurements per update; state re-seeded every step.
  python tools/fastslam_bench.py [--cpu]    (--cpu also times the oracle on the host cores)"""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
sc = pkg.scenarios
N, NM, NZ, HYP = [int(os.environ.get(k, d)) for k, d in (("FS_N", 2000), ("FS_NM", 200), ("FS_NZ", 30), ("FS_HYP", 1))]
scen = sc.make_scenario(N, NM, NZ, seed=4242, rmax=25.0)
f = pkg.FastSLAM(N, gm_capacity=384, max_hypotheses=HYP)    # FS_HYP > 1: MH-FastSLAM (dense table: keep FS_NM, FS_NZ <= 64)
sc.load_scenario(f, scen)
for i in range(N):
    f.import_gm(i, np.zeros(NM), scen["mean"][i], scen["cov"][i])
f.set_fastslam_config(f.fs_config)
f.save_state()
for _ in range(5):
    f.restore_state(); f.fastslam_update(scen["Z"])
f.synchronize()
S = 100
t0 = time.perf_counter()
for _ in range(S):
    f.restore_state()
    f.fastslam_update(scen["Z"])
f.synchronize()
dt = time.perf_counter() - t0
ns = f.last_kernel_ns()
print("device: %.4f ms/update (%.1f updates/s); associate+KF kernel %.1f us, prune+new landmarks %.1f us; map size after %d; particles after %d" %
      (dt / S * 1e3, S / dt, ns[0] / 1e3, ns[3] / 1e3, int(f.gm_sizes().mean()), f.n))
if "--cpu" in sys.argv:
    import importlib
    ob = importlib.import_module("oracle.binding")
    n = 256
    sub = dict(scen); sub.update(n=n, poses=scen["poses"][:n], w=scen["w"][:n], mean=scen["mean"][:n], cov=scen["cov"][:n], particle_w=scen["particle_w"][:n])
    def make():
        o = ob.OracleFilter(n)
        sc.load_scenario(o, sub)
        for i in range(n):
            o.import_gm(i, np.zeros(NM), scen["mean"][i], scen["cov"][i])
        ocfg = o.default_fastslam_config()
        ocfg.maxNDataAssocHypotheses = HYP
        o.set_fastslam_config(ocfg)
        return o
    make().fastslam_update(scen["Z"])          # warm-up instance: the first OpenMP region pays for starting the thread team
    o = make()
    t0 = time.perf_counter()
    o.fastslam_update(scen["Z"])
    dt = time.perf_counter() - t0
    print("oracle (OpenMP, %d host threads): %.1f ms for %d particles -> %.2f updates/s at %d particles" % (os.cpu_count(), dt * 1e3, n, 1.0 / (dt * N / n), N))
