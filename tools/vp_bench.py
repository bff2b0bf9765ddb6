#!/usr/bin/env python3
"""Victoria Park (configs[3]) update rate -- not the headline metric (bench.py), a measurement for DESIGN.md.
5000 particles, Victoria-Park-shaped maps (40 landmarks, 12 measurements per update); state re-seeded every step."""
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
N, NM, NZ = [int(os.environ.get(k, d)) for k, d in (("VP_N", 5000), ("VP_NM", 40), ("VP_NZ", 12))]
scen = sc.make_vp_scenario(N, NM, NZ, seed=4321, scan="ragged")
f = pkg.RBPHDFilter(N, gm_capacity=int(os.environ.get("VP_CAP", 192)), model=pkg.capi.MODEL_VICTORIAPARK_3D)
sc.load_scenario(f, scen)
f.save_state()
for _ in range(5):
    f.restore_state(); f.update(scen["Z"])
S = int(os.environ.get("VP_STEPS", 100))
t0 = time.perf_counter()
acc = np.zeros(4)
for _ in range(S):
    f.restore_state()
    f.update(scen["Z"])
    acc += np.array(f.last_kernel_ns(), dtype=np.float64)
dt = time.perf_counter() - t0
ns = acc / S      # mean over the timed steps
print("VP RB-PHD update, %d particles x %d landmarks x %d measurements: %.4f ms/update (%.1f updates/s); kernels us: update_map %.1f, weighting %.1f, merge+prune %.1f" %
      (N, NM, NZ, dt / S * 1e3, S / dt, ns[0] / 1e3, ns[1] / 1e3, ns[2] / 1e3))
if "--cpu" in sys.argv:
    import importlib
    ob = importlib.import_module("oracle.binding")
    n = 512
    sub = dict(scen); sub.update(n=n, poses=scen["poses"][:n], w=scen["w"][:n], mean=scen["mean"][:n], cov=scen["cov"][:n], particle_w=scen["particle_w"][:n])
    ow = ob.OracleFilter(n, model=pkg.capi.MODEL_VICTORIAPARK_3D)                     # warm-up instance: the first OpenMP region pays for starting the thread team
    sc.load_scenario(ow, sub)
    ow.update(scen["Z"])
    o = ob.OracleFilter(n, model=pkg.capi.MODEL_VICTORIAPARK_3D)
    sc.load_scenario(o, sub)
    t0 = time.perf_counter()
    o.update(scen["Z"])
    dt = time.perf_counter() - t0
    print("oracle (OpenMP, all host threads): %.1f ms for %d particles -> %.2f updates/s at %d particles" % (dt * 1e3, n, 1.0 / (dt * N / n), N))
