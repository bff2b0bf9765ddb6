#!/usr/bin/env python3
"""Tuning aid (GPU box): what a cost-ordered launch of the Victoria Park step buys.  configs[3]'s shape, re-seeded state: measure every
particle's duration in a step, launch the next steps with the longest first, time both forms (HIP events of the fused kernel)."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
if os.environ.get("RFS_LIB"):
    pkg.engine.LIB = os.environ["RFS_LIB"]
import bench
sc = pkg.scenarios
wl = bench.WORKLOADS["c4"]
scen = bench.make_scen(sc, wl, wl["n"])
f = pkg.RBPHDFilter(wl["n"], gm_capacity=wl["cap"], model=pkg.capi.MODEL_VICTORIAPARK_3D)
sc.load_scenario(f, scen)
f.save_state()
Z = scen["Z"]


def timed(steps=60):
    f.set_step_timing_stride(1)
    for _ in range(5):
        f.restore_state(); f.update_async(Z)
    f.synchronize(); f.kernel_time_stats()
    for _ in range(steps):
        f.restore_state(); f.update_async(Z)
    f.synchronize()
    ka, n = f.kernel_time_stats()
    return ka[0] / 1e3


f.step_launch_order(mode=0)           # slot == particle
base = timed()
cost = f.step_launch_order(mode=0)
print("identity order: fused kernel %.1f us; per-particle ticks p50 %.0f p90 %.0f max %.0f" % (base, np.percentile(cost, 50), np.percentile(cost, 90), cost.max()))
for name, order in (("longest first", np.argsort(-cost, kind="stable")), ("shortest first", np.argsort(cost, kind="stable")),
                    ("8 classes, longest first", np.argsort(-np.digitize(cost, np.quantile(cost, np.linspace(0, 1, 9)[1:-1])), kind="stable"))):
    f.step_launch_order(order.astype(np.int32))
    t = timed()
    c2 = f.step_launch_order(order.astype(np.int32))
    print("%-26s fused kernel %.1f us (%+.1f %%); correlation of the durations with the identity run's %.3f" % (name, t, 100 * (t / base - 1), np.corrcoef(cost, c2)[0, 1]))
f.step_launch_order(mode=0, want_costs=False)
print("identity again: %.1f us" % timed())
f.step_launch_order(mode=2, want_costs=False)
print("automatic (every post kernel sorts its step's durations into %d classes): %.1f us" % (32, timed()))
