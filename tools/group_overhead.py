#!/usr/bin/env python3
"""What the one collective of the path costs a step where it can be measured on a 1-GPU box: configs[1]'s shape on (a) a plain handle
(step + division in the post kernel), (b) an rfsgpu_group of one shard over RCCL in the round-4 order (update, all-reduce, divide
kernel, all on the shard's stream), (c) the same group with the normalisation trailing by one step (rfsgpu_group_update_deferred:
the all-reduce on a side stream beside the next step's kernel).  State re-seeded every step in all three."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
import bench
pkg = load_package()
sc = pkg.scenarios
wl = bench.WORKLOADS["c2a"]
scen = bench.make_scen(sc, wl, wl["n"])
Z = scen["Z"]
S = int(os.environ.get("GO_STEPS", 400))


def timed(fn, sync):
    for _ in range(200):
        fn()
    sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(S):
            fn()
        sync()
        best = min(best, (time.perf_counter() - t0) / S * 1e6)
    return best


f = pkg.RBPHDFilter(wl["n"], gm_capacity=wl["cap"])
sc.load_scenario(f, scen)
f.save_state()
ta = timed(lambda: (f.restore_state(), f.step_async(Z, True)), f.synchronize)
f.close()
g = pkg.FilterGroup(wl["n"], [0], gm_capacity=wl["cap"])
assert g.collective() == "rccl", g.collective()
sc.load_scenario(g, scen)
s0 = g.shards[0]
s0._call("save_state")
tb = timed(lambda: (s0._call("restore_state"), g.update_nosums(Z), g.normalize()), g.synchronize)
tc = timed(lambda: (s0._call("restore_state"), g.update_deferred(Z)), g.synchronize)
print("plain handle %.2f us/step | group over RCCL, collective + divide on the step's stream %.2f (+%.2f) | normalisation trailing by one step %.2f (+%.2f)" %
      (ta, tb, tb - ta, tc, tc - ta))
g.close()
