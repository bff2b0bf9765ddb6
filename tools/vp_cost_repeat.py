#!/usr/bin/env python3
"""Is a particle's duration in the fused Victoria Park step a property of the PARTICLE (repeats launch after launch on the same state)
or of where / when its wave ran?  (-DRFS_PROFILE build: tools/vp_sections.py --build.)  Prints the correlation of the per-particle
durations of consecutive launches on the re-seeded state and what list scheduling on 3072 slots would make of the measured durations
in slot order and longest-first -- the question behind a cost-ordered launch."""
import ctypes as C
import heapq
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
pkg = load_package()
lib = C.CDLL(os.path.join(ROOT, "tools", "_build", "librfsgpu_prof.so"))
pkg.engine._lib = lib
sc = pkg.scenarios
N, NM, NZ = [int(os.environ.get(k, d)) for k, d in (("VP_N", 5000), ("VP_NM", 40), ("VP_NZ", 12))]
scen = sc.make_vp_scenario(N, NM, NZ, seed=4321, scan="ragged")
f = pkg.RBPHDFilter(N, gm_capacity=192, model=pkg.capi.MODEL_VICTORIAPARK_3D)
sc.load_scenario(f, scen)
out = (C.c_longlong * 64)()
lib.rfsgpu_debug_sections(f._h, out)
f.save_state()
m44 = np.uint64(0xfffffffffff)


def one():
    f.restore_state()
    f.update_async(scen["Z"])
    f.synchronize()
    pp = (C.c_longlong * (4 * N))()
    assert lib.rfsgpu_debug_per_particle_fused(f._h, pp) == 0
    raw = np.frombuffer(pp, dtype=np.int64).reshape(N, 4).copy()
    a = (raw.view(np.uint64) & m44).astype(np.float64) * 0.01
    return a


def makespan(d, slots=3072):
    h = [0.0] * slots
    for x in d:
        t = heapq.heappop(h)
        heapq.heappush(h, t + x)
    return max(h)


for _ in range(3):
    one()
runs = [one() for _ in range(4)]
durs = [a[:, 3] - a[:, 0] for a in runs]
for k in range(1, len(durs)):
    print("corr(duration launch %d, launch %d) = %.3f" % (k - 1, k, np.corrcoef(durs[k - 1], durs[k])[0, 1]))
for name, sl in (("map update", (0, 1)), ("weighting", (1, 2)), ("merge+prune", (2, 3))):
    x, y = runs[0][:, sl[1]] - runs[0][:, sl[0]], runs[1][:, sl[1]] - runs[1][:, sl[0]]
    print("  %-12s corr %.3f   p50 %.1f us" % (name, np.corrcoef(x, y)[0, 1], np.median(x)))
d0, d1 = durs[0], durs[1]
print("kernel (launch 1) %.1f us" % ((runs[1][:, 3]).max() - runs[1][:, 0].min()))
print("list scheduling of launch 1's durations on 3072 slots: slot order %.1f us, longest-first by launch 0's durations %.1f us, by its own %.1f us, shortest-first %.1f us"
      % (makespan(d1), makespan(d1[np.argsort(-d0)]), makespan(d1[np.argsort(-d1)]), makespan(d1[np.argsort(d0)])))
sizes = f.gm_sizes()
print("corr(duration, mixture size after the step) %.3f" % np.corrcoef(d1, sizes)[0, 1])

rng = np.random.default_rng(0)
for nclass in (2, 3, 4, 8):
    for qs in ([0.385], [0.3], [0.5]) if nclass == 2 else ([None],):
        if nclass == 2:
            edges = np.quantile(d0, qs)
        else:
            edges = np.quantile(d0, np.linspace(0, 1, nclass + 1)[1:-1])
        cls = np.searchsorted(edges, d0)            # 0 = shortest class
        key = -cls + 1e-3 * rng.random(N)           # longest class first, arrival order inside a class
        order = np.argsort(key, kind="stable")
        print("  %d classes by launch 0's durations (edges at %s us): launch 1 would take %.1f us" % (nclass, np.round(edges, 1).tolist(), makespan(d1[order])))
