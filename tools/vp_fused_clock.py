#!/usr/bin/env python3
"""Per-particle clock of the fused Victoria Park step (a -DRFS_PROFILE build: tools/vp_sections.py --build): when each wave started
and ended, how long its phases took, how many waves were running over time -- the tail analysis behind DESIGN.md section 8 (C4)."""
import ctypes as C
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
for _ in range(4):
    f.restore_state()
    f.update_async(scen["Z"])
    f.synchronize()
pp = (C.c_longlong * (4 * N))()
assert lib.rfsgpu_debug_per_particle_fused(f._h, pp) == 0
raw = np.frombuffer(pp, dtype=np.int64).reshape(N, 4).copy()
m44 = np.uint64(0xfffffffffff)
a = (raw.view(np.uint64) & m44).astype(np.float64) * 0.01      # microseconds
t0 = a[:, 0].min()
q = lambda v: "min %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (v.min(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), np.percentile(v, 99), v.max())
print("wave start after kernel start [us]:", q(a[:, 0] - t0))
print("map update  [us]:", q(a[:, 1] - a[:, 0]))
print("weighting   [us]:", q(a[:, 2] - a[:, 1]))
print("merge+prune [us]:", q(a[:, 3] - a[:, 2]))
dur = a[:, 3] - a[:, 0]
print("particle total [us]:", q(dur), " sum over particles %.0f us" % dur.sum())
end = a[:, 3] - t0
print("end after kernel start [us]:", q(end))
T = end.max()
grid = np.linspace(0, T, 41)
running = [(int(((a[:, 0] - t0 <= g) & (end > g)).sum())) for g in grid]
print("waves running at 40 points of the launch (3072 slots at 12 waves per CU):", running)
print("kernel = %.1f us; sum of wave durations / 3072 slots = %.1f us (a launch without tail or start-up would take that)" % (T, dur.sum() / 3072.0))
late = a[:, 0] - t0 > 1.0
print("waves of the first round %d (duration p50 %.1f), later waves %d (duration p50 %.1f)" % ((~late).sum(), np.median(dur[~late]), late.sum(), np.median(dur[late]) if late.any() else 0))
sizes = f.gm_sizes()
print("corr(duration, mixture size after the step) %.3f" % np.corrcoef(dur, sizes)[0, 1])
avg, cnt = f.kernel_time_stats()
print("fused kernel (events) %.1f us" % (avg[0] * 1e-3))
