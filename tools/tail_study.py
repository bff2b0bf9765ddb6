#!/usr/bin/env python3
"""Which workgroups end the fused step kernel last, and why (profile build: tools/kernel_sections.py --build): per-workgroup phase clocks against
per-particle merge counters, C2a.  Tuning aid; prints correlations and the make-up of the slowest 5 % of the workgroups."""
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
n, nm, nz, cap = 2000, 200, 30, 384
scen = sc.make_scenario(n, nm, nz, seed=12345)
f = pkg.RBPHDFilter(n, gm_capacity=cap)
sc.load_scenario(f, scen)
out = (C.c_longlong * 64)()
lib.rfsgpu_debug_sections(f._h, out)
f.save_state()
m44 = np.uint64(0xfffffffffff)
runs = []
for _ in range(4):
    f.restore_state()
    f.update_async(scen["Z"])
    f.synchronize()
    pp = (C.c_longlong * (4 * n))()
    assert lib.rfsgpu_debug_per_particle_fused(f._h, pp) == 0
    raw = np.frombuffer(pp, dtype=np.int64).reshape(n, 4).copy()
    hw = (raw[:, 0].view(np.uint64) >> np.uint64(44)) & np.uint64(0xffff)
    xcc = (raw[:, 0].view(np.uint64) >> np.uint64(60)) & np.uint64(0xf)
    a = (raw.view(np.uint64) & m44).astype(np.float64) * 0.01
    pm = (C.c_longlong * (4 * n))()
    assert lib.rfsgpu_debug_per_particle(f._h, pm) == 0
    mg = np.frombuffer(pm, dtype=np.int64).reshape(n, 4).copy()
    runs.append((a, mg, hw, xcc))
a, mg, hw, xcc = runs[-1]
t0 = a[:, 0].min()
end = a[:, 3] - t0
mu, wt, me = a[:, 1] - a[:, 0], a[:, 2] - a[:, 1], a[:, 3] - a[:, 2]
p2 = mg[:, 1].astype(float)
pairs = ((mg[:, 3] >> 16) & 0xffff).astype(float)
N = (mg[:, 3] & 0xffff).astype(float)
print("kernel %.1f us; end p50 %.1f p95 %.1f p99 %.1f" % (end.max(), np.percentile(end, 50), np.percentile(end, 95), np.percentile(end, 99)))
print("corr(end, map update) %.2f  corr(end, weighting) %.2f  corr(end, merge) %.2f" % tuple(np.corrcoef(end, v)[0, 1] for v in (mu, wt, me)))
print("corr(merge us, replay cycles) %.2f  corr(merge us, listed pairs) %.2f  corr(merge us, N) %.2f" % tuple(np.corrcoef(me, v)[0, 1] for v in (p2, pairs, N)))
slow = end >= np.percentile(end, 95)
for nm_, v in (("map update", mu), ("weighting", wt), ("merge+prune", me)):
    print("  %-12s all p50 %.1f us   slowest 5%% of workgroups p50 %.1f us (+%.1f)" % (nm_, np.median(v), np.median(v[slow]), np.median(v[slow]) - np.median(v)))
print("  replay cycles: all p50 %d, slowest 5%% p50 %d; listed pairs %d vs %d" % (np.median(p2), np.median(p2[slow]), np.median(pairs), np.median(pairs[slow])))
e0 = runs[-2][0][:, 3] - runs[-2][0][:, 0].min()
print("corr(end this launch, end previous launch) %.2f  -> how much of a workgroup's lateness is the particle's own" % np.corrcoef(end, e0)[0, 1])
cu = (xcc.astype(np.int64) << 16) | ((hw.astype(np.int64) >> 8) & 0xff)
keys, inv, cnt = np.unique(cu, return_inverse=True, return_counts=True)
cu_end = np.zeros(len(keys)); np.maximum.at(cu_end, inv, end)
print("CUs: %d; workgroups per CU %s; last end per CU p50 %.1f max %.1f; by workgroups on the CU: %s" % (len(keys), dict(zip(*np.unique(cnt, return_counts=True))), np.median(cu_end), cu_end.max(),
      {int(c): round(float(np.median(cu_end[cnt == c])), 1) for c in np.unique(cnt)}))
