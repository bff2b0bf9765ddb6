#!/usr/bin/env python3
"""Phase clock of the fused step kernel per particle (a -DRFS_PROFILE build: tools/kernel_sections.py --build), C2a by
default: when each workgroup started, and how long its map update / weighting / merge+prune phases took (100 MHz ticks)."""
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
n, nm, nz, cap = [int(x) for x in (sys.argv[1:5] + [2000, 200, 30, 384][len(sys.argv[1:5]):])]
scen = sc.make_scenario(n, nm, nz, seed=12345, **({"rmax": float(os.environ["KS_RMAX"])} if "KS_RMAX" in os.environ else {}))
f = pkg.RBPHDFilter(n, gm_capacity=cap)
sc.load_scenario(f, scen)
out = (C.c_longlong * 64)()
lib.rfsgpu_debug_sections(f._h, out)
f.save_state()
for _ in range(3):
    f.restore_state()
    f.update_async(scen["Z"])
    f.synchronize()
pp = (C.c_longlong * (4 * n))()
assert lib.rfsgpu_debug_per_particle_fused(f._h, pp) == 0
raw = np.frombuffer(pp, dtype=np.int64).reshape(n, 4).copy()
hw = (raw[:, 0].view(np.uint64) >> np.uint64(44)) & np.uint64(0xffff)
xcc = (raw[:, 0].view(np.uint64) >> np.uint64(60)) & np.uint64(0xf)
raw[:, 0] = (raw[:, 0].view(np.uint64) & np.uint64(0xfffffffffff)).astype(np.int64)
raw[:, 1:] = (raw[:, 1:].view(np.uint64) & np.uint64(0xfffffffffff)).astype(np.int64)
a = raw.astype(np.float64) * 0.01   # microseconds
t0 = a[:, 0].min()
q = lambda v: "min %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (v.min(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), np.percentile(v, 99), v.max())
print("workgroup start after kernel start [us]:", q(a[:, 0] - t0))
print("map update   [us]:", q(a[:, 1] - a[:, 0]))
print("weighting    [us]:", q(a[:, 2] - a[:, 1]))
print("merge+prune  [us]:", q(a[:, 3] - a[:, 2]))
print("particle total [us]:", q(a[:, 3] - a[:, 0]))
print("end after kernel start [us]:", q(a[:, 3] - t0))
avg, cnt = f.kernel_time_stats()
print("fused kernel (events) %.1f us" % (avg[0] * 1e-3))
mu = a[:, 1] - a[:, 0]
slow = mu > 0.5 * (np.percentile(mu, 50) + mu.max())
print("slow map updates: %d of %d" % (slow.sum(), n))
for mod in (8, 32, 256):
    h = np.bincount(np.nonzero(slow)[0] % mod, minlength=mod)
    print("  slow count by particle index mod %d: min %d max %d; top residues %s" % (mod, h.min(), h.max(), np.argsort(-h)[:8].tolist()))
if os.environ.get("FS_DUMP"):
    f.restore_state(); f.update_map(scen["Z"])
    sizes = f.gm_sizes()
    print("  new Gaussians after the map update: slow mean %.1f, fast mean %.1f; corr(duration, new) %.3f" % (sizes[slow].mean() - nm, sizes[~slow].mean() - nm, np.corrcoef(mu, sizes)[0, 1]))
cu_key = (xcc.astype(np.int64) << 16) | ((hw.astype(np.int64) >> 8) & 0xff)    # XCC, SE/SH/CU bits of HW_ID
keys, inv, cnt = np.unique(cu_key, return_inverse=True, return_counts=True)
slow_per = np.bincount(inv, weights=slow.astype(float))
print("  %d distinct (XCC, SE, CU) ids; workgroups per id min %d max %d; ids with NO slow workgroup %d, ids where ALL are slow %d" %
      (len(keys), cnt.min(), cnt.max(), int((slow_per == 0).sum()), int((slow_per == cnt).sum())))
print("  slow fraction by workgroups-on-the-id:", {int(c): round(float(slow_per[cnt == c].sum() / (cnt[cnt == c].sum())), 3) for c in np.unique(cnt)})
print("  slow fraction by XCC:", [round(float(slow[xcc == x].mean()), 2) for x in range(8)])
simd = (hw.astype(np.int64) >> 4) & 3
print("  wave 0's SIMD: workgroups per SIMD %s, slow fraction per SIMD %s" % (np.bincount(simd, minlength=4).tolist(), [round(float(slow[simd == k].mean()), 2) if (simd == k).any() else None for k in range(4)]))
# per CU: how many wave-0s share a SIMD with a slow one
per = {}
for k in range(n):
    per.setdefault((int(cu_key[k]), int(simd[k])), []).append(bool(slow[k]))
import collections
hist = collections.Counter((len(v), sum(v)) for v in per.values())
print("  (wave-0s on one SIMD of a CU, slow among them): count", sorted(hist.items()))
print("  map update duration histogram [us]:", np.histogram(mu, bins=12)[0].tolist(), "edges", np.round(np.histogram(mu, bins=12)[1], 1).tolist())
wave_id = hw.astype(np.int64) & 15
print("  wave 0's hardware wave slot: slow fraction by slot", {int(k): round(float(slow[wave_id == k].mean()), 2) for k in np.unique(wave_id)}, "counts", np.bincount(wave_id).tolist())
order = np.argsort(a[:, 0])
rank_in_cu = np.zeros(n, int)
seen = {}
for k in order:
    key = int(cu_key[k]); rank_in_cu[k] = seen.get(key, 0); seen[key] = rank_in_cu[k] + 1
print("  slow fraction by arrival order on the CU:", [round(float(slow[rank_in_cu == r].mean()), 2) for r in range(rank_in_cu.max() + 1)])
end = a[:, 3] - t0
cu_end = np.array([end[inv == k].max() for k in range(len(keys))])
cu_sum = np.array([(a[inv == k, 3] - a[inv == k, 0]).sum() for k in range(len(keys))])
print("  per (XCC, CU): last workgroup ends [us]:", q(cu_end), "; sum of its workgroups' durations [us]:", q(cu_sum))
print("  corr(workgroups on the CU, CU end) %.3f; mean end by count: %s" % (np.corrcoef(cnt, cu_end)[0, 1], {int(c): round(float(cu_end[cnt == c].mean()), 1) for c in np.unique(cnt)}))
