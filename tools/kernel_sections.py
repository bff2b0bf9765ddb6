#!/usr/bin/env python3
"""Kernel-section timing for tuning (not part of the product): builds a -DRFS_PROFILE copy of the library where
lane 0 of one particle stamps s_memtime at section boundaries, runs the C2a step, prints cycles per section."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
prof_lib = os.path.join(ROOT, "tools", "_build", "librfsgpu_prof.so")
if "--build" in sys.argv:
    os.makedirs(os.path.dirname(prof_lib), exist_ok=True)
    bm = pkg.build_mod
    cmd = [bm.hipcc()] + bm.FLAGS + ["-DRFS_PROFILE"] + os.environ.get("KS_FLAGS", "").split() + [os.path.join(bm.CSRC, "rfsgpu_engine.hip"), "-o", prof_lib]
    subprocess.check_call(cmd)
    sys.exit(0)
lib = C.CDLL(prof_lib)
pkg.engine._lib = lib
sc = pkg.scenarios
_pos = [a for a in sys.argv[1:] if not a.startswith("--")]
n, nm, nz, cap = [int(x) for x in (_pos[:4] + [2000, 200, 30, 384][len(_pos[:4]):])]
scen = sc.make_scenario(n, nm, nz, seed=12345, **({"rmax": float(os.environ["KS_RMAX"])} if "KS_RMAX" in os.environ else {}))
f = pkg.RBPHDFilter(n, gm_capacity=cap)
sc.load_scenario(f, scen)
out = (C.c_longlong * 64)()
lib.rfsgpu_debug_sections(f._h, out)  # allocates the stamp buffer
f.save_state()
f.set_phase_timing(True)   # (f.update below = the three stand-alone kernels)
for _ in range(3):
    f.restore_state()
    if "--fused" in sys.argv:      # the same stamps inside the fused step kernel (weighting and merge phases; particle 7)
        f.update_async(scen["Z"])
        f.synchronize()
    else:
        f.update(scen["Z"])
lib.rfsgpu_debug_sections(f._h, out)
t = np.array(list(out), dtype=np.int64)
names = {0: ["update_map: pass1", "update_map: pass2"],
         16: ["weight: sort+write", "weight: eval pts", "weight: intensity", "weight: L table", "weight: components", "weight: partitions", "weight: final"],
         32: ["merge: stage", "merge: phase1", "merge: phase2", "merge: prune"]}
for base, ns in names.items():
    for k, nm_ in enumerate(ns):
        d = t[base + k + 1] - t[base + k]
        print(f"{nm_:28s} {d:10d} cycles")
print("update_map pass0: precompute %d, gates %d, maha+lik %d, scan+write %d, fold %d" % (t[4]-t[0], t[5]-t[4], t[6]-t[5], t[7]-t[6], t[8]-t[7]))
print("weight: key load + chunk sort %d, merge ranks + scatter %d, fallback check + sorted write-out %d" % (t[16+9]-t[16], t[16+10]-t[16+9], t[16+8]-t[16+10]))
print("weight partitions: masks+components %d, log table %d, component masks/zero merge %d, enumeration %d; partitions %d" % (t[28]-t[27], t[29]-t[28], t[29]-t[29], t[30]-t[29], t[31]))
print("merge phase2: rows %d, speculative %d, validate %d, tail %d" % (t[41]-t[34], t[42]-t[41], t[43]-t[42], t[35]-t[43]))
print("merge fallbacks: unlistable %d, slack %d, >8 merges %d, claim conflicts %d of %d active rows" % (t[52], t[53], t[54], t[55], t[56]))
print("merge: grid build %d, candidate scan %d; phase2 rows %d merges %d chunks %d N %d" % (t[40]-t[33], t[34]-t[40], t[48], t[49], t[50], t[51]))
print("merge candidate scan (particle 7, 3 runs): neighbours examined per run %d; trips of four per chunk-wave iteration (max over the runs) %s" % (t[58] // 3, [int(t[k]) for k in (57, 59, 60, 61, 62, 63)]))
print("kernel ns (events):", f.last_kernel_ns())

import numpy as np
pp = (C.c_longlong * (4 * n))()
if lib.rfsgpu_debug_per_particle(f._h, pp) == 0:
    a = np.frombuffer(pp, dtype=np.int64).reshape(n, 4)
    tot, p2, fb, nn = a[:, 0], a[:, 1] & 0xffffff, a[:, 2] & 255, a[:, 3]
    walk1, rounds = (a[:, 1] >> 24) & 0xffffff, a[:, 1] >> 48
    print("merge per particle: slack-type rows mean %.2f max %d; unlistable rows mean %.2f max %d" % (((a[:, 2] >> 8) & 255).mean(), ((a[:, 2] >> 8) & 255).max(), ((a[:, 2] >> 16) & 255).mean(), ((a[:, 2] >> 16) & 255).max()))
    q = lambda v: "min %d p50 %d p90 %d p99 %d max %d" % (v.min(), np.percentile(v, 50), np.percentile(v, 90), np.percentile(v, 99), v.max())
    print("merge per particle: total cycles", q(tot))
    print("merge per particle: phase-2 cycles", q(p2))
    print("merge per particle: rounds of 64 rows", q(rounds), "; cycles in the rounds' first walks", q(walk1))
    print("merge per particle: sequential fallbacks", q(fb), "mean %.2f" % fb.mean())
    print("merge per particle: validation sub-rounds (-DMERGE_VALIDATE_ROUNDS=0: rows validated one by one)", q((a[:, 2] >> 24) & 255), "; re-walked", q((a[:, 2] >> 32) & 255), "; cycles in re-walks", q((a[:, 2] >> 40) & 0xfffff))
    print("merge per particle: N", q(nn & 0xffff))
    print("merge per particle: listed pairs", q((nn >> 16) & 0xffff))
    print("merge per particle: near-failing neighbours", q(nn >> 32))

f.restore_state()
f.update_map(scen["Z"])
f.importance_weighting()
if lib.rfsgpu_debug_per_particle(f._h, pp) == 0:
    a = np.frombuffer(pp, dtype=np.int64).reshape(n, 4)
    q = lambda v: "min %d p50 %d p90 %d p99 %d max %d" % (v.min(), np.percentile(v, 50), np.percentile(v, 90), np.percentile(v, 99), v.max())
    print("weight per particle: total cycles", q(a[:, 0]))
    print("weight per particle: components+partitions cycles", q(a[:, 1]))
    print("weight per particle: evaluation points", q(a[:, 2]))
print("sparse intensity sums, all particles, the 3 launches of this run: %d evaluation points summed by the dense loop (list overflow or prior check)" % t[8])
if t[15] > 0:
    print("sparse intensity sums (last group of the wave, particle 7): setup + sweep %d cycles, exact terms %d cycles; %d pairs listed for %d evaluation points, %d row trips, %d dense fall-backs"
          % (t[10] - t[9], t[11] - t[10], t[12], t[15], t[13], t[14]))
