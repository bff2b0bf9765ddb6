#!/usr/bin/env python3
"""Section clock of the Victoria Park kernels for tuning (not part of the product): a -DRFS_PROFILE build in which lane 0 of
particle 7 stamps the cycle counter at section boundaries (vp.h, DBG_T).   python tools/vp_sections.py --build ; python tools/vp_sections.py"""
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
    subprocess.check_call([bm.hipcc()] + bm.FLAGS + ["-DRFS_PROFILE"] + os.environ.get("KS_FLAGS", "").split() + [os.path.join(bm.CSRC, "rfsgpu_engine.hip"), "-o", prof_lib])
    sys.exit(0)
lib = C.CDLL(prof_lib)
pkg.engine._lib = lib
sc = pkg.scenarios
N, NM, NZ = [int(os.environ.get(k, d)) for k, d in (("VP_N", 5000), ("VP_NM", 40), ("VP_NZ", 12))]
scen = sc.make_vp_scenario(N, NM, NZ, seed=4321, scan="ragged")
f = pkg.RBPHDFilter(N, gm_capacity=192, model=pkg.capi.MODEL_VICTORIAPARK_3D)
sc.load_scenario(f, scen)
out = (C.c_longlong * 64)()
lib.rfsgpu_debug_sections(f._h, out)
f.save_state()
f.set_phase_timing("--fused" not in sys.argv)
for _ in range(4):
    f.restore_state()
    f.update(scen["Z"])
lib.rfsgpu_debug_sections(f._h, out)
t = np.array(list(out), dtype=np.int64)
names = {0: ["update: Pd (wave)", "update: KF precompute", "update: gates+values", "update: survivor lists+fold", "update: emit", "update: missed-detection weights"],
         16: ["weight: rank sort (+copy)", "weight: evaluation points (Pd)", "weight: intensity sums", "weight: L table", "weight: partitions"],
         32: ["merge: stage", "merge: prefilter", "merge: scan", "merge: fence", ]}
for base, ns in names.items():
    for k, nm_ in enumerate(ns):
        print(f"{nm_:36s} {t[base + k + 1] - t[base + k]:10d} cycles")
print("merge: prune (from fence)            %10d cycles" % (t[36] - t[35]))
print("merge scan: speculative walk %d, validation (+ serial redo) %d, commit %d cycles; rows with merges %d, collisions %d, entries absorbed %d, N %d" % (t[37] - t[34], t[38] - t[37], t[35] - t[38], t[40], t[41], t[42], t[43]))
print("particle 7 end to end: update %d, weighting %d, merge %d cycles" % (t[6] - t[0], t[21] - t[16], t[36] - t[32]))
print("kernel ns (events):", f.last_kernel_ns())
