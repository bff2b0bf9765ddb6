#!/usr/bin/env python3
"""MH-FastSLAM association kernel: the block-form Murty's cycle split printed by a -DRFS_PROFILE build
(tools/kernel_sections.py --build makes it).  FS_N / FS_NM / FS_NZ / FS_HYP as in tools/fastslam_bench.py."""
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
N, NM, NZ, HYP = [int(os.environ.get(k, d)) for k, d in (("FS_N", 2000), ("FS_NM", 50), ("FS_NZ", 30), ("FS_HYP", 3))]
scen = sc.make_scenario(N, NM, NZ, seed=4242, rmax=25.0)
f = pkg.FastSLAM(N, gm_capacity=384, max_hypotheses=HYP)
sc.load_scenario(f, scen)
for i in range(N):
    f.import_gm(i, np.zeros(NM), scen["mean"][i], scen["cov"][i])
f.set_fastslam_config(f.fs_config)
out = (C.c_longlong * 64)()
lib.rfsgpu_debug_sections(f._h, out)   # allocates the stamp buffer
f.save_state()
f.fastslam_update(scen["Z"])
f.restore_state()
f.fastslam_update(scen["Z"])
print("kernel ns:", f.last_kernel_ns())
f.restore_state()
pp = (C.c_longlong * (4 * N))()
assert lib.rfsgpu_debug_per_particle_fused(f._h, pp) == 0
a = np.frombuffer(pp, dtype=np.int64).reshape(N, 4).astype(np.float64)
t0 = a[:, 0].min()
q = lambda v: "min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % (v.min(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), v.max())
print("start after kernel start [us]:", q((a[:, 0] - t0) * 0.01))
print("table + reduce [us]:", q((a[:, 1] - a[:, 0]) * 0.01))
print("Murty          [us]:", q((a[:, 2] - a[:, 1]) * 0.01))
print("Murty done after kernel start [us]:", q((a[:, 2] - t0) * 0.01))
print("reduced dimension:", q(a[:, 3]))
