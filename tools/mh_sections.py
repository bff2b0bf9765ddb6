#!/usr/bin/env python3
"""MH-FastSLAM association kernel: per-particle cycle counters of a -DRFS_PROFILE build (tools/kernel_sections.py --build
makes it).  FS_N / FS_NM / FS_NZ / FS_HYP as in tools/fastslam_bench.py."""
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
lib.rfsgpu_debug_sections(f._h, out)  # allocates the stamp buffer
f.save_state()
for _ in range(2):
    f.restore_state()
    f.fastslam_update(scen["Z"])
pp = (C.c_longlong * (4 * f.n))()
f.restore_state()
assert lib.rfsgpu_debug_per_particle(f._h, pp) == 0
a = np.frombuffer(pp, dtype=np.int64).reshape(-1, 4)[:N]
q = lambda v: "min %d p50 %d p90 %d max %d" % (v.min(), np.percentile(v, 50), np.percentile(v, 90), v.max())
print("murty total cycles   ", q(a[:, 0]))
print("child solver cycles  ", q(a[:, 1]))
print("children solved      ", q(a[:, 2]))
print("reduced dimension    ", q(a[:, 3]))
print("cycles per child solve", q(a[:, 1] // np.maximum(a[:, 2], 1)))
print("kernel ns:", f.last_kernel_ns())
