#!/usr/bin/env python3
"""C5 Murty jobs: cycle counters printed by a -DRFS_PROFILE build (tools/kernel_sections.py --build makes it)."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
pkg = load_package()
lib = C.CDLL(os.path.join(ROOT, "tools", "_build", "librfsgpu_prof.so"))
pkg.engine._lib = lib
sc = pkg.scenarios
N = int(os.environ.get("C5_N", 1000))
scen = sc.make_scenario(N, 200, 50, seed=555, n_clutter=10, n_eval=40, weighting_md=10.0, weights=(0.8, 1.0))
f = pkg.RBPHDFilter(N, gm_capacity=448)
sc.load_scenario(f, scen)
f.set_phase_timing(True)
f.update(scen["Z"])
print("kernel ns:", f.last_kernel_ns())
