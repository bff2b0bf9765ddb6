#!/usr/bin/env python3
"""Tuning / test aid: which extended dimensions do the Murty jobs of a scenario have?  (profile build: RFS_LIB=tools/_build/librfsgpu_op.so)
   python tools/murty_dims.py n_particles n_landmarks n_z n_eval gate seed"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
if os.environ.get("RFS_LIB"):
    pkg.engine.LIB = os.environ["RFS_LIB"]
sc = pkg.scenarios
n, nl, nz, ne, gate, seed = [float(x) for x in sys.argv[1:7]]
scen = sc.make_scenario(int(n), int(nl), int(nz), seed=int(seed), n_clutter=int(os.environ.get("NCL", 10)), n_eval=int(ne), weighting_md=gate, weights=(0.8, 1.0),
                        **({"rmax": float(os.environ["RMAX"])} if "RMAX" in os.environ else {}))
f = pkg.RBPHDFilter(int(n), gm_capacity=768)
sc.load_scenario(f, scen)
f.update(scen["Z"])
print("weights finite:", bool(__import__("numpy").all(__import__("numpy").isfinite(f.get_weights()))))
