#!/usr/bin/env python3
"""Debug aid (GPU box): where does a build variant's weighting phase leave the oracle?  One small rank-sort scenario, phase by phase.
(Used to find the -disable-machine-cse fault, DESIGN 8: exp() of a negative argument returned 0 -- a 64-bit literal cut to 32 bits.)
    RFS_LIB=tools/_build/librfsgpu_<name>.so python tools/nocse_probe.py [n_lm cap n_eval]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
import importlib
pkg = load_package()
sc = pkg.scenarios
ob = importlib.import_module("oracle.binding")
ob.build()
if os.environ.get("RFS_LIB"):
    pkg.engine.LIB = os.environ["RFS_LIB"]
n_lm = int(sys.argv[1]) if len(sys.argv) > 1 else 30
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 64
kw = {}
if len(sys.argv) > 3:
    kw["n_eval"] = int(sys.argv[3])
plain = os.environ.get("PROBE_PLAIN") == "1"        # the scenario's own weights (no spread, no ties)
scen = sc.make_scenario(10, n_lm, 12, seed=n_lm + cap, **kw)
if not plain:
    rng = np.random.default_rng(cap)
    w = scen["w"]
    k = n_lm // 4
    w[:, :k] = 10.0 ** rng.uniform(-200, 0, (w.shape[0], k))
    w[:, k:2 * k] = 0.4 + 1e-13 * rng.integers(0, 50, (w.shape[0], k))
    w[:, 2 * k:2 * k + 5] = 0.25
dev = pkg.RBPHDFilter(scen["n"], gm_capacity=cap)
orc = ob.OracleFilter(scen["n"])
for f in (dev, orc):
    sc.load_scenario(f, scen)
    f.update_map(scen["Z"])
bad_map = 0
for i in range(scen["n"]):
    try:
        sc.assert_gm_close(dev.export_gm(i), orc.export_gm(i), 1e-10, 1e-12, ordered=True)
    except AssertionError as e:
        bad_map += 1
        print("after update_map, particle", i, str(e)[:200].replace("\n", " "))
print("lib", os.environ.get("RFS_LIB", "shipped"), "| maps wrong after update_map:", bad_map, "| sizes", [len(dev.export_gm(i)[0]) for i in range(scen["n"])])
for f in (dev, orc):
    f.importance_weighting()
wd, wo = dev.get_weights(), orc.get_weights()
print("weights dev", wd)
print("weights orc", wo)
print("rel err   ", np.abs(wd - wo) / np.maximum(np.abs(wo), 1e-300))
for i in range(scen["n"]):
    gd, go = dev.export_gm(i), orc.export_gm(i)
    wdv, wov = np.asarray(gd[0]), np.asarray(go[0])
    if wdv.shape != wov.shape or not np.array_equal(wdv, wov):
        diff = np.nonzero(wdv != wov)[0] if wdv.shape == wov.shape else None
        print("sorted order differs, particle", i, "at ranks", None if diff is None else diff[:12], "n", wdv.shape, wov.shape)
        if diff is not None and len(diff):
            r = diff[0]
            print("   dev", wdv[max(0, r - 2):r + 4], "\n   orc", wov[max(0, r - 2):r + 4])
            print("   dev is a permutation of orc:", np.array_equal(np.sort(wdv), np.sort(wov)), "| dev sorted descending:", bool(np.all(np.diff(wdv) <= 0)))
