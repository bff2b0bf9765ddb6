#!/usr/bin/env python3
"""Test aid (GPU box): every warm-started child solve of the Murty search against the solve from scratch.

A library built with -DMURTY_WARM_CHECK=1 (tools/variant_bench.py --build warmcheck="-DMURTY_WARM_CHECK=1") runs BOTH solvers on
every child of the RB-PHD partition sums' small form and compares what their assignments are worth on the child's own table; the
kernel prints a running total and the first mismatches.  This drives it with configs[4]'s scene (C5_N particles) and a few random
ones, compares the particle weights with the oracle (1e-9, the parity tests' tolerance) and reports the totals.
    RFS_LIB=tools/_build/librfsgpu_warmcheck.so python tools/murty_warm_check.py"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
import importlib
pkg = load_package()
if os.environ.get("RFS_LIB"):
    pkg.engine.LIB = os.environ["RFS_LIB"]
sc = pkg.scenarios
ob = importlib.import_module("oracle.binding")
rng = np.random.default_rng(int(os.environ.get("SEED", 7)))
cases = [dict(n=int(os.environ.get("C5_N", 96)), nl=200, nz=50, seed=555, kw=dict(n_clutter=10, n_eval=40, weighting_md=10.0, weights=(0.8, 1.0)))]
for _ in range(int(os.environ.get("CASES", 6))):
    cases.append(dict(n=int(rng.integers(4, 24)), nl=int(rng.choice([60, 120, 200])), nz=int(rng.integers(20, 50)), seed=int(rng.integers(1 << 30)),
                      kw=dict(n_clutter=int(rng.integers(2, 12)), n_eval=int(rng.choice([15, 30, 40])), weighting_md=float(rng.choice([6.0, 8.0, 10.0])), weights=(0.8, 1.0))))
bad = 0
for c in cases:
    scen = sc.make_scenario(c["n"], c["nl"], c["nz"], seed=c["seed"], **c["kw"])
    dev = pkg.RBPHDFilter(c["n"], gm_capacity=448)
    orc = ob.OracleFilter(c["n"])
    for f in (dev, orc):
        sc.load_scenario(f, scen)
        f.update(scen["Z"])
    wd, wo = dev.get_weights(), orc.get_weights()
    rel = float(np.max(np.abs(wd / wd.sum() - wo / wo.sum()) / (wo / wo.sum())))
    ok = rel < 1e-9 and np.array_equal(dev.gm_sizes(), orc.gm_sizes())
    bad += 0 if ok else 1
    print("case", c, "murty calls (oracle)", orc.murty_calls(), "max relative weight difference %.3g" % rel, "OK" if ok else "MISMATCH", flush=True)
    dev.close()
print("murty_warm_check: %d cases, %d beyond 1e-9" % (len(cases), bad))
sys.exit(1 if bad else 0)
