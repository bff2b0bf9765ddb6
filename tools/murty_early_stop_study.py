"""Would an early stop of the Murty-200 loop pay at configs[4]?  (measure before building.)

The caller adds exp(score) over the ranked assignments (include/RBPHDFilter.hpp:948-959); scores come out non-increasing, so
once exp(s_k) is below 2^-56 of the running sum every later addition rounds to no change (half an ulp of a sum in
[2^e, 2^(e+1)) is at least 2^-54 of it; the factor 4 covers the solver's 1e-12 tolerances).  The oracle's study hook
(RFSOR_MURTY_STUDY) records, per partition of the real C5 jobs, the call at which that happens.  CPU only."""
import argparse
import collections
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=int, default=128)
    ap.add_argument("--seed-offset", type=int, default=0)
    args = ap.parse_args()
    out = tempfile.mktemp(prefix="murty_study_")
    os.environ["RFSOR_MURTY_STUDY"] = out
    from __graft_entry__ import load_package
    import bench
    from oracle import binding as ob
    pkg = load_package()
    sc = pkg.scenarios
    wl = bench.WORKLOADS["c5"]
    scen = bench.make_scen(sc, wl, args.particles, args.seed_offset)
    f = ob.OracleFilter(args.particles)
    sc.load_scenario(f, scen)
    f.update(scen["Z"])
    rows = np.loadtxt(out).reshape(-1, 6)
    os.unlink(out)
    by = collections.defaultdict(list)
    for n, nr, nc, kstop, ktot, inc in rows:
        by[int(n)].append((int(kstop), int(ktot), inc))
    print(f"{len(rows)} Murty partitions over {args.particles} particles ({len(rows) / args.particles:.2f} per particle); largest score increase between two calls: {rows[:, 5].max():.2e}")
    print(" dim   jobs  calls(ref) mean   stop: share of jobs | calls with stop (mean, median, p90)   calls saved")
    tot_ref = tot_new = 0
    for n in sorted(by):
        a = np.array(by[n], dtype=float)
        ks, kt = a[:, 0], a[:, 1]
        eff = np.where(ks > 0, ks, kt)
        tot_ref += kt.sum() * 1.0
        tot_new += eff.sum()
        print(f" {n:3d}  {len(a):5d}   {kt.mean():7.1f}        {np.mean(ks > 0):5.2f}               {eff.mean():6.1f} {np.median(eff):6.0f} {np.percentile(eff, 90):6.0f}     {1 - eff.sum() / kt.sum():5.2f}")
    print(f"all: calls {tot_ref:.0f} -> {tot_new:.0f} ({1 - tot_new / tot_ref:.2%} saved)")
