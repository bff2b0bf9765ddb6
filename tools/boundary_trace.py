#!/usr/bin/env python3
"""Where one RBPHDFilter::update through the boundary spends its time (tuning aid): the rfsgpu_update_io sequence at configs[1]'s
shape, to be run under  rocprofv3 --kernel-trace --memory-copy-trace --hip-trace --output-format csv -d <dir> -- python tools/boundary_trace.py
and summarised with  python tools/boundary_trace.py --summarise <dir>  (per update: every device activity with start / end relative
to the first one, and the host API calls)."""
import csv
import glob
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    d = sys.argv[2]
    acts = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acts.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][:60]))
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acts.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Name", "copy"))[:40]))
    acts.sort()
    # cut into updates at every restore_state kernel; print the median-length one of the last 100
    cuts = [k for k, a in enumerate(acts) if "restore_state" in a[2]]
    groups = [acts[cuts[k]:cuts[k + 1]] for k in range(len(cuts) - 1)][-100:]
    groups.sort(key=lambda g: g[-1][1] - g[0][0])
    g = groups[len(groups) // 2]
    t0 = g[0][0]
    for s, e, n in g:
        print("%9.2f %9.2f  (%7.2f us)  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
    apis = {}
    for f in glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            a = apis.setdefault(r["Function"], [0, 0])
            a[0] += 1
            a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, (n, t) in sorted(apis.items(), key=lambda kv: -kv[1][1])[:14]:
        print("%-34s calls %6d  avg %8.2f us" % (k, n, t / n / 1e3))
    sys.exit(0)

from __graft_entry__ import load_package
import bench
pkg = load_package()
sc = pkg.scenarios
wl = bench.WORKLOADS["c2a"]
scen = bench.make_scen(sc, wl, wl["n"])
g = pkg.RBPHDFilter(wl["n"], gm_capacity=wl["cap"])
sc.load_scenario(g, scen)
g.save_state()
Z = scen["Z"]
x = np.ascontiguousarray(scen["poses"], dtype=np.float64)
cov = np.ascontiguousarray(np.broadcast_to(np.asarray(scen["pose_cov"], dtype=np.float64), (wl["n"], 3, 3)))
w1 = np.ones(wl["n"])
mode = os.environ.get("BT_MODE", "io")
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(300):
        g.restore_state()
        if mode == "io":
            g.update_io(Z, poses=x, pose_cov=cov, weights=w1)
        else:
            g.set_poses(x, cov); g.set_weights(w1); g.update(Z); g.get_weights()
    print("%s: %.2f us per restore + update" % (mode, (time.perf_counter() - t0) / 300 * 1e6))
