#!/usr/bin/env python3
"""Tuning aid (not part of the product): cumulative instruction counts along the fused Victoria Park step kernel.

  python tools/vp_cut_profile.py --build      here: one -DRFS_STOP_AT=k library per cut point (tools/_build/)
  python tools/vp_cut_profile.py --run        GPU box: each library under rocprofv3 --pmc (SQ_INSTS_VALU/SALU/LDS, SQ_WAVE_CYCLES) + kernel trace
  python tools/vp_cut_profile.py --child lib  (internal) a few fused updates of the configs[3] workload through library lib
"""
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "_build")
CUTS = [(101, "map update: Pd of the landmarks (+ shifted copies)"), (102, "  + KF quantities"), (103, "  + gates / values"), (104, "  + survivor lists, normaliser fold"),
        (105, "  + emit (new Gaussians)"), (106, "map update done"), (110, "weighting: rank sort"), (111, "  + evaluation points (Pd)"), (112, "  + intensity sums"),
        (113, "  + likelihood table"), (114, "weighting done (+ partitions)"), (120, "merge: stage (+ row list)"), (121, "  + scan"), (999, "whole kernel (+ prune)")]


def lib_of(k):
    return os.path.join(OUT, f"librfsgpu_vpcut{k}.so")


if sys.argv[1] == "--build":
    import __graft_entry__ as g
    bm = g.load_package().build_mod
    os.makedirs(OUT, exist_ok=True)
    procs = []
    for k, _ in CUTS:
        cmd = [bm.hipcc()] + bm.FLAGS + [f"-DRFS_STOP_AT={k}"] + os.environ.get("CUT_FLAGS", "").split() + [os.path.join(bm.CSRC, "rfsgpu_engine.hip"), "-o", lib_of(k)]
        procs.append((k, subprocess.Popen(cmd, stderr=subprocess.DEVNULL)))
    for k, p in procs:
        print(k, "rc", p.wait())
elif sys.argv[1] == "--child":
    import __graft_entry__ as g
    pkg = g.load_package()
    pkg.engine.LIB = sys.argv[2]
    sc = pkg.scenarios
    scen = sc.make_vp_scenario(5000, 40, 12, seed=4321, scan="ragged")
    f = pkg.RBPHDFilter(5000, gm_capacity=192, model=pkg.capi.MODEL_VICTORIAPARK_3D)
    sc.load_scenario(f, scen)
    f.save_state()
    for _ in range(6):
        f.restore_state()
        f.update_async(scen["Z"])
    f.synchronize()
else:
    os.chdir("/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    rows = []
    for k, label in CUTS:
        d = f"/tmp/vpcut/{k}"
        agg, dur = {}, []
        subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "--output-format", "csv", "-d", d, "-o", "p", "--",
                        sys.executable, os.path.join(ROOT, "tools", "vp_cut_profile.py"), "--child", lib_of(k)],
                       env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
        for fcsv in glob.glob(f"{d}/**/p_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(fcsv)):
                if "vp_step_fused" in r["Kernel_Name"]:
                    agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for fcsv in glob.glob(f"{d}/**/p_kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(fcsv)):
                if "vp_step_fused" in r["Kernel_Name"]:
                    dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
        m = {c: sum(v[1:]) / max(1, len(v) - 1) for c, v in agg.items() if len(v) > 1}
        rows.append((k, label, m, (sum(dur[1:]) / max(1, len(dur) - 1)) if len(dur) > 1 else float("nan")))
    print("cut | cumulative VALU | delta VALU | cumulative SALU | LDS | kernel us under the counters   (wave-instructions per launch, 5000 waves)")
    prev = 0.0
    for k, label, m, us in rows:
        v = m.get("SQ_INSTS_VALU", float("nan"))
        print("%4d %-58s %12.0f %+12.0f %12.0f %10.0f %8.1f" % (k, label, v, v - prev, m.get("SQ_INSTS_SALU", float("nan")), m.get("SQ_INSTS_LDS", float("nan")), us))
        prev = v
