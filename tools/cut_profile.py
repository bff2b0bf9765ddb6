#!/usr/bin/env python3
"""Tuning aid (not part of the product): cumulative instruction counts along the fused step kernel.

  python tools/cut_profile.py --build          here: cross-compiles one -DRFS_STOP_AT=k library per cut point (tools/_build/)
  python tools/cut_profile.py --run [wl]       GPU box: each library under rocprofv3 --pmc (SQ_INSTS_VALU/SALU/LDS, SQ_WAVE_CYCLES) + kernel trace
  python tools/cut_profile.py --child lib wl   (internal) a few fused steps of workload wl (c2a | c3) through library lib
"""
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "_build")
CUTS = [(1, "map update, pass 0: KF quantities + Pd"), (2, "  + packed-fp32 gate sweep"), (3, "  + exact gates / likelihood of the candidates"),
        (4, "  + scan, list write, new means/covariances"), (5, "  + normaliser fold (end of pass 0)"), (6, "map update phase 1 (all passes)"),
        (7, "map update phase 2"), (8, "map update done"), (10, "weighting: chunk rank sort"), (11, "  + cross-chunk ranks"), (12, "  + permutation out"),
        (13, "  + evaluation points, weight sums"), (14, "  + likelihood table (wave 0 alone; no intensity)"),
        (15, "  + partitions (wave 0 alone; no intensity)"), (16, "weighting done (with the intensity pass)"), (20, "merge: stage"),
        (21, "  + grid build"), (22, "  + candidate scan (1a)"), (23, "  + exact pair tests (1b)"), (24, "  + phase 2 (replay)"), (99, "whole kernel")]


def lib_of(k):
    return os.path.join(OUT, f"librfsgpu_cut{k}.so")


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
    import bench
    wl = bench.WORKLOADS[sys.argv[3]]
    sc = pkg.scenarios
    kw = {"rmax": wl["rmax"]} if wl.get("rmax") else {}
    scen = sc.make_scenario(wl["n"], wl["nm"], 30, seed=12345, **kw)
    f = pkg.RBPHDFilter(wl["n"], gm_capacity=wl["cap"])
    sc.load_scenario(f, scen)
    f.save_state()
    for _ in range(6):
        f.restore_state()
        f.update_async(scen["Z"])
    f.synchronize()
else:
    wl = sys.argv[2] if len(sys.argv) > 2 else "c2a"
    os.chdir("/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    rows = []
    for k, label in CUTS:
        if not os.path.exists(lib_of(k)):      # (a subset of the cut libraries was built)
            continue
        d = f"/tmp/cutprof/{k}"
        agg = {}
        for n, ctrs in enumerate((["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES"], ["SQ_ACTIVE_INST_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_BUSY_CYCLES"])):
            subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + ctrs + ["--output-format", "csv", "-d", f"{d}_{n}", "-o", "p", "--",
                            sys.executable, os.path.join(ROOT, "tools", "cut_profile.py"), "--child", lib_of(k), wl],
                           env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
            for fcsv in glob.glob(f"{d}_{n}/**/p_counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(fcsv)):
                    if "step_fused" in r["Kernel_Name"]:
                        agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            if n == 0:
                dur = []
                for fcsv in glob.glob(f"{d}_{n}/**/p_kernel_trace.csv", recursive=True):
                    for r in csv.DictReader(open(fcsv)):
                        if "step_fused" in r["Kernel_Name"]:
                            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
        m = {c: sum(v[1:]) / max(1, len(v) - 1) for c, v in agg.items() if len(v) > 1}
        nw = bench_waves = None
        rows.append((k, label, m, (sum(dur[1:]) / max(1, len(dur) - 1)) if len(dur) > 1 else float("nan")))
        print(k, label, {c: round(v) for c, v in m.items()}, "us(pmc run) %.1f" % rows[-1][3], flush=True)
    print("\ncut | cumulative VALU | delta VALU | cumulative SALU | LDS | kernel us under the counters")
    prev = 0.0
    for k, label, m, us in rows:
        v = m.get("SQ_INSTS_VALU", float("nan"))
        print("%3d %-58s %12.0f %+12.0f %12.0f %10.0f %8.1f" % (k, label, v, v - prev, m.get("SQ_INSTS_SALU", float("nan")), m.get("SQ_INSTS_LDS", float("nan")), us))
        if k not in (14, 15):
            prev = v if k not in (1, 2, 3, 4, 5) or True else prev
