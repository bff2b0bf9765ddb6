#!/usr/bin/env python3
"""Summarise a tools/profile_workloads.sh output directory into profiles/<tag>_*: per workload the rocprofv3 kernel stats
(csv, as rocprofv3 wrote it), a markdown table with the PMC traffic where collected, and the bench.py lines.
HBM traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes (rocprofv3 reports KiB; gfx950's FETCH_SIZE tallies
128-byte read requests at 64 bytes -- MI355X_MICROARCH.md, HBM section).   usage: tools/profile_summarise.py <dir> <tag>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
os.makedirs(prof, exist_ok=True)


def short(n):
    return n.split("(")[0].replace("void ", "")


def find(pattern):
    f = glob.glob(os.path.join(src, pattern), recursive=True)
    return f[0] if f else None


def pmc(name, ctr):
    f = find(f"{name}_pmc_{ctr}/**/*counter_collection.csv")
    if not f:
        return {}
    agg = collections.defaultdict(list)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if r.get("Counter_Name", ctr) == ctr:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


md = [f"# {tag}: rocprofv3 summaries per workload (tools/profile_workloads.sh {tag}; MI355X, 1 GPU)\n"]
for name, what in (("c2a", "bench.py --workload c2a (BASELINE configs[1])"), ("c3", "bench.py --workload c3 (configs[2]'s shard, 2500 x 500 x 30)"),
                   ("c4", "bench.py --workload c4 (configs[3], Victoria Park, 5000 particles)"), ("c5", "bench.py --workload c5 (configs[4], Murty stress, 1000 particles)")):
    st = find(f"{name}_trace/**/*kernel_stats.csv")
    if not st:
        md.append(f"## {name}: no trace found\n")
        continue
    shutil.copy(st, os.path.join(prof, f"{tag}_{name}_kernel_stats.csv"))
    fetch, write = pmc(name, "FETCH_SIZE"), pmc(name, "WRITE_SIZE")
    md.append(f"## {name}: {what}\n")
    md.append("| kernel | calls | avg us | min us | max us | % of GPU time" + (" | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM bytes/launch |" if fetch else " |"))
    md.append("|---|---|---|---|---|---" + ("|---|---|---|" if fetch else "|"))
    with open(st) as fh:
        for r in csv.DictReader(fh):
            k = short(r["Name"])
            if float(r["Percentage"]) < 0.5:
                continue
            line = f"| `{k[:80]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {float(r['Percentage']):.1f}"
            if fetch:
                if k in fetch:
                    b = (2 * fetch[k] + write.get(k, 0.0)) * 1024
                    line += f" | {fetch[k]:.1f} | {write.get(k, 0.0):.1f} | {b/1e6:.2f} MB |"
                else:
                    line += " | | | |"
            else:
                line += " |"
            md.append(line)
    md.append("")
    for logname in (f"{name}_bench.log", f"{name}_trace.log"):
        p = os.path.join(src, logname)
        if os.path.exists(p):
            lines = [l.strip() for l in open(p) if l.startswith("{") or l.startswith("VP ") or l.startswith("C5 ")]
            if lines:
                md.append(f"`{logname}`:\n\n```\n{lines[-1]}\n```\n")
                if logname.endswith("_bench.log") and lines[-1].startswith("{"):
                    json.dump(json.loads(lines[-1]), open(os.path.join(prof, f"{tag}_{name}_bench_line.json"), "w"), indent=1)
p = os.path.join(src, "c2b_bench.log")
if os.path.exists(p):
    lines = [l.strip() for l in open(p) if l.startswith("{")]
    if lines:
        md.append("## c2b: bench.py --workload c2b (steady state, no re-seed)\n\n```\n" + lines[-1] + "\n```\n")
        json.dump(json.loads(lines[-1]), open(os.path.join(prof, f"{tag}_c2b_bench_line.json"), "w"), indent=1)
for extra, title in (("c5_exact.log", "c5 in RFSGPU_PARTITION_EXACT mode (tools/c5_bench.py --exact)"), ("matperm.log", "batched MatPerm::calc, n = 8..20 (tools/matperm_bench.py)")):
    p = os.path.join(src, extra)
    if os.path.exists(p):
        lines = [l.rstrip() for l in open(p) if l.startswith("C5 ") or l.startswith("|")]
        if lines:
            md.append(f"## {title}\n\n" + "\n".join(lines) + "\n")
open(os.path.join(prof, f"{tag}_summary.md"), "w").write("\n".join(md) + "\n")
print("\n".join(md)[:6000])
