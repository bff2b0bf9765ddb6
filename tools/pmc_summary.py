#!/usr/bin/env python3
"""Summarise a tools/profile_round.sh output directory into profiles/<tag>_*.{md,csv,json}.
HBM traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes: rocprofv3 reports both counters in KiB, and on
gfx950 FETCH_SIZE tallies 128-B read requests at 64 B (MI355X_MICROARCH.md, HBM section) -- confirmed here on
restore_state_kernel, a pure copy of known size (read == written bytes), whose FETCH_SIZE is 0.53 x its WRITE_SIZE."""
import collections
import csv
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
os.makedirs(prof, exist_ok=True)
shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(prof, f"{tag}_kernel_stats.csv"))


def short(n):
    n = n.split("(")[0]
    return n.replace("void ", "")


def pmc(which):
    agg = collections.defaultdict(list)
    with open(os.path.join(src, f"pmc_{which}", f"{which}_counter_collection.csv")) as fh:
        for r in csv.DictReader(fh):
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


fetch, write = pmc("fetch"), pmc("write")
dur = {}
with open(os.path.join(src, "trace", "trace_kernel_stats.csv")) as fh:
    for r in csv.DictReader(fh):
        dur[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]))
out = {}
lines = ["| kernel | calls | avg us | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM bytes/launch (2*F+W)*1024 | GB/s |", "|---|---|---|---|---|---|---|"]
for k, (calls, avg) in sorted(dur.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    if k not in fetch or calls < 10:
        continue
    b = (2 * fetch[k] + write.get(k, 0.0)) * 1024
    out[k] = dict(calls=calls, avg_ns=avg, fetch_kib=fetch[k], write_kib=write.get(k, 0.0), hbm_bytes=b)
    lines.append(f"| `{k}` | {calls} | {avg/1e3:.2f} | {fetch[k]:.1f} | {write.get(k,0):.1f} | {b/1e6:.2f} MB | {b/avg:.1f} |")
json.dump(out, open(os.path.join(prof, f"{tag}_pmc.json"), "w"), indent=1)
# (round 1 also wrote profiles/pmc_latest.json for bench.py to read; bench.py measures the traffic live since round 2)
bench = [l for l in open(os.path.join(src, "bench_trace.log")) if '"metric"' in l]
with open(os.path.join(prof, f"{tag}_summary.md"), "w") as fh:
    fh.write(f"# {tag}: rocprofv3 summary (tools/profile_round.sh {tag}; MI355X, 1 GPU)\n\n")
    fh.write("Three separate passes of `python bench.py --steps 50 --warmup 5 --no-cpu-baseline`: `--kernel-trace --stats`, "
             "`--kernel-trace --pmc FETCH_SIZE`, `--kernel-trace --pmc WRITE_SIZE`.\n\n")
    fh.write("\n".join(lines) + "\n\n")
    if bench:
        fh.write("bench.py line of the traced run:\n\n```\n" + bench[-1].strip() + "\n```\n")
print("\n".join(lines))
