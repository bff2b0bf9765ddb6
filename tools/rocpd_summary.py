#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (--kernel-trace --stats) into the per-kernel stats table
that gets committed under profiles/.  Usage: tools/rocpd_summary.py <results.db> [out.md]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute(
    "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(sgpr_count), "
    "max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows) or 1
lines = ["| kernel | calls | total ms | avg us | min us | max us | % | vgpr | sgpr | lds B | scratch B | grid | wg |",
         "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
for r in rows:
    name = r[0].split("(")[0]
    lines.append(f"| `{name[:70]}` | {r[1]} | {r[2]/1e6:.3f} | {r[3]/1e3:.2f} | {r[4]/1e3:.2f} | {r[5]/1e3:.2f} | {100*r[2]/tot:.1f} | "
                 f"{r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} | {r[11]} |")
txt = "\n".join(lines)
if len(sys.argv) > 2:
    open(sys.argv[2], "a").write(txt + "\n")
print(txt)
