#!/bin/bash
# WRITE_SIZE / FETCH_SIZE per launch of murty_jobs_kernel at C5 for library variants (tools/variant_bench.py --build <names>).  usage: tools/c5_write_size.sh <variant> ...   (GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  for c in WRITE_SIZE FETCH_SIZE; do
    rm -rf /tmp/c5ws; RFS_LIB=tools/_build/librfsgpu_$v.so C5_STEPS=4 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/c5ws -o p -- python tools/c5_bench.py > /tmp/c5ws.log 2>&1
    python - "$v" "$c" <<'PY'
import csv, glob, sys
vals = {}
for f in glob.glob("/tmp/c5ws/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "murty_jobs" in r["Kernel_Name"]:
            vals.setdefault(r["Kernel_Name"].split("(")[0][-28:], []).append(float(r["Counter_Value"]))
for k, v in vals.items():
    v = sorted(v)
    print(sys.argv[1], sys.argv[2], k, "launches", len(v), "median KiB %.0f" % v[len(v) // 2])
PY
  done
  RFS_LIB=tools/_build/librfsgpu_$v.so python tools/c5_bench.py 2>&1 | grep "C5 RB"
done
