#!/bin/bash
# Instruction-cache counters of the hot kernels (tuning aid).  usage: tools/pmc_icache.sh <tag>   (GPU box)
set -u
TAG=${1:-x}
OUT=gpurun_out/ic_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--steps 20 --warmup 3 --no-cpu-baseline --no-pmc --no-boundary"
k=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQC_TC_INST_REQ SQC_ICACHE_BUSY_CYCLES" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAVE_CYCLES"; do
  k=$((k+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$k -o p -- python bench.py $ARGS > $OUT/log$k.txt 2>&1
done
python - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in agg.items():
    if not any(s in name for s in ("update_map", "weight_multi", "gm_merge", "step_fused")):
        continue
    print(name)
    for c, v in sorted(cs.items()):
        print("   %-28s %14.0f" % (c, sum(v) / len(v)))
PY
