#!/bin/bash
# SQ / instruction-cache counters of murty_jobs_kernel at C5 (tuning aid).  usage: tools/pmc_c5.sh <tag> [extra bench args]   (GPU box)
set -u
TAG=${1:-x}; shift
OUT=gpurun_out/c5pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--workload c5 --steps 4 --warmup 2 --no-cpu-baseline --no-pmc $*"
k=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_ANY" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INST_LEVEL_SMEM" "SQ_INSTS_FLAT SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_WAVE32_LDS"; do
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
    if "murty_jobs" not in name:
        continue
    print(name)
    for c, v in sorted(cs.items()):
        print("   %-28s mean %16.0f  max %16.0f  (n %d)" % (c, sum(v) / len(v), max(v), len(v)))
PY
