#!/bin/bash
# SQ instruction-mix counters for the hot kernels (tuning aid): are the kernels issue-bound or latency-bound?
# usage: tools/pmc_sq.sh <tag>      (GPU box; writes gpurun_out/sq_<tag>/)
set -u
TAG=${1:-x}
OUT=gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--steps 20 --warmup 3 --no-cpu-baseline --no-pmc"
k=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_ANY"; do
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
        print("   %-24s %14.0f" % (c, sum(v) / len(v)))
PY
