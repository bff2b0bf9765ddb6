#!/bin/bash
# SQ counters of the fused Victoria Park step (tuning aid): occupancy, VALU busy, waits.   usage: bash tools/vp_pmc.sh <tag>
TAG=${1:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/vp_sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
k=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES"; do
  k=$((k+1))
  VP_STEPS=10 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$k -o p -- python $GRAFT_REPO_ROOT/tools/vp_bench.py > $OUT/log$k.txt 2>&1
done
python - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in agg.items():
    if "vp_step_fused" not in name:
        continue
    print(name)
    for c, v in sorted(cs.items()):
        print("   %-28s %16.0f" % (c, sum(v) / len(v)))
PY
