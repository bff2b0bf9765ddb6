#!/bin/bash
# rocprofv3 evidence for a round, on the GPU box, for every workload of SURVEY 8(d):
#   c2a / c3  bench.py (its own JSON line carries the live PMC traffic of the fused kernel) + a --kernel-trace --stats pass
#   c4 / c5   bench.py --workload c4 | c5 (Victoria Park 5000 particles; Murty stress 1000 particles): the same, + two PMC passes over
#             all kernels (FETCH_SIZE, WRITE_SIZE)
# PMC passes carry --kernel-trace only (gpurun refuses pmc + other trace domains).   usage: tools/profile_workloads.sh <tag>
set -u
TAG=${1:-rXX}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run_trace() {  # name, command...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${name}_trace -o trace -- "$@" > $OUT/${name}_trace.log 2>&1
}
run_pmc() {    # name, counter, command...
  local name=$1 ctr=$2; shift 2
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/${name}_pmc_$ctr -o pmc -- "$@" > $OUT/${name}_pmc_$ctr.log 2>&1
}
python bench.py --workload c2a > $OUT/c2a_bench.log 2>&1
python bench.py --workload c3 > $OUT/c3_bench.log 2>&1
python bench.py --workload c4 > $OUT/c4_bench.log 2>&1
python bench.py --workload c5 --steps 30 --warmup 3 > $OUT/c5_bench.log 2>&1
python bench.py --workload c2b --no-cpu-baseline --no-pmc > $OUT/c2b_bench.log 2>&1
run_trace c2a python bench.py --workload c2a --steps 50 --warmup 5 --no-cpu-baseline --no-pmc
run_trace c3 python bench.py --workload c3 --steps 50 --warmup 5 --no-cpu-baseline --no-pmc
run_trace c4 python bench.py --workload c4 --steps 50 --warmup 5 --no-cpu-baseline --no-pmc
run_trace c5 python bench.py --workload c5 --steps 20 --warmup 3 --no-cpu-baseline --no-pmc
python tools/c5_bench.py --exact > $OUT/c5_exact.log 2>&1
python tools/matperm_bench.py > $OUT/matperm.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  run_pmc c4 $c python bench.py --workload c4 --steps 10 --warmup 2 --no-cpu-baseline --no-pmc
  run_pmc c5 $c python bench.py --workload c5 --steps 4 --warmup 1 --no-cpu-baseline --no-pmc
done
find $OUT -name "*.csv" | wc -l
grep -h '"metric"' $OUT/c2a_bench.log | tail -1 | cut -c1-200
