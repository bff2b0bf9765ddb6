#!/bin/bash
# Collects the rocprofv3 evidence for a round on the GPU box:
#   1. --kernel-trace --stats   (per-kernel durations; must agree with bench.py's HIP-event times)
#   2. --pmc FETCH_SIZE         (separate pass, with --kernel-trace only -- gpurun refuses pmc + other trace domains)
#   3. --pmc WRITE_SIZE         (separate pass: FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2)
# usage: tools/profile_round.sh <tag> [bench args...]
set -u
TAG=${1:-rXX}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--steps 50 --warmup 5 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python bench.py $ARGS > $OUT/bench_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python bench.py $ARGS > $OUT/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python bench.py $ARGS > $OUT/bench_write.log 2>&1
find $OUT -name "*.csv" | head -20
grep -h '"metric"' $OUT/bench_trace.log | tail -1 | cut -c1-300
