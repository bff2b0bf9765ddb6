#!/bin/bash
# tuning aid: bench library variants (tools/variant_bench.py --build ...) on every workload, two alternating passes.  usage: tools/flag_sweep.sh <variant> ...  (GPU box)
for w in c2a c2b c3 c4 c5; do
  S=300; [ $w = c5 ] && S=40
  for r in 1 2; do
    VB_WORKLOAD=$w VB_STEPS=$S python tools/variant_bench.py --run "$@" 2>&1 | awk -v w=$w '{print w, $1, $2, $3, $4, $5}' | cut -c1-100
  done
done
