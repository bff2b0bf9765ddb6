#!/bin/bash
# A/B of library variants on the Victoria Park workload (tools/variant_bench.py --build name=flags ...; then: bash tools/vp_ab.sh name1 name2 ...)
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  for v in "$@"; do
    echo -n "$v: "; RFS_LIB=tools/_build/librfsgpu_$v.so python tools/vp_bench.py 2>/dev/null | sed 's/VP RB-PHD update, //'
  done
done
