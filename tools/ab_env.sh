#!/bin/bash
# usage: tools/ab_env.sh VAR "v1 v2 ..." [bench args]   -- bench.py under VAR=v for every v, one line each
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  env $VAR=$v python bench.py --no-cpu-baseline --no-boundary --no-pmc "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', d['value'], d['ms_per_step'], {k[:22]:v.get('ms') for k,v in d['config']['kernels'].items() if isinstance(v,dict) and 'ms' in v})"
done
