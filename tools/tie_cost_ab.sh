#!/bin/bash
# What the std::sort tie-order correction (csrc/stdsort_replay.h) costs where mixtures DO hold equal weights: the steady-state
# workload C2b (predict with births + update, not re-seeded: every mixture carries a run of tied birth weights) with the shipped
# library against a build with -DSS_TIE_ORDER_OFF=1 (ties by index: NOT the reference's order; tools/variant_bench.py --build
# base="" tieoff="-DSS_TIE_ORDER_OFF=1").   usage: bash tools/tie_cost_ab.sh [variants...]
cd "$(dirname "$0")/.."
for v in ${@:-base tieoff base tieoff}; do python -c "
import sys; sys.path.insert(0, '.'); import __graft_entry__ as g; pkg = g.load_package(); pkg.engine.LIB = 'tools/_build/librfsgpu_$v.so'; import bench
sys.argv = ['bench.py', '--workload', 'c2b', '--steps', '300', '--warmup', '20', '--no-cpu-baseline', '--no-pmc', '--no-boundary']; bench.main()" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], {k:v.get('ms') for k,v in d['config']['kernels'].items() if isinstance(v,dict) and 'ms' in v})"; done
