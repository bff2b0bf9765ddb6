# tuning aid: per-job durations (10 ns ticks) of the Murty jobs at C5 under different grid sizes (profile builds)
for v in "$@"; do echo "== $v"; RFS_LIB=tools/_build/librfsgpu_$v.so C5_STEPS=1 timeout 300 python tools/c5_bench.py 2>&1 | grep "murty job:" | python -c "
import sys,re,collections
d=collections.defaultdict(list)
for l in sys.stdin:
    m=re.search(r'n (\d+) started at tick (\d+), ended at (\d+)',l)
    if m: d[int(m.group(1))].append((int(m.group(3))-int(m.group(2)))/100.0)
for n in sorted(d): v=sorted(d[n]); print('  n %2d: %4d jobs seen, duration us: min %.0f median %.0f max %.0f'%(n,len(v),v[0],v[len(v)//2],v[-1]))
"; done
