# tuning aid: when each sampled Murty job starts / ends (us from the first start), by block, for profile builds
for v in "$@"; do echo "== $v"; RFS_LIB=tools/_build/librfsgpu_$v.so C5_STEPS=1 timeout 300 python tools/c5_bench.py 2>&1 | grep "murty job:\|murty tail" | python -c "
import sys,re
rows=[]; tail=None
for l in sys.stdin:
    m=re.search(r'block (\d+) n (\d+) started at tick (\d+), ended at (\d+)',l)
    if m: rows.append(tuple(int(x) for x in m.groups()))
    m=re.search(r'murty tail.*ends at tick (\d+)',l)
    if m: tail=int(m.group(1))
# the bench runs warm-up updates too: keep the last launch (largest ticks): split by gaps
rows.sort(key=lambda r:r[2])
t0=rows[-1][2]
# find the last cluster: walk back while gap < 20 ms
k=len(rows)-1
while k>0 and rows[k][2]-rows[k-1][2] < 2000000: k-=1
rows=rows[k:]
t0=min(r[2] for r in rows)
rows.sort(key=lambda r:r[0])
for b,n,s,e in rows[::max(1,len(rows)//40)]: print('  block %4d n %2d: start %7.0f end %7.0f us'%(b,n,(s-t0)/100.0,(e-t0)/100.0))
print('  last end %.0f us'%(max(r[3] for r in rows)-t0)/100.0 if False else '  last end %.0f us; jobs sampled %d'%((max(r[3] for r in rows)-t0)/100.0,len(rows)))
"; done
