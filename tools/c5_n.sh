for v in old q2; do for n in 125 250 1000; do echo "== $v N=$n"; C5_N=$n RFS_LIB=tools/_build/librfsgpu_$v.so timeout 200 python tools/c5_bench.py 2>&1 | grep "C5 RB" | cut -c1-140; done; done
