# tuning aid: C5 (Murty stress) with variants of the library built by tools/variant_bench.py --build (args: variant names)
for v in "$@"; do echo "== $v"; RFS_LIB=tools/_build/librfsgpu_$v.so timeout 200 python tools/c5_bench.py 2>&1 | grep "C5 RB"; done
if [ -f tools/_build/librfsgpu_q2p.so ] && [ -n "$C5_PROFILE" ]; then echo "== profile"; RFS_LIB=tools/_build/librfsgpu_q2p.so C5_STEPS=1 timeout 300 python tools/c5_bench.py 2>&1 | grep "${C5_GREP:-quad solver}" | sort | uniq | head -${C5_PROFILE}; fi
