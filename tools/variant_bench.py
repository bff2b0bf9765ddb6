#!/usr/bin/env python3
"""Tuning aid (not part of the product): build variants of the HIP library with extra -D flags and bench each.

  python tools/variant_bench.py --build name1="-DA=1 -DB=2" name2="..."     (here: cross-compiles into tools/_build/)
  python tools/variant_bench.py --run name1 name2                           (on the GPU box)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "_build")


def lib_of(name):
    return os.path.join(OUT, f"librfsgpu_{name}.so")


if sys.argv[1] == "--build":
    import __graft_entry__ as g
    bm = g.load_package().build_mod
    os.makedirs(OUT, exist_ok=True)
    # through the assembly text like the product build (build.compile_library): a flag under which the backend emits an instruction the
    # target lacks fails here instead of giving a library that computes something else (the -disable-machine-cse episode)
    from concurrent.futures import ThreadPoolExecutor

    def one(spec):
        name, flags = spec.split("=", 1)
        try:
            bm.compile_library(flags.split() + [os.path.join(bm.CSRC, "rfsgpu_engine.hip"), "-ldl"], lib_of(name))
            return name, 0
        except subprocess.CalledProcessError as e:
            return name, e.returncode
    with ThreadPoolExecutor(max_workers=4) as ex:
        for name, rc in ex.map(one, sys.argv[2:]):
            print(name, "rc", rc)
else:
    steps = os.environ.get("VB_STEPS", "200")
    wl = os.environ.get("VB_WORKLOAD", "c2a")
    for name in sys.argv[2:]:
        code = (f"import sys; sys.path.insert(0, {ROOT!r}); import __graft_entry__ as g; pkg = g.load_package(); "
                f"pkg.engine.LIB = {lib_of(name)!r}; import bench; "
                f"sys.argv = ['bench.py', '--workload', '{wl}', '--steps', '{steps}', '--warmup', '20', '--no-cpu-baseline', '--no-pmc', '--no-boundary']; bench.main()")
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            ks = d["config"]["kernels"]
            flat = {k: v["ms"] for k, v in ks.items() if "ms" in v}
            flat.update({k: v["ms"] for k, v in ks.get("standalone_phases_untimed_pass", {}).items()})
            print(name, d["value"], d["ms_per_step"], flat, flush=True)
        except Exception:
            print(name, "FAILED", r.stdout[-300:], r.stderr[-600:], flush=True)
