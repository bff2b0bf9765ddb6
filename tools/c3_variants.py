import sys, os, subprocess, json
ROOT='/root/repo'
for name in sys.argv[1:]:
    lib=os.path.join(ROOT,'tools','_build',f'librfsgpu_{name}.so')
    code=(f"import sys; sys.path.insert(0,{ROOT!r}); import __graft_entry__ as g; pkg=g.load_package(); pkg.engine.LIB={lib!r}; import bench; "
          "sys.argv=['bench.py','--workload','c3','--steps','100','--warmup','10','--no-cpu-baseline','--no-pmc']; bench.main()")
    r=subprocess.run([sys.executable,'-c',code],capture_output=True,text=True,cwd=ROOT)
    try:
        d=json.loads(r.stdout.strip().splitlines()[-1]); print(name,'c3',d['value'],d['ms_per_step'],d['config']['kernels']['phd_step_fused']['ms'],flush=True)
    except Exception: print(name,'FAILED',r.stdout[-200:],r.stderr[-400:])
