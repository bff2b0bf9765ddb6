import ctypes as C, os, sys
import numpy as np
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/tests")
from __graft_entry__ import load_package
pkg = load_package()
lib = C.CDLL(os.path.join(ROOT, "tools", "_build", "librfsgpu_prof.so"))
pkg.engine._lib = lib
sc = pkg.scenarios
import importlib.util
spec = importlib.util.spec_from_file_location("tg", ROOT+"/tests/test_gpu_parity.py"); tg = importlib.util.module_from_spec(spec); spec.loader.exec_module(tg)
for kind in ["crowded", "half_crowded", "untrusted", "faint_parents", "tiny_weights"]:
    scen = tg._intensity_scenario(sc, kind, 77)
    f = pkg.RBPHDFilter(scen["n"], gm_capacity=640)
    sc.load_scenario(f, scen)
    out = (C.c_longlong * 64)()
    lib.rfsgpu_debug_sections(f._h, out)
    f.update_async(scen["Z"]); f.synchronize()
    lib.rfsgpu_debug_sections(f._h, out)
    t = np.array(list(out), dtype=np.int64)
    print(kind, "particle 7: pairs listed %d for %d evaluation points, row trips %d, dense fall-backs %d; N %s" % (t[12], t[15], t[13], t[14], f.gm_sizes()[:8]))
