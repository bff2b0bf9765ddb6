"""ctypes binding of the parity oracle (oracle/librbphd_oracle.so) and of oracle/_ref/librfs_ref.so.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess
import sys
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)


def build(force=False):
    so = os.path.join(_HERE, "librbphd_oracle.so")
    src = os.path.join(_HERE, "rbphd_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)
    return so


def load():
    return C.CDLL(build())


def build_fast():
    """The same restatement compiled the way SURVEY 8(d) specifies the CPU baseline: -O3 -march=native -fopenmp (FMA
    contraction on), for bench.py's `cpu_baseline` leg ONLY -- the parity tests keep the strict -O2 -ffp-contract=off
    library.  Compiled on the machine that runs it (-march=native), keyed on the CPU model; returns (path, flags)."""
    import hashlib
    try:
        model = [l for l in open("/proc/cpuinfo") if l.startswith("model name")][0].split(":", 1)[1].strip()
    except Exception:
        model = "unknown"
    tag = hashlib.sha1(model.encode()).hexdigest()[:10]
    src = os.path.join(_HERE, "rbphd_oracle.cpp")
    out_dir = os.path.join(_HERE, "_fast")
    os.makedirs(out_dir, exist_ok=True)
    for flags in (["-O3", "-march=native", "-fopenmp"], ["-O3", "-march=x86-64-v3", "-fopenmp"], ["-O3", "-fopenmp"]):
        so = os.path.join(out_dir, "librbphd_oracle_fast_%s_%s.so" % (tag, hashlib.sha1(" ".join(flags).encode()).hexdigest()[:6]))
        if os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(src):
            return so, flags
        cmd = ["g++", "-std=c++17"] + flags + ["-fPIC", "-shared", "-Wno-unused-variable", "-I" + os.path.join(_ROOT, "include"), "-o", so, src]
        if subprocess.call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) == 0:
            return so, flags
    raise RuntimeError("could not compile the -O3 oracle for the CPU baseline")


def cpu_info():
    """(model name, logical CPUs, physical cores) of this host."""
    model, cores = "unknown", set()
    phys = core = None
    try:
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                model = l.split(":", 1)[1].strip()
            elif l.startswith("physical id"):
                phys = l.split(":", 1)[1].strip()
            elif l.startswith("core id"):
                core = l.split(":", 1)[1].strip()
                cores.add((phys, core))
    except Exception:
        pass
    return model, os.cpu_count() or 1, (len(cores) or os.cpu_count() or 1)


def load_ref():
    """The two reference classes compiled from /root/reference (None when not built)."""
    so = os.path.join(_HERE, "_ref", "librfs_ref.so")
    if not os.path.exists(so):
        return None
    return C.CDLL(so)


def _capi():
    if _ROOT not in sys.path:
        sys.path.insert(0, _ROOT)
    from __graft_entry__ import load_package
    return load_package().capi


class OracleFilter:
    """Same interface as the product's filter class, backed by the CPU restatement."""

    def __new__(cls, n_particles, stable_sort=False, lib=None, **kw):
        # stable_sort=False (default): sortByWeight is std::sort, exactly as the reference (include/GaussianMixture.hpp:523-534) --
        # the mode every parity test uses; True: ties by index (what the device did until round 4; tools/tie_order_study.py)
        capi = _capi()
        lib = lib or load()
        obj = capi.CFilter(lib, "rfsor_", n_particles, **kw)
        lib.rfsor_set_stable_sort(obj._h, C.c_int(1 if stable_sort else 0))
        obj.murty_calls = lambda: _long(lib.rfsor_murty_calls, obj._h)
        obj.lonerow_bug_hits = lambda: _long(lib.rfsor_lonerow_bug_hits, obj._h)
        obj.fs_solver_max_dim = lambda: _long(lib.rfsor_fs_solver_max_dim, obj._h)
        obj.set_stable_sort = lambda on: lib.rfsor_set_stable_sort(obj._h, C.c_int(1 if on else 0))
        return obj


def vp_pd(f, pose, lx, lS):
    """Oracle probe: MeasurementModel_VictoriaPark::probabilityOfDetection -> (Pd, isCloseToSensingLimit)."""
    lib = load()
    lib.rfsor_vp_pd.restype = C.c_double
    pose, lx, lS = (np.ascontiguousarray(a, dtype=np.float64) for a in (pose, lx, lS))
    close = C.c_int()
    v = lib.rfsor_vp_pd(f._h, pose.ctypes.data_as(C.c_void_p), lx.ctypes.data_as(C.c_void_p), lS.ctypes.data_as(C.c_void_p), C.byref(close))
    return v, bool(close.value)


def vp_measure(f, pose, lx, lS):
    lib = load()
    pose, lx, lS = (np.ascontiguousarray(a, dtype=np.float64) for a in (pose, lx, lS))
    z, S, H = np.empty(3), np.empty((3, 3)), np.empty((3, 3))
    lib.rfsor_vp_measure(f._h, *(a.ctypes.data_as(C.c_void_p) for a in (pose, lx, lS, z, S, H)))
    return z, S, H


def vp_clutter(f):
    lib = load()
    lib.rfsor_vp_clutter.restype = C.c_double
    return lib.rfsor_vp_clutter(f._h)


def _long(fn, h):
    fn.restype = C.c_long
    return fn(h)


def set_threads(n):
    load().rfsor_set_threads(C.c_int(n))


def max_threads():
    return load().rfsor_max_threads()


def mat_perm(A):
    return _capi().mat_perm(load(), "rfsor_", A)


def resample_decide(weights, eff_n_threshold, u01):
    """ParticleFilter::resample decision (include/ParticleFilter.hpp:399-492). Returns (fired, w_norm, src_slot)."""
    lib = load()
    w = np.ascontiguousarray(weights, dtype=np.float64).copy()
    src = np.empty(w.size, dtype=np.int32)
    lib.rfsor_resample_decide.restype = C.c_int
    fired = lib.rfsor_resample_decide(w.ctypes.data_as(C.c_void_p), C.c_int(w.size), C.c_double(eff_n_threshold),
                                      C.c_double(u01), src.ctypes.data_as(C.c_void_p))
    return bool(fired), w, src


def permlex_all(nM, nZ, lib=None, sym="rfsor_permlex_all", max_perm=200000):
    lib = lib or load()
    out = np.zeros((max_perm, nM + nZ), dtype=np.uint32)
    fn = getattr(lib, sym)
    fn.restype = C.c_int
    k = fn(C.c_uint(nM), C.c_uint(nZ), out.ctypes.data_as(C.c_void_p), C.c_int(max_perm))
    return out[:k].copy()


def hungarian(Cm):
    lib = load()
    Cm = np.ascontiguousarray(Cm, dtype=np.float64).copy()
    n = Cm.shape[0]
    soln = np.empty(n, dtype=np.int32)
    cost = C.c_double()
    lib.rfsor_hungarian.restype = C.c_int
    ok = lib.rfsor_hungarian(Cm.ctypes.data_as(C.c_void_p), C.c_int(n), soln.ctypes.data_as(C.c_void_p), C.byref(cost))
    return bool(ok), soln, cost.value, Cm


def murty(Cm, nR=None, nC=None, kmax=200):
    lib = load()
    Cm = np.ascontiguousarray(Cm, dtype=np.float64).copy()
    n = Cm.shape[0]
    nR = n if nR is None else nR
    nC = n if nC is None else nC
    scores = np.empty(kmax)
    assign = np.empty((kmax, n), dtype=np.int32)
    lib.rfsor_murty.restype = C.c_int
    k = lib.rfsor_murty(Cm.ctypes.data_as(C.c_void_p), C.c_int(n), C.c_int(nR), C.c_int(nC), C.c_int(kmax),
                        scores.ctypes.data_as(C.c_void_p), assign.ctypes.data_as(C.c_void_p))
    return scores[:k].copy(), assign[:k].copy()


def partition_likelihood(L, evalPd, clutter, clutter_integral):
    lib = load()
    L = np.ascontiguousarray(L, dtype=np.float64)
    nE, nZ = L.shape
    pd = np.ascontiguousarray(evalPd, dtype=np.float64)
    mc, lr = C.c_long(0), C.c_long(0)
    lib.rfsor_partition_likelihood.restype = C.c_double
    v = lib.rfsor_partition_likelihood(L.ctypes.data_as(C.c_void_p), C.c_int(nE), C.c_int(nZ), pd.ctypes.data_as(C.c_void_p),
                                       C.c_double(clutter), C.c_double(clutter_integral), C.byref(mc), C.byref(lr))
    return v, mc.value, lr.value


def ref_bruteforce(Cm, kmax=100000):
    lib = load_ref()
    Cm = np.ascontiguousarray(Cm, dtype=np.float64)
    n = Cm.shape[0]
    scores = np.empty(kmax)
    assign = np.empty((kmax, n), dtype=np.uint32)
    lib.rfsref_bruteforce.restype = C.c_int
    cnt = lib.rfsref_bruteforce(Cm.ctypes.data_as(C.c_void_p), C.c_int(n), C.c_int(kmax), scores.ctypes.data_as(C.c_void_p),
                                assign.ctypes.data_as(C.c_void_p))
    k = min(cnt, kmax)
    return scores[:k].copy(), assign[:k].copy()
