"""hipcc driver: compiles csrc/*.hip for gfx950 into one in-tree shared library (librfsgpu.so)."""
import os
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librfsgpu.so")
SOURCES = ["rfsgpu_engine.hip"]
# Two code objects for the one target, picked by the runtime from the device's XNACK mode: without XNACK replay to allow for, the compiler may
# reuse a load's address registers early and drops the padding around memory clauses (+0.6 % on the fused step, profiles/r04p_*); a device run
# with XNACK on still finds its image.  -parallel-jobs: both in the time of one.
# -disable-machine-licm (round 5): the backend's loop-invariant code motion hoists constant pairs (libm coefficients) and per-lane addresses out
# of the long search / phase loops and then spills them under the kernels' register caps -- without it murty_jobs_kernel<8,8> needs no scratch at
# all (60 B per lane with it, 74 MB of spill traffic per launch at configs[4]), the step kernel with the predict at its head 120 VGPRs and no
# scratch (128 + 60 B: C2b 0.1121 -> 0.1058 ms per cycle), the Victoria Park step 132 VGPRs (152); configs[1], [2] and [3] are unchanged to 1 %.
FLAGS = ["--offload-arch=gfx950:xnack-", "--offload-arch=gfx950:xnack+", "-parallel-jobs=2", "-O3", "-mllvm", "-disable-machine-licm", "-std=c++17", "-fPIC", "-shared", "-fgpu-rdc" if False else "-Wall", "-Wno-unused-result", "-Wno-unused-value",
         "-Wno-unused-function", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "rfsgpu.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=()):
    """Compile the extension if sources are newer than the library. Returns the library path."""
    if not force and not stale():
        return LIB
    return compile_library(list(extra_flags) + [os.path.join(CSRC, s) for s in SOURCES] + ["-ldl"], LIB, verbose)   # (-ldl: librccl is dlopen'ed by the multi-GPU group)


def compile_library(args, lib, verbose=False):
    """hipcc FLAGS args -o lib, with the device code taken THROUGH its assembly text (-save-temps, in a scratch directory that is removed
    afterwards).  Same instructions as the integrated path (checked: the disassembly of the two code objects is identical), but an
    instruction the target does not have stops the build instead of being encoded as something else: with `-mllvm -disable-machine-cse`
    this backend emits `s_mov_b64 s[a:b], <64-bit literal>`, which gfx950 lacks -- the integrated assembler cuts the literal to its low
    32 bits without a word (1024.0 became 0 and exp() of every negative argument 0: DESIGN 8, round 5), its text parser refuses the line."""
    tmp = tempfile.mkdtemp(prefix="rfsgpu_build_")
    try:
        out = os.path.join(tmp, os.path.basename(lib))
        cmd = [hipcc()] + FLAGS + ["-save-temps=obj"] + args + ["-o", out]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        shutil.move(out, lib)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return lib


# Test support, not product: the library with every Murty child solved FROM SCRATCH by the reference-faithful solver (-DMURTY_WARM=0: what
# rounds 1-5 shipped).  tests/test_gpu_parity.py holds the shipped library -- children started from their parent's dual variables, round 6
# -- against it (partition sums to 1e-12) and it against the oracle.  Built next to the other test artefacts (git-ignored, travels to the
# GPU box).
COLD_LIB = os.path.join(ROOT, "tests", "support", "_build", "librfsgpu_coldmurty.so")
COLD_FLAGS = ["-DMURTY_WARM=0"]


def build_cold_murty_variant(force=False, verbose=False):
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "rfsgpu.h")]
    if not force and os.path.exists(COLD_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(COLD_LIB) for d in deps):
        return COLD_LIB
    os.makedirs(os.path.dirname(COLD_LIB), exist_ok=True)
    return compile_library(COLD_FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-ldl"], COLD_LIB, verbose)


# ... and the shipped search with only 16 of the open-node array's positions in LDS (-DMURTY_HEAP_LDS=16; 512 in the product): the
# positions beyond live in the job's arena, a path configs[4]'s jobs never reach with 512 -- same weights bit for bit.
SMALLQ_LIB = os.path.join(ROOT, "tests", "support", "_build", "librfsgpu_smallqueue.so")


def build_small_queue_variant(force=False, verbose=False):
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "rfsgpu.h")]
    if not force and os.path.exists(SMALLQ_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(SMALLQ_LIB) for d in deps):
        return SMALLQ_LIB
    os.makedirs(os.path.dirname(SMALLQ_LIB), exist_ok=True)
    return compile_library(["-DMURTY_HEAP_LDS=16"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-ldl"], SMALLQ_LIB, verbose)


SIM = os.path.join(HERE, "host", "rbphdslam2d_sim")
SIM_FASTSLAM = os.path.join(HERE, "host", "fastslam2d_sim")
SIM_VP = os.path.join(HERE, "host", "rbphdslam_vp")


def build_host(force=False, verbose=False):
    """Compile the C++ host driver (plain g++, links the C-ABI library only)."""
    src = os.path.join(HERE, "host", "rbphdslam2d_sim.cpp")
    src_vp = os.path.join(HERE, "host", "rbphdslam_vp.cpp")
    hdrs = [os.path.join(HERE, "host", "rbphd_filter.hpp"), os.path.join(HERE, "host", "xml_cfg.hpp")]
    newest = max([os.path.getmtime(p) for p in [src, src_vp, LIB] + hdrs])
    if not force and all(os.path.exists(p) and os.path.getmtime(p) > newest for p in (SIM, SIM_FASTSLAM, SIM_VP)):
        return SIM
    cmd = ["g++", "-std=c++17", "-O2", "-fopenmp", "-Wall", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "host"), src,
           "-L" + HERE, "-lrfsgpu", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath," + HERE, "-Wl,-rpath,/opt/rocm/lib", "-o", SIM]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    # the same simulator around the FastSLAM filter class (reference: src/fastslam2dSim.cpp)
    cmd2 = [c for c in cmd[:-1]] + [SIM_FASTSLAM, "-DUSE_FASTSLAM"]
    if verbose:
        print(" ".join(cmd2))
    subprocess.check_call(cmd2)
    # the Victoria Park host loop (reference: src/rbphdslam_VictoriaPark.cpp)
    cmd3 = [src_vp if c == src else c for c in cmd[:-1]] + [SIM_VP]
    if verbose:
        print(" ".join(cmd3))
    subprocess.check_call(cmd3)
    return SIM
