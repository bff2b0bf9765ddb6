"""Boundary check: compile and link the UNMODIFIED reference drivers against the drop-in.

`src/rbphdslam2dSim.cpp` and `src/rbphdslam_VictoriaPark.cpp` are compiled where they lie under /root/reference, with
`integration/include/` (our `RBPHDFilter.hpp`) ahead of the reference's `include/` on the include path, and linked against
`librfsgpu.so`.  Eigen3 and Boost are not in this image; `tests/support/stubs/` holds small stand-ins for the parts those
translation units use.  This is TEST SUPPORT for the boundary only: nothing built here is a parity oracle, a golden
vector source or a CPU baseline (the filter inside these binaries is the GPU engine), and no reference source is copied --
the outputs (two executables) go to tests/support/_build/, which is git-ignored.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
OUT = os.path.join(HERE, "_build")
PKG = os.path.join(ROOT, "rfs-slam_amd")

COMMON = ["TimeStamp", "Timer", "MeasurementModel_RngBrg", "KalmanFilter_RngBrg"]
DRIVERS = {
    "rbphdslam2dSim": ["rbphdslam2dSim", "ProcessModel_Odometry2D"] + COMMON,
    "rbphdslam_VictoriaPark": ["rbphdslam_VictoriaPark", "ProcessModel_Ackerman2D", "MeasurementModel_VictoriaPark"] + COMMON,
}


def reference_present():
    return os.path.isfile(os.path.join(REF, "src", "rbphdslam2dSim.cpp"))


def binary(name):
    return os.path.join(OUT, name)


def command(name):
    srcs = [os.path.join(REF, "src", s + ".cpp") for s in DRIVERS[name]]
    return (["g++", "-std=c++17", "-O2", "-w",
             "-I" + os.path.join(ROOT, "integration", "include"),   # RBPHDFilter.hpp -> the GPU-backed class template
             "-I" + os.path.join(ROOT, "include"),                  # rfsgpu.h
             "-I" + os.path.join(HERE, "stubs"),                    # stand-in Eigen / Boost (test support)
             "-I" + os.path.join(REF, "include")]                   # every other reference header, unmodified
            + srcs + ["-L" + PKG, "-lrfsgpu", "-Wl,-rpath,$ORIGIN/../../../rfs-slam_amd", "-Wl,-rpath," + PKG,
                      "-Wl,-rpath,/opt/rocm/lib", "-o", binary(name)])


def stale(name):
    b = binary(name)
    if not os.path.exists(b):
        return True
    t = os.path.getmtime(b)
    deps = [os.path.join(ROOT, "integration", "RBPHDFilter_rfsgpu.hpp"), os.path.join(ROOT, "include", "rfsgpu.h"),
            os.path.join(PKG, "librfsgpu.so")]
    for d, _, fs in os.walk(os.path.join(HERE, "stubs")):
        deps += [os.path.join(d, f) for f in fs]
    return any(os.path.getmtime(d) > t for d in deps)


def build(name, force=False, verbose=False):
    """Returns (returncode, compiler output).  The library must exist already (rfs-slam_amd/build.py)."""
    if not force and not stale(name):
        return 0, ""
    os.makedirs(OUT, exist_ok=True)
    cmd = command(name)
    if verbose:
        print(" ".join(cmd))
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    return p.returncode, p.stdout


def build_all(verbose=False):
    for name in DRIVERS:
        rc, out = build(name, verbose=verbose)
        if rc != 0:
            raise RuntimeError("reference driver %s does not compile against the binding:\n%s" % (name, out[-4000:]))
