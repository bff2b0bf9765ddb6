"""rfs-slam_amd: MI355X-native RB-PHD SLAM update engine (HIP kernels behind the C ABI in include/rfsgpu.h).

The directory name carries a hyphen (project convention), so it is loaded through
`__graft_entry__.load_package()` under the module name `rfs_slam_amd`.

Layout: csrc/ (HIP kernels + the C-ABI, one shared library librfsgpu.so), capi.py (ctypes binding),
engine.py (library loader + the RBPHDFilter host mirror), build.py (hipcc driver).
There is no CPU fallback: constructing a filter without the built extension or without a gfx950
device raises.
"""
from . import capi  # noqa: F401
from . import build as build_mod  # noqa: F401
from . import engine  # noqa: F401
from . import scenarios  # noqa: F401
from . import sharded  # noqa: F401
from . import vp_driver  # noqa: F401
from . import sim2d_driver  # noqa: F401
from .engine import FastSLAM, FilterGroup, RBPHDFilter, load_library, mat_perm  # noqa: F401
