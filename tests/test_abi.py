"""CPU-only checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol
include/rfsgpu.h declares; struct layouts agree; without a GPU the engine refuses loudly (no fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, has_gpu


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "rfsgpu.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rfsgpu_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(pkg):
    pkg.build_mod.build()
    lib = pkg.load_library()
    syms = declared_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in rfsgpu.h but not exported"
    assert sorted("rfsgpu_" + s for s in pkg.capi.ABI_SYMBOLS) == syms
    assert lib.rfsgpu_abi_version() == 1


def test_struct_layouts_match_header(pkg):
    # sizes as the C compiler lays them out (doubles 8-aligned)
    assert C.sizeof(pkg.capi.FilterConfig) == 104
    assert C.sizeof(pkg.capi.RngBrgConfig) == 72
    assert C.sizeof(pkg.capi.KFConfig) == 16
    assert C.sizeof(pkg.capi.Timing) == 14 * 8
    cfg = pkg.capi.FilterConfig()
    lib = pkg.load_library()
    lib.rfsgpu_default_filter_config(C.byref(cfg))
    # RBPHDFilter.hpp:370-382
    assert cfg.birthGaussianWeight == 0.25 and cfg.importanceWeightingEvalPointCount == 8
    assert cfg.gaussianMergingThreshold == 0.5 and cfg.gaussianPruningThreshold == 0.2
    assert cfg.newGaussianCreateInnovMDThreshold == 0.2 and cfg.minUpdatesBeforeResample == 1


@pytest.mark.skipif(has_gpu(), reason="checks the no-device behaviour")
def test_no_gpu_means_loud_failure_not_fallback(pkg):
    with pytest.raises(pkg.capi.EngineError) as e:
        pkg.RBPHDFilter(8)
    assert e.value.status in (pkg.capi.ERR_NO_DEVICE, pkg.capi.ERR_HIP)
    with pytest.raises(pkg.capi.EngineError):
        pkg.mat_perm(np.eye(3))


def test_product_package_never_touches_the_oracle():
    """The shipped path must not import / link / execute anything under oracle/."""
    pkgdir = os.path.join(ROOT, "rfs-slam_amd")
    for dp, _, files in os.walk(pkgdir):
        for fn in files:
            if fn.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                txt = open(os.path.join(dp, fn)).read()
                assert "rbphd_oracle" not in txt and "rfsor_" not in txt and "from oracle" not in txt and "import oracle" not in txt, fn


def test_resample_plan_matches_oracle_restatement(pkg, ob):
    rng = np.random.default_rng(5)
    for n in (5, 64, 257):
        for _ in range(5):
            w = rng.uniform(0, 1, n) ** 6
            u = float(rng.uniform())
            fired, wn, src = ob.resample_decide(w, n + 1.0, u)  # threshold > n: always resample
            assert fired
            plan = pkg.engine.systematic_resample_plan(wn, u)
            assert np.array_equal(plan, src)
            # every source keeps itself; kept slots are exactly the sampled ones
            assert np.all(src[src] == src)


def core_symbols(tag="RFSGPU_CORE"):
    txt = open(os.path.join(ROOT, "include", "rfsgpu.h")).read()
    return sorted({w for line in re.findall(tag + r":(.*)", txt) for w in line.split()})


def test_stable_core_is_small_declared_and_sufficient_for_the_reference_side_binding(pkg):
    """include/rfsgpu.h marks a stable core (98 exports are too many for a maintainer to face): every core name is a
    declared export, the core stays small, and the reference-side binding uses nothing outside it -- checked on the binding's
    source and, where they have been built (needs /root/reference), on the undefined symbols of the linked reference drivers."""
    core = core_symbols()
    assert 15 <= len(core) <= 25, core
    assert set(core) <= set(declared_symbols())
    src = open(os.path.join(ROOT, "integration", "RBPHDFilter_rfsgpu.hpp")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    multi = core_symbols("RFSGPU_CORE_MULTI")          # the group counterparts the binding calls when RFSGPU_DEVICES lists several GPUs
    assert 15 <= len(multi) <= 22 and set(multi) <= set(declared_symbols())
    used = set(re.findall(r"\b(rfsgpu_[a-z_0-9]+)\s*\(", src)) & set(declared_symbols())     # (the binding's helper classes also start with rfsgpu_)
    assert used and used <= set(core) | set(multi), used - set(core) - set(multi)
    assert set(multi) <= used                            # (the list names exactly what the facade forwards to)
    import subprocess
    for name in ("rbphdslam2dSim", "rbphdslam_VictoriaPark"):
        exe = os.path.join(ROOT, "tests", "support", "_build", name)
        if os.path.exists(exe):
            und = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True).stdout
            assert set(re.findall(r"\b(rfsgpu_\w+)", und)) <= set(core) | set(multi)


def test_library_build_goes_through_the_assembler_text(pkg, tmp_path):
    """build.compile_library takes the device code through its assembly text (-save-temps): the target's text parser refuses
    `s_mov_b64 s[a:b], <64-bit literal>` -- an operand gfx950 does not have, which this backend emits under -disable-machine-cse
    and whose literal the integrated path cuts to 32 bits in silence (exp() of a negative argument was 0).
    The premise (the assembler refuses the line; it takes the two-instruction form) and the build path's flag."""
    import inspect
    import shutil
    import subprocess
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if not os.path.exists(clang):
        pytest.skip("no ROCm LLVM here")
    assert "-save-temps" in inspect.getsource(pkg.build_mod.compile_library)
    bad = tmp_path / "bad.s"
    bad.write_text('\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"\n\t.text\n\ts_mov_b64 s[0:1], 0x4090000000000000\n')
    r = subprocess.run([clang, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(bad), "-o", str(tmp_path / "bad.o")],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "invalid operand" in r.stderr
    good = tmp_path / "good.s"
    good.write_text('\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"\n\t.text\n\ts_mov_b32 s0, 0\n\ts_mov_b32 s1, 0x40900000\n')
    subprocess.check_call([clang, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(good), "-o", str(tmp_path / "good.o")])
    shutil.rmtree(tmp_path, ignore_errors=True)
