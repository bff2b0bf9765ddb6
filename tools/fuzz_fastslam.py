#!/usr/bin/env python3
"""Randomised differential run of the FastSLAM / MH-FastSLAM update, device vs oracle (GPU box): random shapes, hypothesis
counts 1..5, likelihood-difference windows, candidate thresholds; four cycles with particle growth, forced and conditional
resample(nParticles_init), candidate inheritance.   python tools/fuzz_fastslam.py [n_cases] [seed]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
import importlib
pkg = load_package()
sc = pkg.scenarios
ob = importlib.import_module("oracle.binding")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def compare(dev, orc):
    assert dev.n == orc.n, (dev.n, orc.n)
    wd, wo = dev.get_weights(), orc.get_weights()
    # (a zero / non-finite weight sum: raw weights, agreement on non-finiteness, counted -- never NaN against NaN)
    global n_degenerate
    sd, so = float(np.sum(wd)), float(np.sum(wo))
    if np.isfinite(sd) and np.isfinite(so) and sd > 0.0 and so > 0.0:
        np.testing.assert_allclose(wd / sd, wo / so, rtol=1e-9, atol=1e-300)
    else:
        n_degenerate += 1
        fd, fo = np.isfinite(wd), np.isfinite(wo)
        assert np.array_equal(fd, fo) and np.array_equal(np.isnan(wd), np.isnan(wo)), "device and oracle disagree on which weights are finite"
        np.testing.assert_allclose(wd[fd], wo[fo], rtol=1e-9, atol=1e-300)
    assert np.array_equal(dev.gm_sizes(), orc.gm_sizes())
    for i in range(dev.n):
        sc.assert_gm_close(dev.export_gm(i), orc.export_gm(i), 1e-10, 1e-12, ordered=True)
        md, cd, sd, kd = dev.export_birth_candidates(i)
        mo, co, so, ko = orc.export_birth_candidates(i)
        assert list(sd) == list(so) and list(kd) == list(ko), i
        np.testing.assert_allclose(md, mo, rtol=1e-9, atol=1e-11)


bad = grown = 0
n_degenerate = 0
for case in range(n_cases):
    hyp = int(rng.choice([1, 1, 2, 3, 5]))
    n0 = int(rng.integers(3, 10))
    kw = dict(n_particles=n0, n_landmarks=int(rng.choice([2, 6, 12, 25, 40])), n_z=int(rng.integers(1, 14)), seed=int(rng.integers(1 << 30)),
              rmax=float(rng.choice([4.0, 5.0, 8.0])))
    if hyp == 1 and rng.random() < 0.4:
        kw.update(n_landmarks=int(rng.choice([70, 130, 200])), n_z=int(rng.integers(10, 40)), rmax=float(rng.choice([12.0, 25.0])))
    vp = rng.random() < 0.3          # Victoria Park model (3-D landmarks, scan-based Pd)
    if vp:
        kw = dict(n_particles=n0, n_landmarks=int(rng.choice([3, 10, 30, 60])), n_z=int(rng.integers(1, 14)), seed=int(rng.integers(1 << 30)),
                  scan=str(rng.choice(["const", "ragged"])))
        scen = sc.make_vp_scenario(**kw)
        if rng.random() < 0.5:
            scen["mean"][:, ::3, 2] = rng.uniform(0.04, 0.2)
            scen["cov"][:, ::3, 0, 0] *= rng.uniform(10, 200)
            scen["cov"][:, ::3, 1, 1] *= rng.uniform(10, 200)
    else:
        scen = sc.make_scenario(**kw)
    diff = float(rng.choice([2.0, 5.0, 50.0]))
    model = pkg.capi.MODEL_VICTORIAPARK_3D if vp else pkg.capi.MODEL_RNGBRG_2D
    dev = pkg.RBPHDFilter(n0, gm_capacity=320, max_particles=n0 * hyp * 4, model=model)
    orc = ob.OracleFilter(n0, model=model)
    try:
        for f in (dev, orc):
            sc.load_scenario(f, scen)
            for i in range(n0):
                f.import_gm(i, np.zeros(scen["w"][i].shape), scen["mean"][i], scen["cov"][i])
            cfg = f.default_fastslam_config()
            cfg.maxNDataAssocHypotheses = hyp
            cfg.maxDataAssocLogLikelihoodDiff = diff
            cfg.landmarkCandidateMeasurementCountThreshold = int(kw["seed"] % 3) + 1
            cfg.landmarkCandidateCurrentMeasurementCountThreshold = int(kw["seed"] % 2)
            cfg.landmarkCandidateMeasurementCheckThreshold = 3
            f.set_fastslam_config(cfg)
        poses = scen["poses"].copy()
        rz, ru = np.random.default_rng(kw["seed"]), np.random.default_rng(kw["seed"] + 1)
        for step in range(4):
            Z = scen["Z"] + rz.normal(0, 3e-3, scen["Z"].shape)
            for f in (dev, orc):
                f.predict_map(False)
                f.fastslam_update(Z)
            par = dev.particle_parents()
            assert np.array_equal(par, orc.particle_parents())
            grown += int(dev.n > len(poses))
            poses = poses[par]
            compare(dev, orc)
            for f in (dev, orc):
                s = f.weight_sums()
                f.normalize_weights(s[0])
            resampled = dev.n > 2 * n0 or step == 2
            if resampled:
                plan = pkg.engine.systematic_resample_plan(orc.get_weights(), float(ru.random()), n_out=n0)
                for f in (dev, orc):
                    f.resample_apply(plan, n_out=n0)
                poses = poses[plan]
                compare(dev, orc)
            for f in (dev, orc):
                f.fastslam_set_resample_occured(resampled)
                f.set_poses(poses, scen["pose_cov"])
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("VP CASE" if vp else "CASE", case, kw, "hyp", hyp, "diff", diff, "->", type(e).__name__, str(e)[:300], flush=True)
    dev.close()
print("fastslam fuzz: %d cases, %d failures (updates that multiplied particles: %d); %d weight comparisons had a zero / non-finite sum "
      "and were made on the raw weights" % (n_cases, bad, grown, n_degenerate))
sys.exit(1 if bad else 0)
