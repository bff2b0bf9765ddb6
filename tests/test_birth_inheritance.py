"""birth-state inheritance after a resampling, the reference's way (RFSGPU_INHERIT_REFERENCE, the default).

Reference: `RBPHDFilter::addBirthGaussians`, include/RBPHDFilter.hpp:1005-1011 -- inside the slot-ascending loop of the first
predict(s) after a resampling, `unused_measurements_[i]` and `birthGaussians_[i]` are copied from SLOT `getParentId()` *in the
state that slot is in at that moment*; ids as `ParticleFilter::resample` (include/ParticleFilter.hpp:446-479) and
`Particle::copy` (include/Particle.hpp:218-223) leave them: a copy keeps its source's id, `nLandmarksInFOV_` is never copied.

CPU: the oracle's restatement hand-checked on the three cases VERDICT r2 names (child above its parent gets an empty list, child
below gets the full list, id != slot after a second resampling), with immediate births and with candidate lists.
GPU: the same scripts and randomised resampling cycles on the device against the oracle (maps, lists, ids), both models.
"""
import numpy as np
import pytest


def _popcount(m):
    return bin(int(m)).count("1")


def _setup_immediate(f, sc, n, n_z, seed=5):
    """Empty maps, n_z measurements: after the update every measurement is unused for every particle; the masks are then set by
    hand.  CountThreshold == 1: every unused measurement is born at once, no candidate list exists."""
    scen = sc.make_scenario(n, 0, n_z, seed=seed)
    sc.load_scenario(f, scen)
    f.update(scen["Z"])
    return scen


def script_immediate(f, sc):
    """Returns what the three cases produce: mixture sizes after each predict, ids after each resampling."""
    n, n_z = 5, 6
    scen = _setup_immediate(f, sc, n, n_z)
    out = {}
    assert list(f.gm_sizes()) == [0] * n
    masks = np.array([0b000001, 0b000110, 0b001000, 0b110001, 0b010000], dtype=np.uint64)   # 1, 2, 1, 3, 1 unused measurements
    f.set_unused_masks(masks)
    # first resampling: slots 1 and 3 survive; slot 0 <- 1 (parent ABOVE), slot 2 <- 3 (parent above), slot 4 <- 3 (parent BELOW)
    f.resample_apply(np.array([1, 1, 3, 3, 3], dtype=np.int32))
    out["ids1"] = f.get_particle_ids()
    out["masks_after_resample"] = f.get_unused_masks().copy()
    f.predict_map(True)
    out["sizes1"] = f.gm_sizes().copy()
    out["masks_after_predict"] = f.get_unused_masks().copy()
    # an update with measurements clears resampleOccured_; new masks by hand; second resampling: slot 4 <- 0
    f.update(scen["Z"])
    out["sizes_after_update"] = f.gm_sizes().copy()
    masks2 = np.array([0b000001, 0b000110, 0b001001, 0b111000, 0b100000], dtype=np.uint64)  # 1, 2, 2, 3, 1
    f.set_unused_masks(masks2)
    f.resample_apply(np.array([0, 1, 2, 3, 0], dtype=np.int32))
    out["ids2"] = f.get_particle_ids()
    f.predict_map(True)
    out["sizes2"] = f.gm_sizes().copy()
    # a second predict before the next update: resampleOccured_ is still set, the copy runs again (from empty lists now)
    f.predict_map(True)
    out["sizes3"] = f.gm_sizes().copy()
    return out


def check_immediate(out):
    ids, par = out["ids1"]
    assert list(ids) == [1, 1, 3, 3, 3]            # a copy keeps its source's id (Particle::copy)
    assert list(par) == [1, 1, 3, 3, 3]            # setParentId(source's id) / setParentId(own id) for the survivors
    # nothing but pose + mixture moved at resampling time
    assert [int(m) for m in out["masks_after_resample"]] == [0b000001, 0b000110, 0b001000, 0b110001, 0b010000]
    # predict: slot 0 takes slot 1's list (not yet consumed: 2 births); slot 1 its own (2); slot 2 takes slot 3's (3);
    # slot 3 its own (3); slot 4 takes slot 3's list AFTER slot 3 consumed it: nothing
    assert list(out["sizes1"]) == [2, 2, 3, 3, 0]
    assert [int(m) for m in out["masks_after_predict"]] == [0, 0, 0, 0, 0]
    base = out["sizes_after_update"]
    ids, par = out["ids2"]
    assert list(ids) == [1, 1, 3, 3, 1]
    assert list(par) == [1, 1, 3, 3, 1]            # slot 0 SURVIVED, yet idParent_ = its id = 1 != 0 (case 1, :466-467)
    grow = out["sizes2"] - base[[0, 1, 2, 3, 0]]   # (slot 4's mixture is now a copy of slot 0's)
    # slot 0 copies slot 1's list (2) although it survived; slot 1 own (2); slot 2 copies slot 3's (3); slot 3 own (3);
    # slot 4 (id 1) copies slot 1's list after slot 1 consumed it: 0
    assert list(grow) == [2, 2, 3, 3, 0]
    assert list(out["sizes3"]) == list(out["sizes2"])


def _cand(k, x0):
    """k candidates far apart, 1 supporting measurement, 0 checks."""
    mean = np.array([[x0 + 10.0 * j, 5.0] for j in range(k)], dtype=np.float64).reshape(k, 2)
    cov = np.tile(np.eye(2) * 0.01, (k, 1, 1))
    return mean, cov, np.ones(k, dtype=np.int32), np.zeros(k, dtype=np.int32)


def script_candidates(f, sc):
    """Candidate lists: CountThreshold 3 (nobody is promoted), CheckThreshold 100 (nobody expires), FOV counts above the
    current-measurement threshold.  Every predict adds one check to each candidate of each list it visits."""
    n = 5
    scen = sc.make_scenario(n, 0, 4, seed=6)
    sc.load_scenario(f, scen)
    cfg = f.get_filter_config()
    cfg.birthGaussianMeasurementCountThreshold = 3
    cfg.birthGaussianMeasurementCheckThreshold = 100
    cfg.birthGaussianCurrentMeasurementCountThreshold = 0
    f.set_filter_config(cfg)
    f.update(scen["Z"])
    f.set_unused_masks(np.zeros(n, dtype=np.uint64))
    for i in range(n):
        f.import_birth_candidates(i, *_cand(i + 1, 100.0 * i))     # slot i: i + 1 candidates around x = 100 i
        f.import_aux(i, [], 5 + i)                                   # nLandmarksInFOV_[i] = 5 + i
    f.resample_apply(np.array([1, 1, 3, 3, 3], dtype=np.int32))
    f.predict_map(True)
    out = {"lists": [f.export_birth_candidates(i) for i in range(n)], "fov": [f.landmarks_in_fov(i) for i in range(n)],
           "sizes": f.gm_sizes().copy()}
    return out


def check_candidates(out):
    cnt = [len(l[2]) for l in out["lists"]]
    # slot 0 <- list of slot 1 (2 candidates), slot 2 <- slot 3 (4), slot 4 <- slot 3 after slot 3's own pass (4)
    assert cnt == [2, 2, 4, 4, 4]
    chk = [list(l[3]) for l in out["lists"]]
    assert chk[0] == [1, 1] and chk[1] == [1, 1] and chk[2] == [1] * 4 and chk[3] == [1] * 4
    assert chk[4] == [2] * 4                      # copied AFTER slot 3's promotion pass (one check older), then checked again
    x = [l[0][:, 0].tolist() if len(l[0]) else [] for l in out["lists"]]
    assert x[0] == x[1] == [100.0, 110.0] and x[2] == x[3] == x[4] == [300.0, 310.0, 320.0, 330.0]
    assert out["fov"] == [5, 6, 7, 8, 9]          # nLandmarksInFOV_ is per slot and is never copied
    assert list(out["sizes"]) == [0] * 5


def test_oracle_reference_inheritance_three_cases(ob, sc):
    orc = ob.OracleFilter(5)
    assert orc.get_birth_inheritance() == 0
    check_immediate(script_immediate(orc, sc))


def test_oracle_reference_inheritance_with_candidate_lists(ob, sc):
    check_candidates(script_candidates(ob.OracleFilter(5), sc))


def test_oracle_eager_mode_is_the_old_rule(ob, sc):
    """RFSGPU_INHERIT_EAGER (rounds 1-2): the child takes the parent's lists and FOV count at resampling time."""
    orc = ob.OracleFilter(5)
    orc.set_birth_inheritance(1)
    _setup_immediate(orc, sc, 5, 6)
    orc.set_unused_masks(np.array([1, 6, 8, 49, 16], dtype=np.uint64))
    orc.resample_apply(np.array([1, 1, 3, 3, 3], dtype=np.int32))
    assert [int(m) for m in orc.get_unused_masks()] == [6, 6, 49, 49, 49]
    orc.predict_map(True)
    assert list(orc.gm_sizes()) == [2, 2, 3, 3, 3]


# ---- device ------------------------------------------------------------------------------------------------------------


def _pair(pkg, ob, n, cap=128, model=None):
    kw = {} if model is None else {"model": model}
    return pkg.RBPHDFilter(n, gm_capacity=cap, **kw), ob.OracleFilter(n, **kw)


def _compare_all(sc, dev, orc, n, lists=True):
    assert np.array_equal(dev.gm_sizes(), orc.gm_sizes())
    for i in range(n):
        sc.assert_gm_close(dev.export_gm(i), orc.export_gm(i), 1e-10, 1e-12, ordered=True)
        assert list(dev.get_unused(i)) == list(orc.get_unused(i)), i
        assert dev.landmarks_in_fov(i) == orc.landmarks_in_fov(i), i
        if lists:
            md, cd, sd, kd = dev.export_birth_candidates(i)
            mo, co, so, ko = orc.export_birth_candidates(i)
            assert list(sd) == list(so) and list(kd) == list(ko), i
            np.testing.assert_allclose(md, mo, rtol=1e-9, atol=1e-11)
            np.testing.assert_allclose(cd, co, rtol=1e-9, atol=1e-13)
    for a, b in zip(dev.get_particle_ids(), orc.get_particle_ids()):
        assert np.array_equal(a, b)


@pytest.mark.gpu
def test_device_reference_inheritance_three_cases(pkg, ob, sc):
    dev, orc = _pair(pkg, ob, 5)
    assert dev.get_birth_inheritance() == pkg.capi.INHERIT_REFERENCE
    od, oo = script_immediate(dev, sc), script_immediate(orc, sc)
    check_immediate(od)
    for k in oo:
        if k.startswith("ids"):
            assert np.array_equal(od[k][0], oo[k][0]) and np.array_equal(od[k][1], oo[k][1])
        else:
            assert np.array_equal(od[k], oo[k]), k
    _compare_all(sc, dev, orc, 5)


@pytest.mark.gpu
def test_device_reference_inheritance_with_candidate_lists(pkg, ob, sc):
    dev, orc = _pair(pkg, ob, 5)
    od, oo = script_candidates(dev, sc), script_candidates(orc, sc)
    check_candidates(od)
    _compare_all(sc, dev, orc, 5)


def _random_plan(rng, n, survivors=None, forced=None):
    """A resampling plan of the shape ParticleFilter::resample produces: survivors keep themselves, every other slot is a copy of
    a survivor -- with children both below and above their parents.  `forced`: {child: parent}."""
    if survivors is None:
        survivors = rng.choice(n, size=int(rng.integers(2, max(3, n // 2))), replace=False)
    surv = np.unique(np.asarray(survivors))
    src = np.arange(n, dtype=np.int32)
    for i in range(n):
        if i not in surv:
            src[i] = int(rng.choice(surv))
    for c, p in (forced or {}).items():
        assert p in surv and c not in surv
        src[c] = p
    return src


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["immediate", "candidates", "vp"])
def test_device_resampling_cycles_match_oracle(pkg, ob, sc, mode):
    """Predict / update / resample cycles with random plans (chains of lower-slot parents several levels deep after a few
    resamplings: ids != slots), TWO predicts after some resamplings, both models: maps, unused lists, FOV counts, candidate
    lists (supports, checks, means, covariances) and ids against the oracle after every step."""
    rng = np.random.default_rng(17)
    n = 24
    if mode == "vp":
        scen = sc.make_vp_scenario(n, 10, 7, seed=3) if hasattr(sc, "make_vp_scenario") else None
        if scen is None:
            pytest.skip("no Victoria Park scenario generator")
        dev, orc = _pair(pkg, ob, n, cap=192, model=pkg.capi.MODEL_VICTORIAPARK_3D)
    else:
        scen = sc.make_scenario(n, 8, 9, seed=21)
        dev, orc = _pair(pkg, ob, n, cap=192)
    for f in (dev, orc):
        sc.load_scenario(f, scen)
        if mode != "immediate":
            cfg = f.get_filter_config()
            cfg.birthGaussianMeasurementCountThreshold = 3
            cfg.birthGaussianMeasurementCheckThreshold = 2
            cfg.birthGaussianMeasurementSupportDist = 2.0
            cfg.birthGaussianCurrentMeasurementCountThreshold = 0
            f.set_filter_config(cfg)
    max_level = 0
    n_res = 0
    for step in range(10):
        Z = scen["Z"].copy()
        Z[:, 0] += rng.normal(0, 2e-3, Z.shape[0])
        if step % 3 == 1:                                   # clutter that nobody has seen: unused measurements, candidates
            Z[-2:, 0] = rng.uniform(1.0, 2.0, 2)
        poses = scen["poses"] + rng.normal(0, 1e-3, scen["poses"].shape)
        for f in (dev, orc):
            f.predict_map(True)
            if step % 4 == 2:
                f.predict_map(True)                         # a second predict while resampleOccured_ is still set
        _compare_all(sc, dev, orc, n)
        for f in (dev, orc):
            f.set_poses(poses, scen["pose_cov"])
            f.update(Z)
            s = f.weight_sums()
            f.normalize_weights(s[0])
        _compare_all(sc, dev, orc, n)
        if step % 2 == 0 or step == 7:
            # the first two plans build a chain: slot 20 <- 10, then slot 10 <- 2 while slot 20 (id 10) survives: afterwards
            # idParent_[20] = 10 < 20 and idParent_[10] = 2 < 10, two levels of lower-slot parents
            if n_res == 0:
                src = _random_plan(rng, n, survivors=[2, 5, 10, 17], forced={20: 10})
            elif n_res == 1:
                src = _random_plan(rng, n, survivors=[2, 7, 20], forced={10: 2})
            else:
                src = _random_plan(rng, n)
            n_res += 1
            for f in (dev, orc):
                f.resample_apply(src)
            ids, par = dev.get_particle_ids()
            lvl = np.zeros(n, dtype=int)
            for i in range(n):
                lvl[i] = 0 if par[i] >= i else lvl[par[i]] + 1
            max_level = max(max_level, int(lvl.max()))
            _compare_all(sc, dev, orc, n)
    assert max_level >= 2          # the level-ordered walk was exercised beyond "copy from a survivor"
    assert np.any(dev.get_particle_ids()[0] != np.arange(n))


@pytest.mark.gpu
def test_external_mode_refuses_an_unacknowledged_birth_predict(pkg, sc):
    """in RFSGPU_INHERIT_EXTERNAL the HOST owns the reference's slot-ordered copy of the per-slot birth lists
    (include/RBPHDFilter.hpp:1005-1011).  A birth predict right after a resampling that reaches the engine without the host having
    applied the rule (rfsgpu_set_unused_masks, rfsgpu_predict_map_level, or the mode set again as an acknowledgement) would
    silently skip the inheritance: it is refused; after the acknowledgement it runs; the Python multi-GPU wrapper gives the handle
    back in the mode it found.  ADVICE r4: READING the masks acknowledges nothing -- a host that only inspects them must not
    silence the guard."""
    n = 12
    scen = sc.make_scenario(n, 30, 8, seed=5)
    f = pkg.RBPHDFilter(n, gm_capacity=128)
    sc.load_scenario(f, scen)
    f.update(scen["Z"])
    f.set_birth_inheritance(pkg.capi.INHERIT_EXTERNAL)
    plan = np.arange(n, dtype=np.int32)
    plan[3] = 7
    f.resample_apply(plan)
    with pytest.raises(pkg.capi.EngineError) as e:
        f.predict_map(True)
    assert "RFSGPU_INHERIT_EXTERNAL" in str(e.value)
    f.predict_map(False)                                   # (no births: nothing to inherit, allowed)
    m = f.get_unused_masks()                               # the host has only LOOKED at the lists: still refused
    with pytest.raises(pkg.capi.EngineError):
        f.predict_map(True)
    with pytest.raises(pkg.capi.EngineError):              # (the one-submission cycle goes through the same guard)
        f.cycle_async(True, scen["Z"])
    f.set_unused_masks(m)                                  # ... and has written them back: the rule is the host's, acknowledged
    f.predict_map(True)
    f.set_birth_inheritance(pkg.capi.INHERIT_REFERENCE)
    sh = pkg.sharded.ShardedRBPHDFilter(f)
    assert f.get_birth_inheritance() == pkg.capi.INHERIT_EXTERNAL
    sh.close()
    assert f.get_birth_inheritance() == pkg.capi.INHERIT_REFERENCE
    f.close()
