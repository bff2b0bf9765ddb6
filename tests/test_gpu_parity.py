"""Parity of the HIP path (through the C ABI) against the oracle on identical seeded inputs (fp64).

Tolerances: normalised particle weights rel 1e-9; GM components matched as multisets on
(w, mu, Sigma) rel 1e-10 / abs 1e-12; integer outputs (mixture sizes, unused lists, FOV counts) exact.
The oracle runs in its reference mode (oracle/binding.py: stable_sort=False): equal weights come out in the order
libstdc++'s std::sort leaves them in, which the device reproduces (csrc/stdsort_replay.h)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

WEIGHT_RTOL = 1e-9
GM_RTOL, GM_ATOL = 1e-10, 1e-12


def make_pair(pkg, ob, sc, scen, cap=512):
    dev = pkg.RBPHDFilter(scen["n"], device_id=0, gm_capacity=cap)
    orc = ob.OracleFilter(scen["n"])
    for f in (dev, orc):
        sc.load_scenario(f, scen)
    return dev, orc


def compare_maps(sc, dev, orc, n, ordered=False):
    assert np.array_equal(dev.gm_sizes(), orc.gm_sizes())
    for i in range(n):
        sc.assert_gm_close(dev.export_gm(i), orc.export_gm(i), GM_RTOL, GM_ATOL, ordered=ordered)


def compare_weights(dev, orc):
    wd, wo = dev.get_weights(), orc.get_weights()
    assert np.all(np.isfinite(wd))
    np.testing.assert_allclose(wd / wd.sum(), wo / wo.sum(), rtol=WEIGHT_RTOL, atol=1e-300)
    # un-normalised too (same scale)
    np.testing.assert_allclose(wd, wo, rtol=1e-8, atol=0)


SCENARIOS = [
    dict(n_particles=16, n_landmarks=12, n_z=5, seed=1),
    dict(n_particles=33, n_landmarks=70, n_z=19, seed=2),          # ragged: 70 = 64 + 6
    dict(n_particles=64, n_landmarks=200, n_z=30, seed=3),         # C2 shape, fewer particles
    dict(n_particles=24, n_landmarks=130, n_z=30, seed=4, frac_in_fov=0.25),   # C2b shape: most landmarks out of FOV
    dict(n_particles=8, n_landmarks=64, n_z=64, seed=5),           # max measurements
    dict(n_particles=20, n_landmarks=1, n_z=3, seed=6),
]


@pytest.mark.parametrize("kw", SCENARIOS)
def test_update_map_phase(pkg, ob, sc, kw):
    scen = sc.make_scenario(**kw)
    dev, orc = make_pair(pkg, ob, sc, scen)
    dev.update_map(scen["Z"])
    orc.update_map(scen["Z"])
    compare_maps(sc, dev, orc, scen["n"], ordered=True)   # append order is (m,z) row-major: exact order
    for i in range(scen["n"]):
        assert np.array_equal(dev.get_unused(i), orc.get_unused(i))
        assert dev.landmarks_in_fov(i) == orc.landmarks_in_fov(i)
        # w_prev: old weight for the nM old Gaussians, 0 for the appended ones
        np.testing.assert_allclose(dev.export_gm(i)[1], orc.export_gm(i)[1], rtol=0, atol=0)


@pytest.mark.parametrize("fused", [False, True])
def test_new_gaussians_whose_normalised_weight_underflows_are_dropped_in_order(pkg, ob, sc, fused):
    """RBPHDFilter.hpp:677 keeps a new Gaussian only if its normalised weight is > 0.  The device writes every survivor of the
    gates to its slab slot first and compacts afterwards when some weight turns out not to be positive -- the rare path, forced
    here: a huge clutter intensity and a share of landmarks with weights near the bottom of the fp64 range make v / normaliser
    underflow to exactly 0 for some (not all) of the appended Gaussians.  Order and content must stay the reference's."""
    scen = sc.make_scenario(24, 90, 20, seed=77, params=dict(clutter=1e10), weights=(0.5, 1.0))
    rng = np.random.default_rng(4)
    tiny = rng.random(scen["w"].shape) < 0.4
    scen["w"][tiny] = 1e-322
    dev, orc = make_pair(pkg, ob, sc, scen)
    if fused:
        dev.update_async(scen["Z"]); dev.synchronize()
        orc.update(scen["Z"])
        compare_maps(sc, dev, orc, scen["n"], ordered=True)
        return
    before = dev.gm_sizes().copy()
    dev.update_map(scen["Z"])
    orc.update_map(scen["Z"])
    compare_maps(sc, dev, orc, scen["n"], ordered=True)
    for i in range(scen["n"]):
        assert np.array_equal(dev.get_unused(i), orc.get_unused(i))
    assert (dev.gm_sizes() > before).any()          # some new Gaussians survive ...
    # ... and some were dropped: an update of the same state with ordinary clutter appends strictly more
    scen2 = dict(scen); scen2["params"] = dict(scen["params"], clutter=1e-4)
    dev2, _ = make_pair(pkg, ob, sc, scen2)
    dev2.update_map(scen["Z"])
    assert (dev2.gm_sizes() > dev.gm_sizes()).any()


@pytest.mark.parametrize("kw", SCENARIOS)
def test_all_phases_stepwise(pkg, ob, sc, kw):
    scen = sc.make_scenario(**kw)
    dev, orc = make_pair(pkg, ob, sc, scen)
    for f in (dev, orc):
        f.update_map(scen["Z"])
        f.importance_weighting()
    compare_weights(dev, orc)
    compare_maps(sc, dev, orc, scen["n"], ordered=True)   # sorted by weight
    for f in (dev, orc):
        f.merge()
    compare_maps(sc, dev, orc, scen["n"], ordered=True)
    for f in (dev, orc):
        f.prune()
    compare_maps(sc, dev, orc, scen["n"], ordered=True)


@pytest.mark.parametrize("kw", SCENARIOS[:4])
def test_update_fused_call_and_normalise(pkg, ob, sc, kw):
    scen = sc.make_scenario(**kw)
    dev, orc = make_pair(pkg, ob, sc, scen)
    for f in (dev, orc):
        f.update(scen["Z"])
    compare_weights(dev, orc)
    compare_maps(sc, dev, orc, scen["n"])
    sd, so = dev.weight_sums(), orc.weight_sums()
    np.testing.assert_allclose(sd, so, rtol=1e-9)
    dev.normalize_weights(sd[0])
    orc.normalize_weights(so[0])
    np.testing.assert_allclose(dev.get_weights(), orc.get_weights(), rtol=WEIGHT_RTOL)
    assert abs(dev.get_weights().sum() - 1) < 1e-12


def test_update_is_one_fused_launch_unless_phase_timing_is_asked_for(pkg, sc):
    """rfsgpu_update (what the reference-side binding calls): one fused launch by default, the whole step booked under
    TimingInfo::mapUpdate; with rfsgpu_set_phase_timing the phases run as separate launches and fill their own buckets.
    Same maps and weights, bit for bit."""
    scen = sc.make_scenario(40, 90, 12, seed=77)
    out = []
    for phases in (False, True):
        f = pkg.RBPHDFilter(scen["n"], gm_capacity=256)
        sc.load_scenario(f, scen)
        if phases:
            f.set_phase_timing(True)
        f.reset_timing()
        f.update(scen["Z"])
        t = f.getTimingInfo()
        ns = f.last_kernel_ns()
        if phases:
            assert t.mapUpdate_wall > 0 and t.particleWeighting_wall > 0 and t.mapMerge_wall > 0 and ns[1] > 0 and ns[2] > 0
        else:
            assert t.mapUpdate_wall > 0 and t.particleWeighting_wall == 0 and t.mapMerge_wall == 0 and ns[0] > 0 and ns[1] == 0 and ns[2] == 0
        out.append((f.get_weights(), [f.export_gm(i) for i in range(scen["n"])]))
    assert np.array_equal(out[0][0], out[1][0])
    for ga, gb in zip(out[0][1], out[1][1]):
        for x, y in zip(ga, gb):
            assert np.array_equal(x, y)


@pytest.mark.parametrize("fused", ["1", "0"])
@pytest.mark.parametrize("kw", SCENARIOS[:5] + [dict(n_particles=9, n_landmarks=40, n_z=7, seed=8, use_cluster=1)])
def test_update_async_matches_oracle(pkg, ob, sc, kw, fused, monkeypatch):
    """rfsgpu_update_async (stream-ordered steps; one fused kernel per step when RFSGPU_FUSED_STEP != 0, the three
    stand-alone kernels otherwise): same results as the oracle's update, and bit-identical to the synchronous call."""
    monkeypatch.setenv("RFSGPU_FUSED_STEP", fused)
    scen = sc.make_scenario(**kw)
    dev, orc = make_pair(pkg, ob, sc, scen)
    ref, _ = make_pair(pkg, ob, sc, scen)
    dev.update_async(scen["Z"])
    dev.synchronize()
    orc.update(scen["Z"])
    ref.update(scen["Z"])
    compare_weights(dev, orc)
    compare_maps(sc, dev, orc, scen["n"], ordered=True)
    assert np.array_equal(dev.get_weights(), ref.get_weights())
    for i in range(scen["n"]):
        for a, b in zip(dev.export_gm(i), ref.export_gm(i)):
            assert np.array_equal(a, b)
        assert np.array_equal(dev.get_unused(i), orc.get_unused(i))
    # a second step on the evolved state (slab parity after a fused step).  The weights are normalised in between as a filter
    # loop does (RBPHDFilter.hpp:537-539): the raw product of two steps underflows to 0 in the 64-measurement scenario, and a
    # comparison of 0 / 0 would check nothing.
    for f in (dev, orc):
        f.normalize_weights(f.weight_sums()[0])
        f.predict_map(True)
    dev.update_async(scen["Z"])
    dev.synchronize()
    orc.update(scen["Z"])
    assert np.all(dev.get_weights() > 0) and np.all(orc.get_weights() > 0)
    compare_weights(dev, orc)
    compare_maps(sc, dev, orc, scen["n"], ordered=True)


@pytest.mark.parametrize("fused", ["1", "0"])
@pytest.mark.parametrize("normalize", [True, False])
def test_step_async_sums_and_normalises_in_the_post_kernel(pkg, ob, sc, fused, normalize, monkeypatch):
    """rfsgpu_step_async: the step's post kernel (Murty partitions, queue reset, weight sums, optional division) gives the
    update_async + weight_sums + normalize_weights results, over several steps, with and without Murty partitions."""
    monkeypatch.setenv("RFSGPU_FUSED_STEP", fused)
    for kw in (dict(n_particles=48, n_landmarks=60, n_z=14, seed=3), dict(n_particles=24, n_landmarks=60, n_z=30, seed=21, n_eval=25, weighting_md=10.0, weights=(0.8, 1.0))):
        scen = sc.make_scenario(**kw)
        dev, orc = make_pair(pkg, ob, sc, scen)
        ref, _ = make_pair(pkg, ob, sc, scen)
        for step in range(3):
            if step:
                for f in (dev, ref, orc):
                    f.predict_map(True)
            dev.step_async(scen["Z"], normalize)
            dev.synchronize()
            ref.update_async(scen["Z"])
            s_ref = ref.weight_sums()
            orc.update(scen["Z"])
            s_orc = orc.weight_sums()
            if kw.get("weighting_md"):
                assert orc.murty_calls() > 0
            ptr = dev.weight_sums_device_ptr()
            import ctypes
            import torch
            sums = torch.as_tensor(pkg.sharded._DevArray(ptr, 2), device="cuda").cpu().numpy()
            np.testing.assert_allclose(sums, s_ref, rtol=1e-13)
            np.testing.assert_allclose(sums, s_orc, rtol=1e-8)
            if normalize:
                ref.normalize_weights(s_ref[0])
                orc.normalize_weights(s_orc[0])
                assert abs(dev.get_weights().sum() - 1) < 1e-12
            np.testing.assert_allclose(dev.get_weights(), ref.get_weights(), rtol=1e-13)
            np.testing.assert_allclose(dev.get_weights(), orc.get_weights(), rtol=1e-8)
            for i in range(scen["n"]):
                for a, b in zip(dev.export_gm(i), ref.export_gm(i)):
                    assert np.array_equal(a, b)
        dev.close(); ref.close(); orc.close()


def test_exact_partition_mode(pkg, ob, sc):
    """RFSGPU_PARTITION_EXACT (opt-in, not the reference's numbers): partitions with nR + nC > 8 by the wave-wide subset recurrence
    instead of Murty-200.  Device == the oracle's independent dynamic programme; and the default mode's weights (Murty, 200-term
    truncation) stay within a loose factor of the exact ones -- the truncation only drops small terms."""
    scen = sc.make_scenario(24, 60, 30, seed=21, n_eval=25, weighting_md=10.0, weights=(0.8, 1.0))
    dev, orc = make_pair(pkg, ob, sc, scen)
    dev.set_partition_mode(True)
    orc.set_partition_mode(True)
    dev.update(scen["Z"])
    orc.update(scen["Z"])
    assert orc.murty_calls() == 0            # every large partition went through the exact path
    compare_weights(dev, orc)
    compare_maps(sc, dev, orc, scen["n"])
    dflt, odef = make_pair(pkg, ob, sc, scen)
    dflt.update(scen["Z"])
    odef.update(scen["Z"])
    assert odef.murty_calls() > 0
    we, wm = dev.get_weights(), dflt.get_weights()
    assert np.all(wm <= we * (1 + 1e-9))     # a truncated sum of positive terms never exceeds the full one
    assert np.all(wm >= we * 0.5)
    # the fused / asynchronous path takes the same branch
    dev2, _ = make_pair(pkg, ob, sc, scen)
    dev2.set_partition_mode(True)
    dev2.update_async(scen["Z"])
    dev2.synchronize()
    np.testing.assert_array_equal(dev2.get_weights(), we)


@pytest.mark.parametrize("kw", [SCENARIOS[1], SCENARIOS[2], SCENARIOS[3], dict(n_particles=12, n_landmarks=500, n_z=30, seed=9, rmax=5.0),
                                dict(n_particles=9, n_landmarks=40, n_z=7, seed=8, use_cluster=1)])
@pytest.mark.parametrize("wpp", [2, 3])
def test_fused_step_with_two_and_three_waves_per_particle(pkg, ob, sc, kw, wpp, monkeypatch):
    """The fused step kernel's two forms, forced with RFSGPU_STEP_WPP (the engine picks three waves per particle when the two-wave grid
    cannot be resident at once -- the configs[2] shard -- and, since round 5, when the launch is small enough for every three-wave
    workgroup to be resident; two waves otherwise, e.g. configs[1]): bit-identical to the synchronous three-kernel path over two steps."""
    monkeypatch.setenv("RFSGPU_STEP_WPP", str(wpp))
    scen = sc.make_scenario(**kw)
    cap = 704 if kw["n_landmarks"] >= 500 else 512
    dev, orc = make_pair(pkg, ob, sc, scen, cap=cap)
    ref, _ = make_pair(pkg, ob, sc, scen, cap=cap)
    for step in range(2):
        if step:
            for f in (dev, ref, orc):
                f.predict_map(True)
        dev.step_async(scen["Z"], False)
        dev.synchronize()
        ref.update(scen["Z"])
        orc.update(scen["Z"])
        assert np.array_equal(dev.get_weights(), ref.get_weights())
        for i in range(scen["n"]):
            for a, b in zip(dev.export_gm(i), ref.export_gm(i)):
                assert np.array_equal(a, b)
        compare_weights(dev, orc)
        compare_maps(sc, dev, orc, scen["n"], ordered=True)


def test_cluster_process_weighting(pkg, ob, sc):
    scen = sc.make_scenario(32, 90, 20, seed=11, use_cluster=True)
    dev, orc = make_pair(pkg, ob, sc, scen)
    for f in (dev, orc):
        f.update(scen["Z"])
    compare_weights(dev, orc)
    compare_maps(sc, dev, orc, scen["n"])


def test_empty_measurement_set_is_a_no_op(pkg, ob, sc):
    scen = sc.make_scenario(8, 20, 6, seed=12)
    dev, orc = make_pair(pkg, ob, sc, scen)
    before = [dev.export_gm(i) for i in range(8)]
    dev.update(np.zeros((0, 2)))
    for i in range(8):
        sc.assert_gm_close(dev.export_gm(i), before[i], 0, 0, ordered=True)
    assert np.array_equal(dev.get_weights(), np.ones(8))


def test_empty_maps(pkg, ob, sc):
    scen = sc.make_scenario(8, 5, 6, seed=13)
    dev = pkg.RBPHDFilter(8, gm_capacity=64)
    orc = ob.OracleFilter(8)
    for f in (dev, orc):
        sc.load_scenario(f, scen, maps=False)
        f.update(scen["Z"])
    assert np.array_equal(dev.gm_sizes(), np.zeros(8, np.int32))
    for i in range(8):
        assert np.array_equal(dev.get_unused(i), np.arange(6))
    np.testing.assert_array_equal(dev.get_weights(), orc.get_weights())   # denorm_min for every particle


def test_multi_step_predict_update_cycle(pkg, ob, sc):
    """birth -> static step -> update -> normalise, several steps, same measurements stream on both sides."""
    scen = sc.make_scenario(16, 25, 10, seed=14)
    dev, orc = make_pair(pkg, ob, sc, scen, cap=256)
    rng = np.random.default_rng(99)
    for step in range(6):
        Z = scen["Z"] + rng.normal(0, 1e-3, scen["Z"].shape)
        for f in (dev, orc):
            f.predict_map(True)
            f.update(Z)
            s = f.weight_sums()
            f.normalize_weights(s[0])
        np.testing.assert_allclose(dev.get_weights(), orc.get_weights(), rtol=1e-8)
        compare_maps(sc, dev, orc, scen["n"])


@pytest.mark.parametrize("model", ["2d", "vp"])
def test_state_ring_steps_equal_restore_and_step(pkg, ob, sc, model):
    """bench.py's re-seeding (round 5): a ring of pre-seeded copies of the saved state, taken one per step by a pointer swap
    (rfsgpu_state_ring_create / _next / _seed), instead of rfsgpu_restore_state inside the timed step.  Every ring step must be the
    restore + step it replaces -- weights, sizes, maps, unused lists bit for bit -- and agree with the oracle; a consumed ring refuses
    the next swap, a re-seeded one serves again; the handle works on after the ring is freed."""
    vp = model == "vp"
    kw = dict(model=pkg.capi.MODEL_VICTORIAPARK_3D) if vp else {}
    scen = sc.make_vp_scenario(12, 30, 9, seed=77, scan="ragged") if vp else sc.make_scenario(12, 60, 14, seed=77)
    dev = pkg.RBPHDFilter(scen["n"], gm_capacity=256, **kw)
    orc = ob.OracleFilter(scen["n"], **kw)
    for f in (dev, orc):
        sc.load_scenario(f, scen)
    dev.save_state()

    def snapshot():
        dev.synchronize()
        return (dev.get_weights().copy(), dev.gm_sizes().copy(), [tuple(a.copy() for a in dev.export_gm(i)) for i in range(scen["n"])],
                [np.array(dev.get_unused(i)) for i in range(scen["n"])])

    def same(a, b):
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
        for ga, gb in zip(a[2], b[2]):
            for x, y in zip(ga, gb):
                np.testing.assert_array_equal(x, y)
        for ua, ub in zip(a[3], b[3]):
            np.testing.assert_array_equal(ua, ub)

    dev.restore_state()
    dev.step_async(scen["Z"], False)
    want = snapshot()
    orc.update(scen["Z"])
    compare_weights(dev, orc)
    compare_maps(sc, dev, orc, scen["n"])
    dev.state_ring_create(3)
    for _ in range(3):
        dev.state_ring_next()
        dev.step_async(scen["Z"], False)
        same(snapshot(), want)
    with pytest.raises(Exception, match="consumed"):
        dev.state_ring_next()
    dev.state_ring_seed()
    for _ in range(2):
        dev.state_ring_next()
        dev.step_async(scen["Z"], False)
        same(snapshot(), want)
    dev.restore_state()                       # the two ways of re-seeding mix
    dev.step_async(scen["Z"], False)
    same(snapshot(), want)
    dev.state_ring_create(0)
    with pytest.raises(Exception, match="no ring"):
        dev.state_ring_next()
    dev.restore_state()
    dev.step_async(scen["Z"], False)
    same(snapshot(), want)


@pytest.mark.parametrize("n_lm,cap", [(25, 256), (150, 384)])
def test_fused_predict_update_cycle_is_the_call_by_call_cycle(pkg, ob, sc, n_lm, cap):
    """rfsgpu_cycle_async (round 5): the predict's map part (births at the poses the PREVIOUS update used,
    Sigma += Q; include/RBPHDFilter.hpp:415-442) at the head of the fused step kernel, new poses and weights arriving with the
    call.  Against (a) the same device doing predict_map / set_poses / set_weights / step_async call by call -- maps, unused
    lists and weights BIT FOR BIT -- and (b) the oracle, through six cycles in which the poses move every cycle (the births of
    cycle k must come out at the poses of update k-1, not at the new ones), a cycle without births, one without a predict,
    and a resampling with foreign parents in between (the inheritance walk: the cycle falls back to the stand-alone kernels and
    must still agree), followed by cycles that take the fused head again."""
    scen = sc.make_scenario(16, n_lm, 12, seed=140 + n_lm)
    fus, orc = make_pair(pkg, ob, sc, scen, cap=cap)
    ref = pkg.RBPHDFilter(scen["n"], gm_capacity=cap)
    sc.load_scenario(ref, scen)
    rng = np.random.default_rng(7)
    poses = scen["poses"].copy()
    saw = set()
    for f in (fus, ref, orc):                  # a first update, so that there are unused measurements to give birth from
        f.update(scen["Z"])
    for step in range(8):
        Z = scen["Z"] + rng.normal(0, 2e-3, scen["Z"].shape)
        Z[-2:, 0] = rng.uniform(1.0, 3.0, 2)                        # two measurements nobody explains -> births next cycle
        poses = poses + rng.normal(0, 0.02, poses.shape)            # the host's propagate
        w_in = rng.uniform(0.5, 1.0, scen["n"])
        predict = None if step == 5 else (False if step == 2 else True)
        fus.cycle_async(predict, Z, poses=poses, weights=w_in, normalize=False)
        saw.add(fus.last_step_variant()[3])
        for f in (ref, orc):
            if predict is not None:
                f.predict_map(bool(predict))
            f.set_poses(poses)
            f.set_weights(w_in)
        ref.step_async(Z, False)
        orc.update(Z)
        fus.synchronize(); ref.synchronize()
        np.testing.assert_array_equal(fus.get_weights(), ref.get_weights())
        assert np.array_equal(fus.gm_sizes(), ref.gm_sizes())
        for i in range(scen["n"]):
            for a, b in zip(fus.export_gm(i), ref.export_gm(i)):
                np.testing.assert_array_equal(a, b)
            assert np.array_equal(fus.get_unused(i), ref.get_unused(i))
        compare_weights(fus, orc)
        compare_maps(sc, fus, orc, scen["n"])
        if step == 3:                                                # a resampling whose plan has foreign parents
            for f in (fus, ref, orc):
                s = f.weight_sums()
                f.normalize_weights(s[0])
            fired, wn, src = ob.resample_decide(orc.get_weights(), scen["n"] + 1.0, 0.41)
            assert fired and np.any(src != np.arange(scen["n"]))
            for f in (fus, ref, orc):
                f.resample_apply(src)
            poses = poses[src]
    assert saw == {2, 3}, saw          # both forms ran: the predict at the head (2) and the fall-back to the stand-alone predict kernel (3: the head only pulls the inputs)
    ref.close()


def test_update_io_is_set_poses_set_weights_update_get_weights(pkg, ob, sc):
    """rfsgpu_update_io: RBPHDFilter::update with its inputs and outputs in one synchronous call == the four
    calls the binding used to make, bit for bit, with and without the predict folded in; device errors come back from THIS call."""
    scen = sc.make_scenario(12, 40, 12, seed=61)
    a = pkg.RBPHDFilter(scen["n"], gm_capacity=192)
    b = pkg.RBPHDFilter(scen["n"], gm_capacity=192)
    for f in (a, b):
        sc.load_scenario(f, scen)
    rng = np.random.default_rng(3)
    poses = scen["poses"].copy()
    for step in range(4):
        Z = scen["Z"] + rng.normal(0, 2e-3, scen["Z"].shape)
        poses = poses + rng.normal(0, 0.01, poses.shape)
        cov = np.tile(np.diag([1e-4, 1e-4, 1e-6]), (scen["n"], 1, 1)) * (1 + step)
        w_in = rng.uniform(0.5, 1.0, scen["n"])
        pred = None if step == 0 else True
        w_a = a.update_io(Z, predict=pred, poses=poses, pose_cov=cov, weights=w_in)
        if pred:
            b.predict_map(True)
        b.set_poses(poses, cov)
        b.set_weights(w_in)
        b.update(Z)
        np.testing.assert_array_equal(w_a, b.get_weights())
        for i in range(scen["n"]):
            for x, y in zip(a.export_gm(i), b.export_gm(i)):
                np.testing.assert_array_equal(x, y)
    # a filter whose updates queue Murty partitions (first the light post-kernel instance, then the eight-wave one): the weights the
    # post kernel delivers to the pinned landing area carry the Murty factors -- same weights as rfsgpu_update's
    mscen = sc.make_scenario(16, 200, 50, seed=555, n_clutter=10, n_eval=40, weighting_md=10.0, weights=(0.8, 1.0))
    m1 = pkg.RBPHDFilter(16, gm_capacity=448)
    m2 = pkg.RBPHDFilter(16, gm_capacity=448)
    for f in (m1, m2):
        sc.load_scenario(f, mscen)
    for rep in range(3):
        w1 = m1.update_io(mscen["Z"], weights=np.ones(16))
        m2.set_weights(np.ones(16))
        m2.update(mscen["Z"])
        np.testing.assert_array_equal(w1, m2.get_weights())
        assert np.ptp(w1) > 0
    m1.close(); m2.close()
    # rfsgpu_set_phase_timing (the binding's RFSGPU_PHASE_TIMING=1) is honoured by the one-call form too: separate launches, per-phase buckets
    p1 = pkg.RBPHDFilter(scen["n"], gm_capacity=192)
    sc.load_scenario(p1, scen)
    p1.set_phase_timing(True)
    p1.reset_timing()
    w_p = p1.update_io(scen["Z"], poses=scen["poses"], weights=np.ones(scen["n"]))
    t = p1.getTimingInfo()
    assert t.mapUpdate_wall > 0 and t.particleWeighting_wall > 0 and t.mapMerge_wall > 0
    p2 = pkg.RBPHDFilter(scen["n"], gm_capacity=192)
    sc.load_scenario(p2, scen)
    np.testing.assert_array_equal(w_p, p2.update_io(scen["Z"], poses=scen["poses"], weights=np.ones(scen["n"])))
    p1.close(); p2.close()
    small = pkg.RBPHDFilter(4, gm_capacity=64)
    sc.load_scenario(small, sc.make_scenario(4, 60, 30, seed=16))
    with pytest.raises(pkg.capi.EngineError) as e:
        small.update_io(sc.make_scenario(4, 60, 30, seed=16)["Z"])
    assert e.value.status == pkg.capi.ERR_CAPACITY
    a.close(); b.close(); small.close()


def test_resample_apply(pkg, ob, sc):
    scen = sc.make_scenario(16, 20, 8, seed=15)
    dev, orc = make_pair(pkg, ob, sc, scen, cap=128)
    for f in (dev, orc):
        f.update(scen["Z"])
        s = f.weight_sums()
        f.normalize_weights(s[0])
    fired, wn, src = ob.resample_decide(orc.get_weights(), 17.0, 0.37)
    assert fired
    assert np.array_equal(src, pkg.engine.systematic_resample_plan(dev.get_weights(), 0.37))
    dev.resample_apply(src)
    orc.resample_apply(src)
    compare_maps(sc, dev, orc, 16, ordered=True)
    assert np.array_equal(dev.get_weights(), np.ones(16))
    for i in range(16):
        assert np.array_equal(dev.get_unused(i), orc.get_unused(i))


def test_capacity_overflow_is_reported(pkg, sc):
    scen = sc.make_scenario(4, 60, 30, seed=16)
    dev = pkg.RBPHDFilter(4, gm_capacity=64)
    sc.load_scenario(dev, scen)
    with pytest.raises(pkg.capi.EngineError) as e:
        dev.update(scen["Z"])
    assert e.value.status == pkg.capi.ERR_CAPACITY


def test_async_edge_cases(pkg, ob, sc):
    """The stream-ordered (fused-kernel) path on the edge cases of the synchronous one: capacity overflow surfaces at the next
    synchronisation, an empty measurement set is a no-op, empty maps give every particle the denorm-min weight, and a run of
    pipelined steps without intermediate syncs equals the same steps done synchronously."""
    scen = sc.make_scenario(4, 60, 30, seed=16)
    dev = pkg.RBPHDFilter(4, gm_capacity=64)
    sc.load_scenario(dev, scen)
    dev.update_async(scen["Z"])                      # no error yet: nothing has been waited for
    with pytest.raises(pkg.capi.EngineError) as e:
        dev.synchronize()
    assert e.value.status == pkg.capi.ERR_CAPACITY

    scen = sc.make_scenario(8, 5, 6, seed=13)
    dev = pkg.RBPHDFilter(8, gm_capacity=64)
    orc = ob.OracleFilter(8)
    for f in (dev, orc):
        sc.load_scenario(f, scen, maps=False)
    dev.update_async(np.zeros((0, 2)))
    dev.synchronize()
    np.testing.assert_array_equal(dev.get_weights(), np.ones(8))
    dev.update_async(scen["Z"])
    dev.synchronize()
    orc.update(scen["Z"])
    assert np.array_equal(dev.gm_sizes(), np.zeros(8, np.int32))
    np.testing.assert_array_equal(dev.get_weights(), orc.get_weights())

    scen = sc.make_scenario(16, 40, 12, seed=19)
    a, _ = make_pair(pkg, ob, sc, scen)
    b, _ = make_pair(pkg, ob, sc, scen)
    rng = np.random.default_rng(1)
    Zs = [scen["Z"] + rng.normal(0, 1e-3, scen["Z"].shape) for _ in range(6)]
    for Z in Zs:                                     # pipelined: predict + update + normalise, never waiting
        a.predict_map(True)
        a.update_async(Z)
        a.weight_sums_async()
        a.normalize_weights(0.0, a.weight_sums_device_ptr())
    a.synchronize()
    for Z in Zs:
        b.predict_map(True)
        b.update(Z)
        s = b.weight_sums()
        b.normalize_weights(s[0])
    np.testing.assert_array_equal(a.get_weights(), b.get_weights())
    for i in range(scen["n"]):
        for x, y in zip(a.export_gm(i), b.export_gm(i)):
            np.testing.assert_array_equal(x, y)
    avg, nsteps = a.kernel_time_stats()
    assert nsteps == 0 or avg[0] > 0                 # (statistics were harvested by synchronize())


def test_get_landmark_and_bad_indices(pkg, sc):
    scen = sc.make_scenario(4, 10, 4, seed=17)
    dev = pkg.RBPHDFilter(4, gm_capacity=64)
    sc.load_scenario(dev, scen)
    assert dev.getGMSize(0) == 10 and dev.getGMSize(4) == -1 and dev.getGMSize(-1) == -1
    mean, cov, w = dev.getLandmark(2, 3)
    np.testing.assert_array_equal(mean, scen["mean"][2, 3])
    assert w == scen["w"][2, 3]
    assert cov[0, 1] == cov[1, 0] == scen["cov"][2, 3, 0, 1]
    assert dev.getLandmark(2, 10) is None and dev.getLandmark(9, 0) is None


def test_mat_perm_known_answers_on_device(pkg, ob):
    from test_oracle_combinatorics import DERANGEMENTS
    for n, want in DERANGEMENTS.items():
        got = pkg.mat_perm(np.ones((n, n)) - np.eye(n))[0]
        assert got == want, (n, got, want)
    rng = np.random.default_rng(0)
    for n in (1, 2, 5, 9, 10, 12, 13, 16, 17):     # 10 .. : the register-resident kernels mat_perm_kernel_t<12 | 16 | 20 | 24>
        A = rng.uniform(-1, 1, (7, n, n))
        np.testing.assert_allclose(pkg.mat_perm(A), ob.mat_perm(A), rtol=1e-10, atol=1e-13)
    for n in (20, 21, 24):
        # identity: every x_i is -1/2, 1/2 or 3/2, every term a multiple of 2^-n below 2^53 in magnitude -- any summation order is exact
        assert pkg.mat_perm(np.eye(n))[0] == 1.0
        # a permutation matrix plus one more entry per row of a 4-cycle: permanent 2 (two perfect matchings), small half-integer x
        Pm = np.eye(n)[rng.permutation(n)]
        cyc = rng.choice(n, 4, replace=False)
        rows = [int(np.nonzero(Pm[:, c])[0][0]) for c in cyc]
        for k in range(4):
            Pm[rows[k], cyc[(k + 1) % 4]] = 1.0
        assert pkg.mat_perm(Pm)[0] == 2.0
    A = rng.uniform(0, 1, (3, 20, 20))              # (Ryser's sum cancels ~8 digits at n = 20 with entries in [0, 1])
    np.testing.assert_allclose(pkg.mat_perm(A), ob.mat_perm(A), rtol=1e-6)


def test_murty_partitions_match_oracle(pkg, ob, sc):
    """C5-style stress: wide weighting gate so partitions exceed nR+nC = 8 and the Murty-200 path runs."""
    scen = sc.make_scenario(24, 60, 30, seed=21, n_eval=25, weighting_md=10.0, weights=(0.8, 1.0))
    dev, orc = make_pair(pkg, ob, sc, scen)
    for f in (dev, orc):
        f.update_map(scen["Z"])
        f.importance_weighting()
    assert orc.murty_calls() > 0, "scenario does not reach the Murty path"
    compare_weights(dev, orc)


def test_murty_partitions_of_every_size_class_match_oracle(pkg, ob, sc):
    """60 measurements, 60 evaluation points, a 12-sigma weighting gate: partitions of extended dimension 9 ... 51 in one update
    (tools/murty_dims.py shows the histogram), i.e. every form of the device search -- the small form (<= 16: table in LDS,
    per-node constraint sets), sub-problems in the 20 x 20 LDS tiles, and sub-problems in the job's arena (> 20)."""
    scen = sc.make_scenario(16, 200, 60, seed=3, n_clutter=10, n_eval=60, weighting_md=12.0, weights=(0.8, 1.0))
    dev, orc = make_pair(pkg, ob, sc, scen, cap=768)
    for f in (dev, orc):
        f.update_map(scen["Z"])
        f.importance_weighting()
    assert orc.murty_calls() > 40, "scenario does not reach the Murty path"
    compare_weights(dev, orc)


def test_device_murty_sums_against_the_reference_bruteforce_fixture(pkg, ob):
    """The device's job kernel pinned to the reference where the path uses Murty: the extended tables of
    tests/golden/murty_extended_ranked.json (dimension 7 ... 9, built as include/RBPHDFilter.hpp:907-940 builds them, ranked and
    de-duplicated by the reference's BruteForceLinearAssignment, tests/golden/make_combinatorics_fixtures.py) go through
    murty_jobs_kernel as queued partitions; each partition sum must equal the sum of exp(score) over the fixture's <= 200 best
    distinct assignments (:948-959) -- and the oracle's Murty the same."""
    import json
    import math
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "murty_extended_ranked.json")) as fh:
        cases = json.load(fh)
    dev = pkg.RBPHDFilter(8, gm_capacity=64)
    w0 = dev.get_weights()
    for rep in range(2):                                    # (second pass: the capped six-wave instance, after the filter has shown Murty work)
        sums = dev.murty_partition_sums([np.array(c["C"]) for c in cases], [c["nR"] for c in cases], [c["nC"] for c in cases])
        for c, got in zip(cases, sums):
            want = 0.0
            for sc_ in c["scores"]:
                want += math.exp(sc_)
            assert abs(got - want) <= 1e-12 * want, (c["nR"], c["nC"], got, want)
            so, _ = ob.murty(np.array(c["C"]), c["nR"], c["nC"], kmax=200)
            wo = 0.0
            for sc_ in so[so >= -1000.0]:
                wo += math.exp(sc_)
            assert abs(got - wo) <= 1e-13 * wo
    np.testing.assert_array_equal(dev.get_weights(), w0)     # the hook leaves the weights alone
    dev.close()


def test_dense_intensity_switch_matches_the_sparse_sums(pkg, ob, sc, monkeypatch):
    """RFSGPU_DENSE_INTENSITY=1: a handle created under it adds every (evaluation point, Gaussian) term,
    as the reference does (include/RBPHDFilter.hpp:776-800) -- deviation 9 (terms below 2^-64 of their sum left out, mixtures of more
    than 128 Gaussians) switched off, so that a future reference-pinned fixture can be run both ways.  Both forms against the oracle,
    and against each other to an ulp of the weight."""
    scen = sc.make_scenario(24, 300, 30, seed=91)
    out = []
    for dense in (False, True):
        if dense:
            monkeypatch.setenv("RFSGPU_DENSE_INTENSITY", "1")
        dev, orc = make_pair(pkg, ob, sc, scen, cap=640)
        for f in (dev, orc):
            f.update(scen["Z"])
        compare_weights(dev, orc)
        compare_maps(sc, dev, orc, scen["n"])
        out.append(dev.get_weights())
        dev.close()
    np.testing.assert_allclose(out[0], out[1], rtol=1e-13)


def test_cpp_host_driver_end_to_end(pkg):
    """The C++ host mirror (rfs-slam_amd/host/rbphd_filter.hpp) driving the device path through the C ABI on the
    shipped C1 configuration (cfg values of the reference's rbphdslam2dSim.xml): the map must converge."""
    import os
    import re
    import subprocess
    exe = pkg.build_mod.build_host()
    cfg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "rbphdslam2dSim_c1.xml")
    out = subprocess.run([exe, "-c", cfg, "-t", "2", "-s", "2", "-n", "200"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    m = re.search(r"RESULT matched=(\d+) landmarks=(\d+) mean_err=([\d.]+) pose_err=([\d.]+)", out.stdout)
    assert m, out.stdout[-2000:]
    matched, total, mean_err, pose_err = int(m.group(1)), int(m.group(2)), float(m.group(3)), float(m.group(4))
    assert total == 50 and matched >= 40, out.stdout[-800:]
    assert mean_err < 0.3 and pose_err < 0.6


def test_cpp_host_driver_sharded_over_device_entries(pkg):
    """The C++ host mirror in its multi-GPU form (RBPHDFilter2d over rfsgpu_group_*: predict / update / normalise / global
    resampling with row migration, map access by global index) on the C1 configuration, the particle set cut into three shards
    that all live on this box's GPU: the filter must behave like the single-handle run of the same seeds."""
    import os
    import re
    import subprocess
    exe = pkg.build_mod.build_host()
    cfg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "rbphdslam2dSim_c1.xml")
    res = []
    for extra in ([], ["--devices", "0,0,0"], ["--devices", "0,0,0,0,0,0,0,0"]):       # (eight entries: configs[2]'s shard count)
        out = subprocess.run([exe, "-c", cfg, "-t", "2", "-s", "2", "-n", "160"] + extra, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        m = re.search(r"RESULT matched=(\d+) landmarks=(\d+) mean_err=([\d.]+) pose_err=([\d.]+)", out.stdout)
        assert m, out.stdout[-2000:]
        res.append((int(m.group(1)), float(m.group(3)), float(m.group(4))))
        if extra:
            assert "sharded over %d device entries" % (extra[1].count(",") + 1) in out.stdout
    (m1, e1, p1), (m3, e3, p3), (m8, e8, p8) = res
    assert m3 >= 40 and e3 < 0.3 and p3 < 0.6
    # same seeds, same arithmetic per particle; only the weight sums are added per shard (last-bit differences), so the runs
    # normally coincide to the printed precision
    assert abs(m1 - m3) <= 3 and abs(e1 - e3) < 0.1
    assert abs(m1 - m8) <= 3 and abs(e1 - e8) < 0.1


def ospa(est, truth, cutoff, order):
    """rfs::OSPA (reference include/OSPA.hpp:122-203): cost matrix min(|a - b|, c) padded with c to a square of n = max(n1, n2),
    optimal assignment (the reference runs its Hungarian method; scipy's solver finds the same optimum), (sum C^p / n)^(1/p)."""
    from scipy.optimize import linear_sum_assignment
    n1, n2 = len(est), len(truth)
    n = max(n1, n2)
    if n == 0:
        return 0.0
    Cm = np.full((n, n), float(cutoff))
    if n1 and n2:
        d = np.linalg.norm(np.asarray(est)[:, None, :] - np.asarray(truth)[None, :, :], axis=2)
        Cm[:n1, :n2] = np.minimum(d, cutoff)
    r, c = linear_sum_assignment(Cm)
    return float((np.sum(Cm[r, c] ** order) / n) ** (1.0 / order))


def test_cpp_host_driver_map_quality_ospa(pkg, tmp_path):
    """End-to-end SLAM quality of the C++ driver on the shipped C1 configuration, from its own log files (the reference's formats:
    landmarkEst.dat `t i mu_x mu_y S_xx S_xy S_yy w`, gtLandmark.dat `x y t_first_obs`): OSPA (include/OSPA.hpp, cutoff 0.5 m,
    order 1) of the final best-particle map (Gaussians with w >= 0.5) against the 50 ground-truth landmarks.  Yardstick: the
    survey's probe of the reference itself on this configuration mapped 48 of 50 landmarks with a mean error of 0.11 m
    (SURVEY 8(b)), i.e. OSPA = (48 * 0.11 + 2 * 0.5) / 50 = 0.126 m; the build must not be worse than 0.15 m."""
    import os
    import subprocess
    exe = pkg.build_mod.build_host()
    cfg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "rbphdslam2dSim_c1.xml")
    vals = []
    for t, sd in ((1, 2), (2, 1), (3, 2)):   # three realisations (the host's random streams are not the reference's boost streams:
        d = os.path.join(tmp_path, f"run_{t}_{sd}")   # a seed pair does not name the same run there and here); median judged
        os.makedirs(d)
        out = subprocess.run([exe, "-c", cfg, "-t", str(t), "-s", str(sd), "-n", "200", "-o", d], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        gt = np.loadtxt(os.path.join(d, "gtLandmark.dat"))[:, :2]
        est = np.loadtxt(os.path.join(d, "landmarkEst.dat"))
        last = est[est[:, 0] == est[:, 0].max()]
        strong = last[last[:, 7] >= 0.5][:, 2:4]
        assert gt.shape[0] == 50
        vals.append(ospa(strong, gt, 0.5, 1.0))
    assert np.median(vals) <= 0.15, vals
    # known answers of the metric itself
    assert ospa(gt, gt, 0.5, 1.0) == 0.0
    assert abs(ospa(gt[:48], gt, 0.5, 1.0) - 2 * 0.5 / 50) < 1e-12
    assert abs(ospa(gt + 0.03, gt, 0.5, 2.0) - 0.03 * np.sqrt(2)) < 1e-9


def test_cpp_fastslam_driver_end_to_end(pkg):
    """fastslam2d_sim: the same simulator around the C++ FastSLAM mirror (rfs_amd::FastSLAM2d -> rfsgpu_fastslam_update) on the
    values of the reference's cfg/fastslam2dSim.xml: the landmark map must converge."""
    import os
    import re
    import subprocess
    pkg.build_mod.build_host()
    exe = pkg.build_mod.SIM_FASTSLAM
    cfg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "fastslam2dSim_c1.xml")
    out = subprocess.run([exe, "-c", cfg, "-t", "2", "-s", "2", "-n", "200"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    m = re.search(r"RESULT matched=(\d+) landmarks=(\d+) mean_err=([\d.]+) pose_err=([\d.]+)", out.stdout)
    assert m, out.stdout[-2000:]
    matched, total, mean_err, pose_err = int(m.group(1)), int(m.group(2)), float(m.group(3)), float(m.group(4))
    assert total == 50 and matched >= 45, out.stdout[-800:]
    assert mean_err < 0.2 and pose_err < 0.6


def test_cpp_mhfastslam_driver_end_to_end(pkg):
    """fastslam2d_sim on the values of the reference's cfg/mhfastslam2dSim.xml (3 association hypotheses per particle): the
    particle set grows inside updates and is resampled back to its initial size by rfs_amd::FastSLAM2d; the map must converge."""
    import os
    import re
    import subprocess
    pkg.build_mod.build_host()
    exe = pkg.build_mod.SIM_FASTSLAM
    cfg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "mhfastslam2dSim_c1.xml")
    out = subprocess.run([exe, "-c", cfg, "-t", "2", "-s", "2", "-n", "100"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-500:]
    m = re.search(r"RESULT matched=(\d+) landmarks=(\d+) mean_err=([\d.]+) pose_err=([\d.]+)", out.stdout)
    assert m, out.stdout[-2000:]
    matched, total, mean_err, pose_err = int(m.group(1)), int(m.group(2)), float(m.group(3)), float(m.group(4))
    assert total == 50 and matched >= 45, out.stdout[-800:]
    assert mean_err < 0.2 and pose_err < 0.6


def test_cpp_host_driver_logs_through_analysis2d_sim(pkg, tmp_path):
    """f3 end to end: host/rbphdslam2d_sim writes the reference driver's log files (gtPose, gtLandmark with first-in-range
    times, odometry, measurement, deadReckoning, particlePose, landmarkEst), tools/analysis2d_sim.py (analysis2dSim,
    src/analysis2dSim.cpp) turns them into the reference's error files: the filter must beat dead reckoning by a wide margin,
    count the landmarks, and keep the COLA map error (cutoff 0.2 m: a landmark further off than that counts as missing) low."""
    import importlib.util
    import os
    import subprocess
    exe = pkg.build_mod.build_host()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(tmp_path, "run")
    os.makedirs(d)
    out = subprocess.run([exe, "-c", os.path.join(root, "tests", "golden", "rbphdslam2dSim_c1.xml"), "-t", "2", "-s", "2", "-n", "200", "-o", d],
                         capture_output=True, text=True, timeout=600)     # (a realisation on which the filter keeps track: see
    assert out.returncode == 0, out.stderr[-2000:]                         #  test_c1_full_run_device_and_oracle_agree_on_map_quality)
    spec = importlib.util.spec_from_file_location("analysis2d_sim", os.path.join(root, "tools", "analysis2d_sim.py"))
    a = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(a)
    dr, pose, mp = a.analyse(d + "/")
    assert pose.shape[0] == 2999 and dr.shape[0] == 2999 and mp.shape[0] == 2999
    assert pose[-1, 4] < 0.5 and pose[-1, 4] < dr[-1, 4] / 3 and pose[:, 4].mean() < 0.3
    assert int(mp[-1, 1]) == 50 and abs(mp[-1, 2] - 50) < 5          # landmarks in range so far, cardinality estimate
    assert mp[-1, 3] < 25 and mp[:, 3].mean() < 15                  # COLA: 0 = perfect, 50 = nothing mapped
    assert np.all(np.diff(mp[:, 1]) >= 0)                           # the observable set only grows
    for name, cols in (("poseEstError.dat", 5), ("deadReckoningError.dat", 5), ("landmarkEstError.dat", 4), ("gtPose.dat", 4), ("deadReckoning.dat", 4),
                       ("odometry.dat", 4), ("measurement.dat", 3), ("gtLandmark.dat", 3)):
        assert len(open(os.path.join(d, name)).readline().split()) == cols, name


@pytest.mark.parametrize("n_lm,cap", [(90, 256), (330, 512)])
def test_tied_weights_come_out_in_reference_order_across_chunks(pkg, ob, sc, n_lm, cap):
    """Equal prior weights (ordinary in real runs: all birth Gaussians share one weight): the reference's sort is
    std::sort (include/GaussianMixture.hpp:523-534), and the oracle runs it.  The device ranks with ties by index (64-entry
    chunks / buckets: ties inside a chunk and ties ACROSS chunks both matter -- 330 landmarks with a handful of distinct
    weights tie across six chunks) and then replays std::sort's partition phase over the tied runs (stdsort_replay.h):
    ORDERED comparison of maps and weights against the oracle."""
    scen = sc.make_scenario(16, n_lm, 14, seed=23)
    scen["w"][:, ::3] = 0.5                                  # many exact ties
    scen["w"][:, 1::3] = np.float64(np.float32(0.7)) + 1e-12 * np.arange(n_lm // 3)[None, :]   # distinct in fp64, equal in fp32
    if n_lm > 128:
        scen["w"][:, 2::3] = np.array([0.9, 0.5, 0.31])[np.arange(n_lm // 3) % 3][None, :]      # a few values, tied all over
    dev = pkg.RBPHDFilter(scen["n"], gm_capacity=cap)
    orc = ob.OracleFilter(scen["n"])
    for f in (dev, orc):
        sc.load_scenario(f, scen)
        f.update_map(scen["Z"])
        f.importance_weighting()
    compare_weights(dev, orc)
    compare_maps(sc, dev, orc, scen["n"], ordered=True)
    for f in (dev, orc):
        f.merge()
        f.prune()
    compare_maps(sc, dev, orc, scen["n"], ordered=True)


@pytest.mark.parametrize("n_lm,cap,fused", [(30, 64, False), (70, 128, True), (120, 192, False), (200, 384, True), (200, 384, False), (420, 640, True)])
@pytest.mark.parametrize("wpp", [2, 3])
def test_rank_sort_over_a_wide_range_of_weights(pkg, ob, sc, n_lm, cap, fused, wpp, monkeypatch):
    """The weighting phase ranks the mixture through a histogram over the upper words of the keys (weighting.h,
    bucket_rank_sort): weights spread over 200 decades (the bucket width adapts), clusters closer than a bucket, a few exact
    ties, and every bucket-count tier of the LDS budget (cap 64 ... 640).  Order and weights against the oracle."""
    monkeypatch.setenv("RFSGPU_STEP_WPP", str(wpp))      # (both forms of the fused step kernel: the engine would pick three waves for a launch this small)
    scen = sc.make_scenario(10, n_lm, 12, seed=n_lm + cap)
    rng = np.random.default_rng(cap)
    w = scen["w"]
    k = n_lm // 4
    w[:, :k] = 10.0 ** rng.uniform(-200, 0, (w.shape[0], k))                # 200 decades
    w[:, k:2 * k] = 0.4 + 1e-13 * rng.integers(0, 50, (w.shape[0], k))       # inside one bucket, a few exact ties
    w[:, 2 * k:2 * k + 5] = 0.25                                            # exact ties
    dev = pkg.RBPHDFilter(scen["n"], gm_capacity=cap)
    orc = ob.OracleFilter(scen["n"])
    for f in (dev, orc):
        sc.load_scenario(f, scen)
    if fused:
        dev.update_async(scen["Z"])
        dev.synchronize()
        orc.update(scen["Z"])
    else:
        for f in (dev, orc):
            f.update_map(scen["Z"])
            f.importance_weighting()
    compare_weights(dev, orc)
    compare_maps(sc, dev, orc, scen["n"], ordered=True)


def _intensity_scenario(sc, kind, seed):
    """Mixtures of more than 128 Gaussians (the size from which the weighting phase sums its intensities over listed pairs,
    weighting.h step 3b) built to take each branch of that path."""
    rng = np.random.default_rng(seed)
    if kind == "crowded":            # every Gaussian reaches every evaluation point: all lists overflow -> the dense loop per point
        scen = sc.make_scenario(12, 300, 20, seed=seed, params=dict(min_weight=0.0))
        scen["cov"] = scen["cov"] * 400.0          # sigma 0.4 ... 2 m on a 2.5 m map
    elif kind == "half_crowded":     # lists of 30 ... 100 entries: some evaluation points overflow, some do not
        scen = sc.make_scenario(12, 300, 20, seed=seed, params=dict(min_weight=0.0))
        scen["cov"] = scen["cov"] * rng.uniform(1.0, 5.0, (12, 300, 1, 1))
    elif kind == "untrusted":        # covariances whose fp32 image is refused: singular, indefinite, correlation 1 - 1e-7, sigma 1e-6, huge
        scen = sc.make_scenario(12, 200, 24, seed=seed)
        c = scen["cov"]
        c[:, 3] = [[1e-3, 1e-3], [1e-3, 1e-3]]                      # det == 0
        c[:, 17] = [[-1e-3, 0.0], [0.0, 2e-3]]                      # indefinite
        c[:, 40, 0, 1] = c[:, 40, 1, 0] = np.sqrt(c[:, 40, 0, 0] * c[:, 40, 1, 1]) * (1 - 1e-7)
        c[:, 77] = np.diag([1e-12, 1e-12])                          # sigma 1e-6 m
        c[:, 101] = np.diag([1e6, 1e6])
        c[:, 150] = np.diag([1e-9, 4e-3])
    elif kind == "faint_parents":    # landmarks of weight 1e-8 under a clutter intensity of 1e-14: their updated copies get weight ~1 and become
        scen = sc.make_scenario(12, 200, 24, seed=seed, params=dict(clutter=1e-14))   # evaluation points whose prior intensity is 2^-30 of their own term
        scen["w"][:, ::2] = 1e-8
    elif kind == "tiny_weights":     # weights far below fp32's range keep their order in the log2 estimate
        scen = sc.make_scenario(12, 200, 24, seed=seed, params=dict(min_weight=0.0), n_eval=4)
        scen["w"] *= 1e-50
    else:
        raise ValueError(kind)
    return scen


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("kind", ["crowded", "half_crowded", "untrusted", "faint_parents", "tiny_weights"])
@pytest.mark.parametrize("wpp", [2, 3])
def test_intensity_sums_over_listed_pairs(pkg, ob, sc, kind, fused, wpp, monkeypatch):
    """importanceWeighting's intensity sums (include/RBPHDFilter.hpp:776-800) on mixtures large enough for the sparse form: the
    branches that leave the ordinary path -- list overflow, Gaussians listed for every point, the prior-intensity check failing --
    give the oracle's weights like the ordinary one."""
    monkeypatch.setenv("RFSGPU_STEP_WPP", str(wpp))      # (both forms of the fused step kernel: the engine would pick three waves for a launch this small)
    scen = _intensity_scenario(sc, kind, 77)
    dev = pkg.RBPHDFilter(scen["n"], gm_capacity=640)
    orc = ob.OracleFilter(scen["n"])
    for f in (dev, orc):
        sc.load_scenario(f, scen)
    if fused:
        dev.update_async(scen["Z"])
        dev.synchronize()
        orc.update(scen["Z"])
    else:
        for f in (dev, orc):
            f.update_map(scen["Z"])
            f.importance_weighting()
    wd, wo = dev.get_weights(), orc.get_weights()
    assert np.array_equal(np.isfinite(wd), np.isfinite(wo))
    ok = np.isfinite(wo)
    np.testing.assert_allclose(wd[ok], wo[ok], rtol=1e-8, atol=0)
    assert (wo[ok] > 0).any()


def _clustered_mixtures(sc, n_particles, n_gauss, kind, seed):
    """Mixtures built to hit every path of the device merge: dense clusters (chains of merges, rows that share partners,
    rows absorbed by earlier rows), coincident means, crowded neighbourhoods (more partners than a row can list),
    a wide spread of covariance sizes (cells set by the largest radius) and degenerate covariances (infinite bound)."""
    rng = np.random.default_rng(seed)
    scen = sc.make_scenario(n_particles, n_gauss, 3, seed=seed)
    n, M = n_particles, n_gauss
    mean = np.zeros((n, M, 2))
    cov = np.zeros((n, M, 2, 2))
    for i in range(n):
        if kind == "clusters":          # ~M/6 clusters of ~6, spacing comparable to the merge radius
            k = max(1, M // 6)
            c = rng.uniform(-8, 8, (k, 2))
            a = rng.integers(0, k, M)
            mean[i] = c[a] + rng.normal(0, 0.04, (M, 2))
            s = rng.uniform(0.03, 0.09, (M, 2))
        elif kind == "crowded":         # a few very crowded spots: far more than 8 partners per row
            k = 3
            c = rng.uniform(-3, 3, (k, 2))
            a = rng.integers(0, k, M)
            mean[i] = c[a] + rng.normal(0, 0.02, (M, 2))
            s = rng.uniform(0.05, 0.1, (M, 2))
        elif kind == "coincident":      # exact duplicates and near-duplicates
            k = max(1, M // 4)
            c = rng.uniform(-5, 5, (k, 2))
            a = rng.integers(0, k, M)
            mean[i] = c[a]
            mean[i, ::3] += rng.normal(0, 1e-3, (len(mean[i, ::3]), 2))
            s = rng.uniform(0.02, 0.05, (M, 2))
        elif kind == "mixed_scales":    # one huge covariance sets the cell size; the rest are tight clusters
            k = max(1, M // 5)
            c = rng.uniform(-20, 20, (k, 2))
            a = rng.integers(0, k, M)
            mean[i] = c[a] + rng.normal(0, 0.05, (M, 2))
            s = rng.uniform(0.03, 0.08, (M, 2))
            s[:: max(1, M // 3)] = rng.uniform(2.0, 4.0, (len(s[:: max(1, M // 3)]), 2))
        elif kind == "chain":           # a line of equally spaced Gaussians: every merge moves the row toward the next one
            mean[i, :, 0] = 0.02 * np.arange(M) + rng.normal(0, 1e-3, M)
            mean[i, :, 1] = rng.normal(0, 1e-3, M)
            s = np.full((M, 2), 0.05)
        else:
            raise ValueError(kind)
        rho = rng.uniform(-0.5, 0.5, M)
        cov[i, :, 0, 0] = s[:, 0] ** 2
        cov[i, :, 1, 1] = s[:, 1] ** 2
        cov[i, :, 0, 1] = cov[i, :, 1, 0] = rho * s[:, 0] * s[:, 1]
    scen["mean"], scen["cov"] = mean, cov
    scen["w"] = rng.uniform(0.005, 1.0, (n, M))
    return scen


@pytest.mark.parametrize("kind,M,cap", [("clusters", 150, 256), ("clusters", 500, 512), ("crowded", 120, 128), ("coincident", 200, 256),
                                        ("mixed_scales", 180, 192), ("chain", 100, 128), ("chain", 300, 320)])
def test_merge_stress_against_oracle(pkg, ob, sc, kind, M, cap):
    """GaussianMixture::merge / prune on adversarial mixtures, two-kernel form (merge, then prune)."""
    scen = _clustered_mixtures(sc, 12, M, kind, seed=700 + M)
    dev, orc = make_pair(pkg, ob, sc, scen, cap=cap)
    for f in (dev, orc):
        f.merge()
    compare_maps(sc, dev, orc, scen["n"], ordered=True)
    assert dev.gm_sizes().max() < M          # the mixtures really merge
    for f in (dev, orc):
        f.prune()
    compare_maps(sc, dev, orc, scen["n"], ordered=True)
    for f in (dev, orc):                     # merging an already merged + pruned mixture again (holes gone, new order)
        f.merge()
    compare_maps(sc, dev, orc, scen["n"], ordered=True)


def test_merge_with_degenerate_covariances(pkg, ob, sc):
    """An indefinite covariance gives an unbounded prefilter radius (its Mahalanobis "distance" can be negative at any
    range): the grid collapses to one cell and every pair takes the exact test, as in the reference.  (An exactly
    singular matrix is outside the parity envelope: its determinant is 0 or 1e-21 depending on FMA contraction.)"""
    scen = _clustered_mixtures(sc, 6, 90, "clusters", seed=77)
    scen["cov"][:, 5] = [[1e-2, 2e-2], [2e-2, 1e-2]]           # indefinite
    scen["cov"][:, 40] = [[1e-2, 3e-2], [3e-2, 1e-2]]          # indefinite
    dev, orc = make_pair(pkg, ob, sc, scen, cap=128)
    for f in (dev, orc):
        f.merge()
        f.prune()
    assert np.array_equal(dev.gm_sizes(), orc.gm_sizes())
    for i in range(scen["n"]):
        a, b = dev.export_gm(i), orc.export_gm(i)
        fin = np.isfinite(b[0]) & np.isfinite(b[2]).all(1) & np.isfinite(b[3]).all((1, 2))
        sc.assert_gm_close(tuple(x[fin] for x in a), tuple(x[fin] for x in b), GM_RTOL, GM_ATOL, ordered=True)


@pytest.mark.parametrize("kind,M", [("clusters", 150), ("crowded", 100), ("chain", 120)])
@pytest.mark.parametrize("wpp", [2, 3])
def test_fused_merge_prune_stress_through_update(pkg, ob, sc, kind, M, wpp, monkeypatch):
    """Same adversarial maps through rfsgpu_update (fused merge+prune kernel) and update_async."""
    monkeypatch.setenv("RFSGPU_STEP_WPP", str(wpp))      # (both forms of the fused step kernel: the engine would pick three waves for a launch this small)
    scen = _clustered_mixtures(sc, 10, M, kind, seed=900 + M)
    dev, orc = make_pair(pkg, ob, sc, scen, cap=512)
    for f in (dev, orc):
        f.update(scen["Z"])
    compare_weights(dev, orc)
    compare_maps(sc, dev, orc, scen["n"], ordered=True)


# ---- FastSLAM 1.0 on the same handle (include/FastSLAM.hpp) -----------------------------------------------------

def _fastslam_pair(pkg, ob, sc, scen, cap=256, **cfg_over):
    dev, orc = make_pair(pkg, ob, sc, scen, cap=cap)
    for f in (dev, orc):
        for i in range(scen["n"]):                       # landmark maps: weights are log-odds of existence
            f.import_gm(i, np.log(scen["w"][i] / (1 - scen["w"][i] * 0.5)), scen["mean"][i], scen["cov"][i])
        cfg = f.default_fastslam_config()
        for k, v in cfg_over.items():
            setattr(cfg, k, v)
        f.set_fastslam_config(cfg)
    return dev, orc


def _compare_fastslam(sc, dev, orc, n):
    compare_weights(dev, orc)
    compare_maps(sc, dev, orc, n, ordered=True)
    for i in range(n):
        md, cd, sd, kd = dev.export_birth_candidates(i)
        mo, co, so, ko = orc.export_birth_candidates(i)
        assert list(sd) == list(so) and list(kd) == list(ko), i
        np.testing.assert_allclose(md, mo, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(cd, (co + np.swapaxes(co, -1, -2)) / 2, rtol=1e-8, atol=1e-12)


FS_SCENARIOS = [
    dict(n_particles=16, n_landmarks=12, n_z=5, seed=61),
    dict(n_particles=24, n_landmarks=70, n_z=19, seed=62),                       # ragged: 70 = 64 + 6
    dict(n_particles=32, n_landmarks=200, n_z=30, seed=63, rmax=25.0),           # 200 landmarks over a 25 m disc
    dict(n_particles=12, n_landmarks=130, n_z=30, seed=64, frac_in_fov=0.25, rmax=12.0),    # most landmarks out of range
    dict(n_particles=8, n_landmarks=3, n_z=20, seed=65),                          # more measurements than landmarks
    dict(n_particles=6, n_landmarks=40, n_z=64, seed=66),                         # max measurements
]


@pytest.mark.parametrize("kw", FS_SCENARIOS)
def test_fastslam_update_matches_oracle(pkg, ob, sc, kw):
    """FastSLAM::updateMap (in-range table, CostMatrix::reduce + Hungarian association, KF correction, existence log-odds,
    pruning, new landmarks) on the device vs the oracle, three predict/update/normalise cycles."""
    scen = sc.make_scenario(**kw)
    dev, orc = _fastslam_pair(pkg, ob, sc, scen)
    rng = np.random.default_rng(kw["seed"])
    for step in range(3):
        Z = scen["Z"] + rng.normal(0, 2e-3, scen["Z"].shape)
        for f in (dev, orc):
            f.predict_map(False)             # FastSLAM::predict: staticStep on every landmark (:376-383)
            f.fastslam_update(Z)
        _compare_fastslam(sc, dev, orc, scen["n"])
        for f in (dev, orc):
            s = f.weight_sums()
            f.normalize_weights(s[0])


def test_fastslam_candidate_lists_and_ambiguous_associations(pkg, ob, sc):
    """Landmark candidates that need several supporting measurements, and a loose likelihood floor so that rows and columns
    of the table compete (the in-kernel Hungarian path)."""
    scen = sc.make_scenario(20, 60, 24, seed=71)
    dev, orc = _fastslam_pair(pkg, ob, sc, scen, landmarkCandidateMeasurementCountThreshold=3, landmarkCandidateMeasurementCheckThreshold=4,
                              landmarkCandidateCurrentMeasurementCountThreshold=0, landmarkCandidateMeasurementSupportDist=3.0,
                              minLogMeasurementLikelihood=-60.0, pruningMeasurementsThreshold=5)
    rng = np.random.default_rng(2)
    seen = 0
    for step in range(5):
        Z = scen["Z"] + rng.normal(0, 5e-3, scen["Z"].shape)
        for f in (dev, orc):
            f.predict_map(False)
            f.fastslam_update(Z)
        _compare_fastslam(sc, dev, orc, scen["n"])
        seen += sum(len(orc.export_birth_candidates(i)[2]) for i in range(scen["n"]))
        for f in (dev, orc):
            s = f.weight_sums()
            f.normalize_weights(s[0])
    assert seen > 0, "no landmark candidate was ever queued"
    assert orc.fs_solver_max_dim() >= 2, "no particle had competing associations (the Hungarian path was not exercised)"


def test_fastslam_host_mirror_resamples_with_candidates(pkg, ob, sc):
    """rfs_slam_amd.FastSLAM.update_and_resample (FastSLAM::update + resampleWithMapCopy) against the oracle driven through the
    same host logic and the same uniform draws: weights, maps and the candidate lists that travel with resampled particles."""
    scen = sc.make_scenario(24, 40, 14, seed=81, rmax=8.0)
    dev = pkg.FastSLAM(scen["n"], gm_capacity=256)
    orc = ob.OracleFilter(scen["n"])
    for f in (dev, orc):
        sc.load_scenario(f, scen)
        for i in range(scen["n"]):
            f.import_gm(i, np.zeros(scen["w"][i].shape), scen["mean"][i], scen["cov"][i])
        cfg = f.default_fastslam_config()
        cfg.landmarkCandidateMeasurementCountThreshold = 2
        cfg.landmarkCandidateCurrentMeasurementCountThreshold = 0
        cfg.landmarkCandidateMeasurementCheckThreshold = 3
        cfg.minUpdatesBeforeResample = 1
        if f is dev:
            dev.fs_config = cfg
        f.set_fastslam_config(cfg)
    dev.setEffectiveParticleCountThreshold(scen["n"])       # always below: resample at every opportunity
    rng_d, rng_o, rz = np.random.default_rng(5), np.random.default_rng(5), np.random.default_rng(6)
    resamples = 0
    for step in range(4):
        Z = scen["Z"] + rz.normal(0, 3e-3, scen["Z"].shape)
        dev.predict_map()
        orc.predict_map(False)
        did = dev.update_and_resample(Z, u01_fn=rng_d.random)
        # the oracle through the same host steps
        orc.fastslam_update(Z)
        s = orc.weight_sums()
        orc.normalize_weights(s[0])
        w = orc.get_weights()
        if did:
            src = pkg.engine.systematic_resample_plan(w, float(rng_o.random()))
            orc.resample_apply(src)
            resamples += 1
        _compare_fastslam(sc, dev, orc, scen["n"])
    assert resamples > 0


@pytest.mark.parametrize("kw", [dict(n_particles=10, n_landmarks=30, n_z=9, seed=91, scan="const"),
                                dict(n_particles=16, n_landmarks=70, n_z=18, seed=92, scan="ragged")])
def test_fastslam_victoria_park_model(pkg, ob, sc, kw):
    """FastSLAM::updateMap with the Victoria Park model (3-D landmarks, scan-based Pd, expected-clutter-number false-alarm
    probability, candidate lists with several supporting measurements), device vs oracle over three cycles."""
    scen = sc.make_vp_scenario(**kw)
    dev, orc = make_vp_pair(pkg, ob, sc, scen)
    for f in (dev, orc):
        for i in range(scen["n"]):
            f.import_gm(i, np.zeros(scen["w"][i].shape), scen["mean"][i], scen["cov"][i])
        cfg = f.default_fastslam_config()
        cfg.landmarkCandidateMeasurementCountThreshold = 2
        cfg.landmarkCandidateCurrentMeasurementCountThreshold = 0
        cfg.landmarkCandidateMeasurementCheckThreshold = 3
        cfg.landmarkCandidateMeasurementSupportDist = 3.0
        cfg.mapExistencePruneThreshold = -5.0
        f.set_fastslam_config(cfg)
    rng = np.random.default_rng(kw["seed"])
    for step in range(3):
        Z = scen["Z"] + rng.normal(0, 1e-3, scen["Z"].shape)
        for f in (dev, orc):
            f.predict_map(False)
            f.fastslam_update(Z)
        _compare_fastslam(sc, dev, orc, scen["n"])
        assert sum(dev.landmarks_in_fov(i) for i in range(scen["n"])) > 0      # landmarks were associated and corrected
        for f in (dev, orc):
            s = f.weight_sums()
            f.normalize_weights(s[0])


@pytest.mark.parametrize("kw,hyp,diff", [(dict(n_particles=6, n_landmarks=12, n_z=6, seed=5, rmax=5.0), 3, 50.0),
                                         (dict(n_particles=10, n_landmarks=25, n_z=10, seed=7, rmax=6.0), 4, 3.0),
                                         (dict(n_particles=8, n_landmarks=4, n_z=9, seed=9, rmax=5.0), 2, 5.0)])
def test_multi_hypothesis_fastslam(pkg, ob, sc, kw, hyp, diff):
    """MH-FastSLAM (maxNDataAssocHypotheses > 1): Murty's k best associations on the dense reduced table, particle copies in
    particle order (pi[h] = nParticles_ - h), per-hypothesis updates, then resample(nParticles_init) with map and candidate
    copies -- device vs oracle: particle counts, parents, weights, maps, candidate lists, over several cycles."""
    scen = sc.make_scenario(**kw)
    n0 = scen["n"]
    dev = pkg.RBPHDFilter(n0, gm_capacity=128, max_particles=n0 * hyp * 4)
    orc = ob.OracleFilter(n0)
    for f in (dev, orc):
        sc.load_scenario(f, scen)
        for i in range(n0):
            f.import_gm(i, np.zeros(scen["w"][i].shape), scen["mean"][i], scen["cov"][i])
        cfg = f.default_fastslam_config()
        cfg.maxNDataAssocHypotheses = hyp
        cfg.maxDataAssocLogLikelihoodDiff = diff
        cfg.landmarkCandidateMeasurementCountThreshold = 2
        cfg.landmarkCandidateCurrentMeasurementCountThreshold = 0
        cfg.landmarkCandidateMeasurementCheckThreshold = 3
        f.set_fastslam_config(cfg)
    rng, rz = np.random.default_rng(11), np.random.default_rng(12)
    grew = 0
    poses = scen["poses"].copy()
    for step in range(4):
        Z = scen["Z"] + rz.normal(0, 3e-3, scen["Z"].shape)
        for f in (dev, orc):
            f.predict_map(False)
            f.fastslam_update(Z)
        assert dev.n == orc.n
        par = dev.particle_parents()
        assert np.array_equal(par, orc.particle_parents())
        grew += int(dev.n > len(poses))
        poses = poses[par]                                     # the host duplicates its poses
        _compare_fastslam(sc, dev, orc, dev.n)
        for f in (dev, orc):
            s = f.weight_sums()
            f.normalize_weights(s[0])
        resampled = dev.n > n0 or step == 2                    # resampleWithMapCopy: back to the initial size
        if resampled:
            w = orc.get_weights()
            plan = pkg.engine.systematic_resample_plan(w, float(rng.random()), n_out=n0)
            for f in (dev, orc):
                f.resample_apply(plan, n_out=n0)
            poses = poses[plan]
            assert dev.n == n0 and orc.n == n0
            _compare_fastslam(sc, dev, orc, n0)
        for f in (dev, orc):
            f.fastslam_set_resample_occured(resampled)
            f.set_poses(poses, scen["pose_cov"])
    assert grew > 0, "no particle was ever multiplied"


def test_mh_fastslam_host_mirror(pkg, ob, sc):
    """rfs_slam_amd.FastSLAM with max_hypotheses > 1 (FastSLAM::update + resampleWithMapCopy: forced resample(nParticles_init)
    once the count passes nParticlesMax, conditional otherwise, resampleOccured_ fed back into the next update's candidate
    inheritance) against the oracle taken through the same host steps with the same uniform draws."""
    scen = sc.make_scenario(n_particles=8, n_landmarks=14, n_z=7, seed=15, rmax=5.0)
    n0 = scen["n"]
    dev = pkg.FastSLAM(n0, gm_capacity=128, max_hypotheses=3, n_particles_max=12)
    orc = ob.OracleFilter(n0)
    assert dev.max_particles >= 36
    for f in (dev, orc):
        sc.load_scenario(f, scen)
        for i in range(n0):
            f.import_gm(i, np.zeros(scen["w"][i].shape), scen["mean"][i], scen["cov"][i])
    cfg = dev.fs_config
    cfg.maxDataAssocLogLikelihoodDiff = 30.0
    cfg.landmarkCandidateMeasurementCountThreshold = 2
    cfg.landmarkCandidateCurrentMeasurementCountThreshold = 0
    cfg.landmarkCandidateMeasurementCheckThreshold = 3
    cfg.minUpdatesBeforeResample = 2
    orc.set_fastslam_config(cfg)
    dev.setEffectiveParticleCountThreshold(n0)                 # resample whenever the counters allow it
    rng_d, rng_o, rz = np.random.default_rng(21), np.random.default_rng(21), np.random.default_rng(22)
    poses = scen["poses"].copy()
    forced = shrunk = 0
    upd_o = 0
    for step in range(5):
        Z = scen["Z"] + rz.normal(0, 3e-3, scen["Z"].shape)
        dev.predict_map()
        orc.predict_map(False)
        did = dev.update_and_resample(Z, u01_fn=rng_d.random)
        # the oracle through FastSLAM::update's host steps
        upd_o += 1
        orc.fastslam_update(Z)
        assert np.array_equal(dev.parents, orc.particle_parents())
        poses = poses[dev.parents]
        n_grown = orc.n
        s = orc.weight_sums()
        orc.normalize_weights(s[0])
        want = n_grown > cfg.nParticlesMax or upd_o >= cfg.minUpdatesBeforeResample
        assert did == want
        if did:
            forced += int(n_grown > cfg.nParticlesMax)
            shrunk += int(n_grown > n0)
            plan = pkg.engine.systematic_resample_plan(orc.get_weights(), float(rng_o.random()), n_out=n0)
            assert np.array_equal(plan, dev.last_resample_plan)
            orc.resample_apply(plan, n_out=n0)
            poses = poses[plan]
            upd_o = 0
        orc.fastslam_set_resample_occured(did)
        assert dev.n == orc.n and (not did or dev.n == n0)
        _compare_fastslam(sc, dev, orc, dev.n)
        for f in (dev, orc):
            f.set_poses(poses, scen["pose_cov"])
    assert forced > 0 and shrunk > 0


def test_fastslam_hypothesis_count_limit(pkg, sc):
    scen = sc.make_scenario(4, 5, 3, seed=1)
    dev = pkg.RBPHDFilter(4, gm_capacity=64)
    sc.load_scenario(dev, scen)
    cfg = dev.default_fastslam_config()
    cfg.maxNDataAssocHypotheses = 17
    dev.set_fastslam_config(cfg)
    with pytest.raises(pkg.capi.EngineError) as e:
        dev.fastslam_update(scen["Z"])
    assert e.value.status == pkg.capi.ERR_UNSUPPORTED
    cfg.maxNDataAssocHypotheses = 3                 # no room for the copies: refused, not truncated
    dev.set_fastslam_config(cfg)
    try:
        dev.fastslam_update(scen["Z"])
    except pkg.capi.EngineError as e2:
        assert e2.status == pkg.capi.ERR_CAPACITY


# ---- Victoria Park model (3-D landmarks, scan-based Pd, birth-candidate lists) -----------------------------------------

def make_vp_pair(pkg, ob, sc, scen, cap=192):
    dev = pkg.RBPHDFilter(scen["n"], device_id=0, gm_capacity=cap, model=pkg.capi.MODEL_VICTORIAPARK_3D)
    orc = ob.OracleFilter(scen["n"], model=pkg.capi.MODEL_VICTORIAPARK_3D)
    for f in (dev, orc):
        sc.load_scenario(f, scen)
    return dev, orc


VP_SCENARIOS = [
    dict(n_particles=12, n_landmarks=30, n_z=9, seed=41, scan="const"),
    dict(n_particles=20, n_landmarks=70, n_z=18, seed=42, scan="ragged"),      # ragged: 70 = 64 + 6; occlusion counts vary
    dict(n_particles=8, n_landmarks=5, n_z=3, seed=43, scan="const"),
]


@pytest.mark.parametrize("kw", VP_SCENARIOS)
def test_vp_phases_stepwise(pkg, ob, sc, kw):
    scen = sc.make_vp_scenario(**kw)
    dev, orc = make_vp_pair(pkg, ob, sc, scen)
    for f in (dev, orc):
        f.update_map(scen["Z"])
    compare_maps(sc, dev, orc, scen["n"], ordered=True)
    for i in range(scen["n"]):
        assert np.array_equal(dev.get_unused(i), orc.get_unused(i))
        assert dev.landmarks_in_fov(i) == orc.landmarks_in_fov(i)
        np.testing.assert_allclose(dev.export_gm(i)[1], orc.export_gm(i)[1], rtol=0, atol=0)
    for f in (dev, orc):
        f.importance_weighting()
    compare_weights(dev, orc)
    compare_maps(sc, dev, orc, scen["n"], ordered=True)
    for f in (dev, orc):
        f.merge()
    compare_maps(sc, dev, orc, scen["n"], ordered=True)
    for f in (dev, orc):
        f.prune()
    compare_maps(sc, dev, orc, scen["n"], ordered=True)


@pytest.mark.parametrize("kind,M,cap", [("clusters", 60, 64), ("crowded", 48, 64), ("chain", 64, 64), ("coincident", 50, 128), ("clusters", 150, 192),
                                        ("chain", 100, 128)])
def test_vp_merge_stress_against_oracle(pkg, ob, sc, kind, M, cap):
    """GaussianMixture::merge / prune for 3-D Gaussians on adversarial mixtures.  M <= 64: the lane-parallel scan of vp.h (every row
    on its own lane, ordered validation) INCLUDING its collisions -- rows that go after the same entry (crowded spots, chains in
    which a merge moves a row onto the next row's partner) are redone by the row-by-row scan; M > 64: the row-by-row scan alone.
    Both the two-kernel form (merge, then prune) and the fused merge + prune inside an update."""
    flat = _clustered_mixtures(sc, 10, M, kind, seed=1300 + M)
    scen = sc.make_vp_scenario(10, M, 4, seed=1300 + M)
    rng = np.random.default_rng(1400 + M)
    scale = 15.0                                   # the 2-D test maps are metres wide; Victoria Park merges at ~1 m
    scen["mean"] = np.concatenate([flat["mean"] * scale + np.array([0.0, 30.0]), rng.uniform(0.3, 0.5, (10, M, 1)) +
                                   (rng.normal(0, 0.01, (10, M, 1)) if kind != "coincident" else 0.0)], axis=2)
    cov = np.zeros((10, M, 3, 3))
    cov[:, :, :2, :2] = flat["cov"] * scale * scale
    cov[:, :, 2, 2] = rng.uniform(0.01, 0.04, (10, M)) ** 2
    cov[:, :, 0, 2] = cov[:, :, 2, 0] = 0.1 * np.sqrt(cov[:, :, 0, 0] * cov[:, :, 2, 2]) * rng.uniform(-1, 1, (10, M))
    scen["cov"] = cov
    scen["w"] = flat["w"]
    dev, orc = make_vp_pair(pkg, ob, sc, scen, cap=cap)
    for f in (dev, orc):
        f.merge()
    compare_maps(sc, dev, orc, scen["n"], ordered=True)
    assert dev.gm_sizes().max() < M
    for f in (dev, orc):
        f.prune()
    compare_maps(sc, dev, orc, scen["n"], ordered=True)
    for f in (dev, orc):
        f.merge()
        f.prune()
    compare_maps(sc, dev, orc, scen["n"], ordered=True)
    if M + 4 <= cap:                               # the same maps through the fused step (permutation, Pd cache, fused prune)
        dev2, orc2 = make_vp_pair(pkg, ob, sc, scen, cap=cap)
        for f in (dev2, orc2):
            f.update(scen["Z"][:1] * 0 + np.array([[75.0, 0.1, 0.4]]))    # one measurement nobody gates: the maps only merge
        compare_weights(dev2, orc2)
        compare_maps(sc, dev2, orc2, scen["n"], ordered=True)


def test_vp_fused_update_and_birth_candidates_multi_step(pkg, ob, sc):
    """Victoria Park cycle: predict (birth candidates with the 5/10/2 thresholds of the shipped cfg, scaled down so that
    promotions happen within the test) -> update, several steps; maps, weights and the candidate lists must agree."""
    scen = sc.make_vp_scenario(10, 12, 8, seed=44, params=dict(birth_count_thr=3, birth_check_thr=2))
    dev, orc = make_vp_pair(pkg, ob, sc, scen)
    rng = np.random.default_rng(3)
    promoted = 0
    for step in range(8):
        Z = scen["Z"] + rng.normal(0, 1, scen["Z"].shape) * np.array([0.05, 0.002, 0.01])
        for f in (dev, orc):
            f.predict_map(True)
            f.update(Z)
            s = f.weight_sums()
            f.normalize_weights(s[0])
        np.testing.assert_allclose(dev.get_weights(), orc.get_weights(), rtol=1e-8)
        compare_maps(sc, dev, orc, scen["n"])
        for i in range(scen["n"]):
            md, cd, sd, kd = dev.export_birth_candidates(i)
            mo, co, so, ko = orc.export_birth_candidates(i)
            assert list(sd) == list(so) and list(kd) == list(ko), (step, i)
            np.testing.assert_allclose(md, mo, rtol=1e-9, atol=1e-10)
            np.testing.assert_allclose(cd, (co + np.swapaxes(co, -1, -2)) / 2, rtol=1e-8, atol=1e-12)
            promoted += len(so)
    assert promoted > 0, "scenario never produced a birth candidate"


@pytest.mark.parametrize("n_particles,n_landmarks,n_z,stride", [(150, 45, 12, 1), (4700, 24, 8, 97), (9000, 10, 4, 499), (20000, 6, 3, 1999)])
def test_vp_step_results_do_not_depend_on_the_launch_order(pkg, sc, n_particles, n_landmarks, n_z, stride):
    """Round 6: the Victoria Park step launches its particles in the order the previous step's post kernel sorted them into (longest
    first, one class of durations per workgroup, csrc/murty.h step_cost_order_class; 190 -> 166 us at configs[3]).  (4700 particles:
    one fetch round of the 256-thread post kernel with every class populated; 9000: two rounds; 20000: more than the sort's 64
    particles per thread, the order stays what it was.)  Which wave slot works on which particle must change no
    result: the same predict / update steps with slot == particle (mode 0), with the automatic order (mode 2, the default) and with a
    reversed and a random frozen order (mode 1) give the same bits -- weights, maps, unused lists, FOV counts; the measured durations
    come back per particle; a list that is not a permutation is refused."""
    scen = sc.make_vp_scenario(n_particles=n_particles, n_landmarks=n_landmarks, n_z=n_z, seed=94)
    n = scen["n"]
    rng = np.random.default_rng(3)
    orders = [("off", 0, None), ("auto", 2, None), ("reversed", 1, np.arange(n - 1, -1, -1)), ("random", 1, rng.permutation(n))]
    fs = []
    for name, mode, order in orders:
        f = pkg.RBPHDFilter(n, gm_capacity=256, model=pkg.capi.MODEL_VICTORIAPARK_3D)
        sc.load_scenario(f, scen)
        f.step_launch_order(order=order, mode=mode, want_costs=False)
        fs.append(f)
    zr = np.random.default_rng(8)
    for step in range(4):
        Z = scen["Z"] + zr.normal(0, 1, scen["Z"].shape) * np.array([0.05, 0.002, 0.01])
        for f in fs:
            f.predict_map(True)
            f.update(Z)
        a = fs[0]
        for (name, _, _), b in zip(orders[1:], fs[1:]):
            assert np.array_equal(a.get_weights(), b.get_weights()), (name, step)
            assert np.array_equal(a.gm_sizes(), b.gm_sizes()), (name, step)
            for i in range(0, n, stride):
                for x, y in zip(a.export_gm(i), b.export_gm(i)):
                    assert np.array_equal(x, y), (name, step, i)
                assert list(a.get_unused(i)) == list(b.get_unused(i)) and a.landmarks_in_fov(i) == b.landmarks_in_fov(i)
        for f in fs:
            s_ = f.weight_sums(); f.normalize_weights(s_[0])
    cost = fs[1].step_launch_order(mode=2)
    assert cost.shape == (n,) and np.all(cost > 0) and np.all(np.isfinite(cost))
    with pytest.raises(RuntimeError):
        fs[1].step_launch_order(order=np.zeros(n, dtype=np.int32))
    for f in fs:
        f.close()


def test_2d_step_results_do_not_depend_on_the_launch_order(pkg, sc):
    """The 2-D fused step takes the cost-ordered launch where its workgroups are not all resident at once (the instantiations without
    phase priorities, csrc/step_fused.h StepLaunchOrder): 3000 particles at capacity 384 are such a launch (2048 two-wave workgroups
    resident).  Same bits with slot == particle, the automatic order and a random frozen one, over predict / update steps; the
    all-resident launch (300 particles) ignores the order and reports no durations."""
    scen = sc.make_scenario(3000, 60, 10, seed=314)
    n = 3000
    rng = np.random.default_rng(5)
    orders = [("off", 0, None), ("auto", 2, None), ("random", 1, rng.permutation(n))]
    fs = []
    for name, mode, order in orders:
        f = pkg.RBPHDFilter(n, gm_capacity=384)
        sc.load_scenario(f, scen)
        f.step_launch_order(order=order, mode=mode, want_costs=False)
        fs.append(f)
    zr = np.random.default_rng(9)
    for step in range(4):
        Z = scen["Z"] + zr.normal(0, 1, scen["Z"].shape) * np.array([0.05, 0.002])
        for f in fs:
            f.predict_map(True)
            f.update(Z)
        assert fs[0].last_step_variant()[1] == 0, "this launch was meant to run without phase priorities"
        a = fs[0]
        for (name, _, _), b in zip(orders[1:], fs[1:]):
            assert np.array_equal(a.get_weights(), b.get_weights()), (name, step)
            assert np.array_equal(a.gm_sizes(), b.gm_sizes()), (name, step)
            for i in range(0, n, 61):
                for x, y in zip(a.export_gm(i), b.export_gm(i)):
                    assert np.array_equal(x, y), (name, step, i)
                assert list(a.get_unused(i)) == list(b.get_unused(i))
        for f in fs:
            s_ = f.weight_sums(); f.normalize_weights(s_[0])
    cost = fs[1].step_launch_order(mode=2)
    assert np.all(cost > 0) and np.all(np.isfinite(cost))
    for f in fs:
        f.close()
    small = sc.make_scenario(300, 60, 10, seed=315)
    f = pkg.RBPHDFilter(300, gm_capacity=384)
    sc.load_scenario(f, small)
    f.update(small["Z"])
    assert f.last_step_variant()[1] == 1 and not np.any(f.step_launch_order(mode=2))
    f.close()


@pytest.mark.parametrize("kw", [dict(n_particles=40, n_landmarks=45, n_z=12, seed=91), dict(n_particles=33, n_landmarks=130, n_z=18, seed=92),
                                dict(n_particles=12, n_landmarks=20, n_z=7, seed=93, use_cluster=1)])
def test_vp_fused_step_is_bit_identical_to_the_three_kernel_path(pkg, sc, kw):
    """rfsgpu_update with the Victoria Park model is ONE launch (vp_step_fused_kernel: the sorted order a permutation in LDS, no
    sorted copy of the 11-plane slab) unless phase timing is asked for: maps, weights, unused lists and FOV counts must have the
    same bits as the three stand-alone kernels give, over several predict / update steps, incl. the SC-PHD weighting."""
    use_cluster = kw.pop("use_cluster", 0)
    scen = sc.make_vp_scenario(**kw)
    fs = []
    for timing in (False, True):
        f = pkg.RBPHDFilter(scen["n"], gm_capacity=256, model=pkg.capi.MODEL_VICTORIAPARK_3D)
        sc.load_scenario(f, scen)
        if use_cluster:
            cfg = f.get_filter_config()
            cfg.useClusterProcess = 1
            f.set_filter_config(cfg)
        f.set_phase_timing(timing)
        fs.append(f)
    rng = np.random.default_rng(8)
    for step in range(4):
        Z = scen["Z"] + rng.normal(0, 1, scen["Z"].shape) * np.array([0.05, 0.002, 0.01])
        for f in fs:
            f.predict_map(True)
            f.update(Z)
        a, b = fs
        assert np.array_equal(a.get_weights(), b.get_weights()), step
        assert np.array_equal(a.gm_sizes(), b.gm_sizes())
        assert a.gm_sizes().max() > 5
        for i in range(scen["n"]):
            for x, y in zip(a.export_gm(i), b.export_gm(i)):
                assert np.array_equal(x, y), (step, i)
            assert list(a.get_unused(i)) == list(b.get_unused(i)) and a.landmarks_in_fov(i) == b.landmarks_in_fov(i)
        for f in fs:
            f.normalize_weights(f.weight_sums()[0])
    t = fs[0].getTimingInfo()
    assert t.mapUpdate_wall > 0 and t.particleWeighting_wall == 0      # the fused path books the whole step under mapUpdate


def test_cycle_and_update_io_fall_back_for_the_victoria_park_model(pkg, sc):
    """rfsgpu_cycle_async / rfsgpu_update_io on a handle whose step kernel has no head (the 3-D model: candidate lists, its own fused
    kernel): the predict kernels, the input copies and the step are enqueued in the reference's order -- the same bits as
    predict_map + set_poses + set_weights + update / get_weights call by call (the unmodified Victoria Park driver's update() goes
    through this path)."""
    scen = sc.make_vp_scenario(n_particles=24, n_landmarks=30, n_z=10, seed=77, scan="ragged")
    a = pkg.RBPHDFilter(scen["n"], gm_capacity=256, model=pkg.capi.MODEL_VICTORIAPARK_3D)
    b = pkg.RBPHDFilter(scen["n"], gm_capacity=256, model=pkg.capi.MODEL_VICTORIAPARK_3D)
    for f in (a, b):
        sc.load_scenario(f, scen)
    rng = np.random.default_rng(9)
    poses = scen["poses"].copy()
    for step in range(4):
        Z = scen["Z"] + rng.normal(0, 1, scen["Z"].shape) * np.array([0.05, 0.002, 0.01])
        poses = poses + rng.normal(0, 0.01, poses.shape)
        w_in = rng.uniform(0.5, 1.0, scen["n"])
        if step % 2 == 0:
            a.cycle_async(True, Z, poses=poses, weights=w_in, normalize=False)
            a.synchronize()
            w_a = a.get_weights()
        else:
            w_a = a.update_io(Z, predict=True, poses=poses, weights=w_in)
        b.predict_map(True)
        b.set_poses(poses)
        b.set_weights(w_in)
        b.update(Z)
        np.testing.assert_array_equal(w_a, b.get_weights())
        assert np.array_equal(a.gm_sizes(), b.gm_sizes()) and a.gm_sizes().max() > 5
        for i in range(scen["n"]):
            for x, y in zip(a.export_gm(i), b.export_gm(i)):
                np.testing.assert_array_equal(x, y)
            assert list(a.get_unused(i)) == list(b.get_unused(i))
    a.close(); b.close()


def test_2d_birth_candidate_list_mode(pkg, ob, sc):
    """The candidate-list branch of addBirthGaussians with the 2-D model (birthGaussianMeasurementCountThreshold > 1)."""
    scen = sc.make_scenario(12, 6, 8, seed=45)
    dev, orc = make_pair(pkg, ob, sc, scen, cap=128)
    for f in (dev, orc):
        cfg = f.get_filter_config()
        cfg.birthGaussianMeasurementCountThreshold = 3
        cfg.birthGaussianMeasurementCheckThreshold = 2
        cfg.birthGaussianMeasurementSupportDist = 2.0
        cfg.birthGaussianCurrentMeasurementCountThreshold = 0
        f.set_filter_config(cfg)
    rng = np.random.default_rng(4)
    seen = 0
    for step in range(7):
        Z = scen["Z"] + rng.normal(0, 1e-3, scen["Z"].shape)
        for f in (dev, orc):
            f.predict_map(True)
            f.update(Z)
            s = f.weight_sums()
            f.normalize_weights(s[0])
        compare_maps(sc, dev, orc, scen["n"])
        for i in range(scen["n"]):
            md, cd, sd, kd = dev.export_birth_candidates(i)
            mo, co, so, ko = orc.export_birth_candidates(i)
            assert list(sd) == list(so) and list(kd) == list(ko), (step, i)
            np.testing.assert_allclose(md, mo, rtol=1e-9, atol=1e-11)
            seen += len(so)
    assert seen > 0


def test_2d_birth_candidate_lists_longer_than_a_wavefront(pkg, ob, sc):
    """Candidate lists of 100-200 entries (clutter that is never seen twice, CheckThreshold 8): the list is staged in LDS and
    walked 64 candidates at a time (birth.h); support matches in the second / third chunk, promotions, expiry with the erase
    moving a tail longer than a wavefront, the ++end() wrap.  Lists, supports, checks and maps against the oracle."""
    scen = sc.make_scenario(6, 5, 6, seed=46)
    dev, orc = make_pair(pkg, ob, sc, scen, cap=256)
    for f in (dev, orc):
        cfg = f.get_filter_config()
        cfg.birthGaussianMeasurementCountThreshold = 3
        cfg.birthGaussianMeasurementCheckThreshold = 8
        cfg.birthGaussianMeasurementSupportDist = 1.0
        cfg.birthGaussianCurrentMeasurementCountThreshold = 0
        f.set_filter_config(cfg)
    rng = np.random.default_rng(9)
    longest, shrank, promoted = 0, False, 0
    prev = None
    persistent = np.column_stack([rng.uniform(1.0, 4.5, 4), rng.uniform(-3, 3, 4)])     # seen in every scan: promoted after three
    for step in range(14):
        clutter = np.column_stack([rng.uniform(0.5, 4.8, 18), rng.uniform(-3.1, 3.1, 18)])
        Z = np.vstack([scen["Z"][:3] + rng.normal(0, 1e-3, (3, 2)), persistent + rng.normal(0, 1e-3, persistent.shape), clutter if step < 11 else clutter[:2]])
        for f in (dev, orc):
            f.predict_map(True)
            f.update(Z)
            s = f.weight_sums()
            f.normalize_weights(s[0])
        compare_maps(sc, dev, orc, scen["n"])
        lens = []
        for i in range(scen["n"]):
            md, cd, sd, kd = dev.export_birth_candidates(i)
            mo, co, so, ko = orc.export_birth_candidates(i)
            assert list(sd) == list(so) and list(kd) == list(ko), (step, i)
            np.testing.assert_allclose(md, mo, rtol=1e-9, atol=1e-11)
            np.testing.assert_allclose(cd, co, rtol=1e-8, atol=1e-12)
            lens.append(len(so))
        longest = max(longest, max(lens))
        if prev is not None and max(lens) < prev:
            shrank = True
        prev = max(lens)
    assert longest > 130 and shrank, (longest, shrank)


@pytest.mark.parametrize("n", [32, 256])
def test_victoria_park_dataset_extract_device_vs_oracle(pkg, ob, sc, n):
    """Config 4 in miniature: the first sensor messages of the Victoria Park dataset (tests/golden/victoria_park_extract.npz,
    extracted from the reference's data files) through the event-driven host loop -- predict with birth candidates, artificial
    clutter, scan-based Pd, update, resampling (with the reference's birth-state inheritance: candidate lists, chains of
    lower-slot parents) -- on the device and on the oracle with the same host RNG stream."""
    import os
    data = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "victoria_park_extract.npz"))
    P = dict(sc.VP_PARAMS)
    runs = []
    for make in (lambda: pkg.RBPHDFilter(n, gm_capacity=192, model=pkg.capi.MODEL_VICTORIAPARK_3D),
                 lambda: ob.OracleFilter(n, model=pkg.capi.MODEL_VICTORIAPARK_3D)):
        f = make()
        sc.apply_vp_params(f, P, np.full(361, 70.0))
        runs.append(pkg.vp_driver.VictoriaParkRun(f, data, P, seed=5).run(n_messages=900))
    dev, orc = runs
    assert dev.n_lidar == orc.n_lidar and dev.n_lidar > 50
    assert dev.n_resamples == orc.n_resamples
    np.testing.assert_allclose(dev.f.get_weights(), orc.f.get_weights(), rtol=1e-6)
    sizes = dev.f.gm_sizes()
    assert np.array_equal(sizes, orc.f.gm_sizes()) and sizes.max() > 3
    for i in range(n):
        sc.assert_gm_close(dev.f.export_gm(i), orc.f.export_gm(i), 1e-7, 1e-9)


@pytest.mark.parametrize("n,steps,seed", [(100, 320, 1), (200, 300, 2)])
def test_c1_trajectory_device_vs_oracle(pkg, ob, sc, n, steps, seed):
    """Config C1 as a TRAJECTORY: the 2-D simulator's filter loop (src/rbphdslam2dSim.cpp:540-643: predict with
    odometry and process noise, ground-truth poses for k <= 100, the measurements of the step, update, N_eff resampling with the
    reference's birth-state inheritance) on the device and on the oracle through ONE realisation (one host RNG stream, shared
    poses), compared after EVERY update: mixture sizes, unused lists, resampling decisions and plans exact; normalised weights
    1e-8; maps (w, mu, Sigma) 1e-7."""
    sd = pkg.sim2d_driver
    data = sd.generate(traj_seed=seed, kmax=steps)
    dev = pkg.RBPHDFilter(n, gm_capacity=256)
    orc = ob.OracleFilter(n)
    seen = dict(updates=0, max_size=0, births=0)

    def check(k, run, fired):
        if len(run.z_of_step) == 0:
            return
        sz = dev.gm_sizes()
        assert np.array_equal(sz, orc.gm_sizes()), k
        wd, wo = dev.get_weights(), orc.get_weights()
        np.testing.assert_allclose(wd, wo, rtol=1e-8, atol=1e-300, err_msg=f"step {k}")
        for i in range(0, n, 7 if k % 10 else 1):            # every particle every 10th step, a stride of 7 in between
            sc.assert_gm_close(dev.export_gm(i), orc.export_gm(i), 1e-7, 1e-9)
            assert list(dev.get_unused(i)) == list(orc.get_unused(i)), (k, i)
        seen["updates"] += 1
        seen["max_size"] = max(seen["max_size"], int(sz.max()))

    run = sd.Sim2dRun([dev, orc], data, seed=seed + 10).run(on_step=check)
    assert seen["updates"] > steps * 0.6 and seen["max_size"] >= 5
    assert run.n_resamples >= 3                                   # several resamplings, each followed by a predict
    assert np.array_equal(dev.get_particle_ids()[0], orc.get_particle_ids()[0])
    assert np.any(dev.get_particle_ids()[0] != np.arange(n))
    md, mo = sd.map_error(dev, 0, data["landmarks"]), sd.map_error(orc, 0, data["landmarks"])
    assert md[0] == mo[0] and md[2] == mo[2] and md[0] >= 3


def test_c1_full_run_device_and_oracle_agree_on_map_quality(pkg, ob, sc):
    """The whole shipped C1 run (3000 steps, 50 landmarks) at 100 particles on device and oracle through one realisation, for two
    realisations: final best-particle maps agree Gaussian by Gaussian, so a realisation on which the filter maps fewer landmarks
    ("36 of 50") does so on the CPU restatement of the reference's algorithm exactly as on the device -- it is
    a property of that realisation (particle depletion early on the trajectory), not of the engine."""
    sd = pkg.sim2d_driver
    results = []
    for seed in (1, 4):
        data = sd.generate(traj_seed=seed)
        dev = pkg.RBPHDFilter(100, gm_capacity=256)
        orc = ob.OracleFilter(100)
        run = sd.Sim2dRun([dev, orc], data, seed=seed).run()
        wd, wo = dev.get_weights(), orc.get_weights()
        np.testing.assert_allclose(wd, wo, rtol=1e-6)
        best = int(np.argmax(wd))
        assert np.array_equal(dev.gm_sizes(), orc.gm_sizes())
        sc.assert_gm_close(dev.export_gm(best), orc.export_gm(best), 1e-6, 1e-8)
        md, mo = sd.map_error(dev, best, data["landmarks"]), sd.map_error(orc, best, data["landmarks"])
        assert md[0] == mo[0] and md[2] == mo[2]
        results.append((seed, md, run.n_resamples))
    assert max(r[1][0] for r in results) >= 40, results           # at least one of the two realisations maps the scene well


def test_victoria_park_dataset_extract_at_full_size_fused_equals_three_kernels(pkg, sc):
    """configs[3] at its 5000 particles on the REAL dataset extract (900 messages: ~90 scans with artificial clutter, resamplings,
    birth candidates): the run with one fused launch per update and the run with the three stand-alone kernels (phase timing on)
    through the same host RNG stream end with the same bits -- weights, every mixture size, and the maps of a sample of
    particles -- and with a sane filter (finite weights, several resamplings, trees mapped)."""
    import os
    data = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "victoria_park_extract.npz"))
    n = 5000
    P = dict(sc.VP_PARAMS)
    runs = []
    for timing in (False, True):
        f = pkg.RBPHDFilter(n, gm_capacity=192, model=pkg.capi.MODEL_VICTORIAPARK_3D)
        sc.apply_vp_params(f, P, np.full(361, 70.0))
        f.set_phase_timing(timing)
        runs.append(pkg.vp_driver.VictoriaParkRun(f, data, P, seed=9).run(n_messages=900))
    a, b = runs
    assert a.n_lidar == b.n_lidar > 50 and a.n_resamples == b.n_resamples >= 2
    wa, wb = a.f.get_weights(), b.f.get_weights()
    assert np.all(np.isfinite(wa)) and np.array_equal(wa, wb)
    sa = a.f.gm_sizes()
    assert np.array_equal(sa, b.f.gm_sizes()) and sa.max() <= 192 and np.median(sa) >= 3
    for i in range(0, n, 97):
        for x, y in zip(a.f.export_gm(i), b.f.export_gm(i)):
            assert np.array_equal(x, y), i
    assert np.array_equal(a.f.get_particle_ids()[0], b.f.get_particle_ids()[0])
    for r in runs:
        r.f.close()


def test_victoria_park_dataset_extract_fastslam(pkg, ob, sc):
    """The same dataset extract through the FastSLAM event loop (src/fastslam_VictoriaPark.cpp: values of
    cfg/fastslam_VictoriaPark_artificialClutter.xml for the FastSLAM-specific keys) on the device and on the oracle."""
    import os
    data = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "victoria_park_extract.npz"))
    n = 32
    P = dict(sc.VP_PARAMS)
    runs = []
    for make in (lambda: pkg.RBPHDFilter(n, gm_capacity=192, model=pkg.capi.MODEL_VICTORIAPARK_3D),
                 lambda: ob.OracleFilter(n, model=pkg.capi.MODEL_VICTORIAPARK_3D)):
        f = make()
        sc.apply_vp_params(f, P, np.full(361, 70.0))
        cfg = f.default_fastslam_config()
        cfg.minLogMeasurementLikelihood = -15.0
        cfg.mapExistencePruneThreshold = -5.0
        cfg.landmarkCandidateMeasurementCountThreshold = 3
        cfg.landmarkCandidateCurrentMeasurementCountThreshold = 0
        cfg.landmarkCandidateMeasurementCheckThreshold = 5
        cfg.landmarkCandidateMeasurementSupportDist = 3.0
        f.set_fastslam_config(cfg)
        runs.append(pkg.vp_driver.VictoriaParkRun(f, data, P, seed=5, filter="fastslam").run(n_messages=900))
    dev, orc = runs
    assert dev.n_lidar == orc.n_lidar and dev.n_lidar > 50
    assert dev.n_resamples == orc.n_resamples
    np.testing.assert_allclose(dev.f.get_weights(), orc.f.get_weights(), rtol=1e-6)
    sizes = dev.f.gm_sizes()
    assert np.array_equal(sizes, orc.f.gm_sizes()) and sizes.max() > 3
    for i in range(n):
        sc.assert_gm_close(dev.f.export_gm(i), orc.f.export_gm(i), 1e-7, 1e-9)


def _killer_weights(tmp_path, n, q):
    """An adversarial weight sequence for libstdc++'s std::sort (McIlroy's adversary run against std::sort itself, quantised by q so
    that the heap-sorted ranges hold equal keys), from tests/support/stdsort_replay_test.cpp."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(str(tmp_path), "stdsort_replay_test")
    if not os.path.exists(exe):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-o", exe, os.path.join(root, "tests", "support", "stdsort_replay_test.cpp")])
    out = subprocess.run([exe, "killer", str(n), str(q)], capture_output=True, text=True, check=True).stdout.split()
    return np.array(out[:n], dtype=np.float64), out[-1] == "1"


@pytest.mark.parametrize("model", ["2d", "vp"])
def test_equal_weights_come_out_in_std_sort_order(pkg, ob, sc, tmp_path, model):
    """sortByWeight is the unstable std::sort (include/GaussianMixture.hpp:523-534); the device replays libstdc++'s partition phase to
    put equal weights in the same order (csrc/stdsort_replay.h).  Mixtures built to stress that: one particle per pattern, all
    landmarks outside the field of view (the update leaves their weights alone and appends nothing, so the sort sees exactly the
    injected sequence): 16 and 17 entries (the insertion-sort threshold), one long run of equal weights, a handful of distinct
    values, births at one weight among distinct ones, a sorted tied prefix, and adversarial sequences that drive std::sort into
    its depth limit (heap-sort branch).  Compared IN ORDER with the oracle (real std::sort) after the weighting phase's sort,
    after merge + prune, and through the fused step."""
    vp = model == "vp"
    nlm = 150 if vp else 300
    rng = np.random.default_rng(77)
    pats = []
    pats.append(np.full(16, 0.5)); pats.append(np.full(17, 0.5))
    pats.append(np.r_[np.full(17, 0.25), [0.9]])
    pats.append(np.full(nlm, 0.01))
    pats.append(rng.integers(1, 4, nlm) / 4.0)
    pats.append(rng.integers(1, 40, nlm) / 40.0)
    pats.append(np.where(rng.random(nlm) < 0.25, 0.01, rng.uniform(0.02, 1.0, nlm)))
    pats.append(np.r_[1.0 - 1e-3 * (np.arange(nlm // 2) // 3), rng.uniform(0.02, 1.0, nlm - nlm // 2)])
    n_heap = 0
    for n_k, q in ((nlm, 2), (nlm, 3), (100, 1), (64, 2), (nlm, 5)):
        w, hit = _killer_weights(tmp_path, n_k, q)
        n_heap += hit
        pats.append(0.02 + 0.9 * w / (w.max() + 1.0))
    assert n_heap >= 2, "no adversarial sequence reaches std::sort's depth limit"
    n = len(pats)
    scen = (sc.make_vp_scenario(n, nlm, 6, seed=3, frac_in_fov=0.0) if vp else sc.make_scenario(n, nlm, 8, seed=3, frac_in_fov=0.0))
    kw = dict(model=pkg.capi.MODEL_VICTORIAPARK_3D) if vp else {}
    outs = []
    for fused in (False, True):
        dev = pkg.RBPHDFilter(n, gm_capacity=448, **kw)
        orc = ob.OracleFilter(n, **kw)
        for f in (dev, orc):
            sc.load_scenario(f, scen, maps=False)
            for i, w in enumerate(pats):
                k = len(w)
                f.import_gm(i, w, scen["mean"][i][:k], scen["cov"][i][:k])
        if fused:
            dev.update_async(scen["Z"]); dev.synchronize()
            orc.update(scen["Z"])
        else:
            for f in (dev, orc):
                f.update_map(scen["Z"])
                f.importance_weighting()
            assert np.array_equal(dev.gm_sizes(), orc.gm_sizes())
            for i in range(n):
                gd, go = dev.export_gm(i), orc.export_gm(i)
                assert np.array_equal(gd[0], np.sort(pats[i])[::-1]), i         # the weights were left alone ...
                assert np.array_equal(gd[2], go[2]) and np.array_equal(gd[0], go[0]), ("order after sortByWeight", i)   # ... and the order is std::sort's
            for f in (dev, orc):
                f.merge(); f.prune()
        assert np.array_equal(dev.gm_sizes(), orc.gm_sizes())
        for i in range(n):
            sc.assert_gm_close(dev.export_gm(i), orc.export_gm(i), GM_RTOL, GM_ATOL, ordered=True)
        outs.append([dev.export_gm(i) for i in range(n)])
        dev.close()
    for a, b in zip(*outs):
        for k in (0, 2, 3):                                                  # (w, mean, cov; w_prev is scratch between updates)
            assert np.array_equal(a[k], b[k])                                # fused step == phase kernels, bit for bit


# ---- BASELINE.json full-size configurations: size-independent properties + oracle parity on a particle subset -------------

def _full_size_check(pkg, ob, sc, scen, cap, subset=24, check_murty=False, fused=None, fused_cap=None):
    n = scen["n"]
    dev = pkg.RBPHDFilter(n, gm_capacity=cap)
    sc.load_scenario(dev, scen)
    dev.update_map(scen["Z"])
    dev.importance_weighting()
    w_before_merge = np.array([dev.export_gm(i)[0].sum() for i in range(0, n, max(1, n // 50))])
    dev.merge()
    w_after_merge = np.array([dev.export_gm(i)[0].sum() for i in range(0, n, max(1, n // 50))])
    np.testing.assert_allclose(w_after_merge, w_before_merge, rtol=1e-12)      # merging conserves the total mixture weight
    dev.prune()
    sizes1 = dev.gm_sizes()
    dev.prune()
    assert np.array_equal(dev.gm_sizes(), sizes1)                                # prune is idempotent
    P = scen["params"]
    for i in range(0, n, max(1, n // 50)):
        w = dev.export_gm(i)[0]
        assert np.all(w >= P["prune_thr"]) and np.all(np.diff(w) <= 0)           # sorted by weight, nothing below threshold
    wd = dev.get_weights()
    assert np.all(np.isfinite(wd)) and np.all(wd > 0)
    # oracle parity on a subset of the particles (same per-particle inputs => same per-particle outputs)
    idx = np.linspace(0, n - 1, subset).astype(int)
    sub = dict(scen)
    sub.update(n=subset, poses=scen["poses"][idx], w=scen["w"][idx], mean=scen["mean"][idx], cov=scen["cov"][idx], particle_w=scen["particle_w"][idx])
    if np.ndim(scen["pose_cov"]) == 3:
        sub["pose_cov"] = scen["pose_cov"][idx]
    orc = ob.OracleFilter(subset)
    sc.load_scenario(orc, sub)
    orc.update(scen["Z"])
    if check_murty:
        assert orc.murty_calls() > 0
    np.testing.assert_allclose(wd[idx], orc.get_weights(), rtol=1e-8)
    for k, i in enumerate(idx):
        sc.assert_gm_close(dev.export_gm(int(i)), orc.export_gm(k), GM_RTOL, GM_ATOL, ordered=True)
    dev.close()
    # The same scenario through the launch bench.py TIMES: rfsgpu_step_async = ONE fused kernel + the post
    # kernel (Murty partitions, weight sums, division), at the full particle count -- the instantiation is checked by name --
    # against the same oracle subset: mixtures in order, particle weights after the global normalisation.
    if fused is not None:
        dev = pkg.RBPHDFilter(n, gm_capacity=fused_cap or cap)       # (bench.py's capacity for this workload)
        sc.load_scenario(dev, scen)
        dev.step_async(scen["Z"], normalize=True)
        dev.synchronize()
        assert dev.last_step_variant() == fused, dev.last_step_variant()
        wf = dev.get_weights()
        np.testing.assert_allclose(wf.sum(), 1.0, rtol=1e-12)
        np.testing.assert_allclose(wf, wd / wd.sum(), rtol=1e-12, atol=0)          # fused step == the four stand-alone kernels
        wo = orc.get_weights()
        np.testing.assert_allclose(wf[idx] / wf[idx].sum(), wo / wo.sum(), rtol=1e-8)
        assert np.array_equal(dev.gm_sizes(), sizes1)
        for k, i in enumerate(idx):
            sc.assert_gm_close(dev.export_gm(int(i)), orc.export_gm(k), GM_RTOL, GM_ATOL, ordered=True)
        dev.close()


def test_bench_two_ranks_on_one_gpu():
    """bench.py's N>1 path (rank-local shards, broadcast measurement set, all-reduce of the weight sums on the engine's
    stream, max-over-ranks timing, aggregate value) with two ranks sharing this box's GPU over gloo."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RFS_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--particles", "256", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=root, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # rank 0 prints exactly one JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 6 and d["warmup"] == 2
    assert d["config"]["particles_total"] == 512
    assert d["config"]["workload_key"] == "c2a"       # the same per-GPU shape as the N = 1 line (weak scaling of configs[1])
    assert abs(d["value"] - 2 * 6 / (d["ms_per_step"] * 6e-3)) < 1e-2 * d["value"]
    assert d["roofline"]["frac"] > 0


def test_full_size_c2(pkg, ob, sc):
    """configs[1]: 2000 particles x 200 GM landmarks x 30 measurements."""
    _full_size_check(pkg, ob, sc, sc.make_scenario(2000, 200, 30, seed=12345), cap=384, fused=(2, 1, 5, 1))   # phd_step_fused_kernel<2, true, 5>


def test_full_size_c3_shard(pkg, ob, sc):
    """configs[2], one GPU's shard: 2500 particles x 500 GM landmarks x 30 measurements (range limit 5 m)."""
    _full_size_check(pkg, ob, sc, sc.make_scenario(2500, 500, 30, seed=777, rmax=5.0), cap=704, fused=(3, 0, 6, 1), fused_cap=640)   # <3, false, 6>


def test_full_size_c4_victoria_park(pkg, ob, sc):
    """configs[3]: Victoria Park model (3-D landmarks, scan-based Pd, artificial-clutter parameters), 5000 particles.
    Two predict/update/normalise cycles at full size on the device; size-independent properties on every particle and
    oracle parity on a subset of the particles (each particle's outputs depend on its own inputs only).  (The synthetic
    state spreads the raw weights over > 30 decades, so a third cycle already hits 0 / inf in the reference arithmetic.)"""
    n, subset = 5000, 24
    scen = sc.make_vp_scenario(n, 40, 12, seed=4321, scan="ragged")
    dev = pkg.RBPHDFilter(n, device_id=0, gm_capacity=192, model=pkg.capi.MODEL_VICTORIAPARK_3D)
    sc.load_scenario(dev, scen)
    idx = np.linspace(0, n - 1, subset).astype(int)
    sub = dict(scen)
    sub.update(n=subset, poses=scen["poses"][idx], w=scen["w"][idx], mean=scen["mean"][idx], cov=scen["cov"][idx],
               particle_w=scen["particle_w"][idx])
    if np.ndim(scen["pose_cov"]) == 3:
        sub["pose_cov"] = scen["pose_cov"][idx]
    orc = ob.OracleFilter(subset, model=pkg.capi.MODEL_VICTORIAPARK_3D)
    sc.load_scenario(orc, sub)
    rng = np.random.default_rng(9)
    for step in range(2):
        Z = scen["Z"] + rng.normal(0, 1e-3, scen["Z"].shape)
        for f in (dev, orc):
            f.predict_map(True)
            f.update(Z)
        wd, wo = dev.get_weights(), orc.get_weights()
        assert np.all(np.isfinite(wd)) and np.all(wd >= 0) and (wd > 0).mean() > 0.9
        np.testing.assert_allclose(wd[idx], wo, rtol=1e-8, atol=1e-300)
        s = dev.weight_sums()
        np.testing.assert_allclose(s[0], wd.sum(), rtol=1e-12)
        dev.normalize_weights(s[0])
        orc.set_weights(wo / wd.sum())            # same global normaliser as the full-size filter
        np.testing.assert_allclose(dev.get_weights().sum(), 1.0, rtol=1e-12)
        sizes = dev.gm_sizes()
        assert sizes.min() > 0 and sizes.max() <= 192
        P = scen["params"]
        for k, i in enumerate(idx):
            g = dev.export_gm(int(i))
            assert np.all(g[0] >= P["prune_thr"]) and np.all(np.diff(g[0]) <= 0)
            sc.assert_gm_close(g, orc.export_gm(k), GM_RTOL, GM_ATOL, ordered=True)
    dev.close()


def test_full_size_c5_murty_stress(pkg, ob, sc):
    """configs[4]: 1000 particles x 50 measurements, 40 evaluation points, 10-sigma weighting gate -> Murty-200 partitions."""
    scen = sc.make_scenario(1000, 200, 50, seed=555, n_clutter=10, n_eval=40, weighting_md=10.0, weights=(0.8, 1.0))
    _full_size_check(pkg, ob, sc, scen, cap=448, subset=16, check_murty=True, fused=(3, 1, 6, 1))   # (1000 particles: every three-wave workgroup resident)


def test_murty_search_with_solver_waves_is_deterministic(pkg, sc):
    """configs[4]'s shape at 200 particles, the same update four times from the same state, on both launch paths: the Murty jobs run
    one searching wave + three solver waves per job with flags instead of barriers (murty.h, murty_kbest_async), and a race there
    would show as weights that differ from run to run.  (The values themselves are checked against the oracle above.)"""
    scen = sc.make_scenario(200, 200, 50, seed=556, n_clutter=10, n_eval=40, weighting_md=10.0, weights=(0.8, 1.0))
    dev = pkg.RBPHDFilter(200, gm_capacity=448)
    sc.load_scenario(dev, scen)
    dev.save_state()
    runs = []
    for timing in (False, True, False, True):
        dev.restore_state()
        dev.set_phase_timing(timing)
        dev.update(scen["Z"])
        runs.append(dev.get_weights().copy())
    dev.close()
    assert np.all(np.isfinite(runs[0])) and np.ptp(runs[0]) > 0
    for r in runs[1:]:
        assert np.array_equal(r, runs[0])


def test_warm_started_murty_children_give_the_sums_of_the_solver_from_scratch(pkg, ob, sc, tmp_path):
    """Round 6: in the RB-PHD partition sums a child of a Murty expansion is solved by ONE augmentation from its
    parent's dual variables (csrc/hungarian_wave.h, hungarian_warm_wave) instead of a solve from scratch.  It returns an optimal
    assignment of the child's table -- where several are optimal to within rounding not necessarily the one the reference's solver
    picks, which the k best SCORES (all that include/RBPHDFilter.hpp:948-959 sums) do not depend on.  A library built with
    -DMURTY_WARM=0 (rfs-slam_amd/build.py: build_cold_murty_variant, test support: every child from scratch by the reference-faithful
    solver, what rounds 1-5 shipped) runs a C5-shaped update in a child process: the shipped library's particle weights equal its
    weights to 1e-12, and both the oracle's to the usual tolerance."""
    import os
    import subprocess
    import sys
    cold = pkg.build_mod.COLD_LIB
    if not os.path.exists(cold):
        pytest.skip("tests/support/_build/librfsgpu_coldmurty.so was not built (built by __graft_entry__.build())")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    kw = dict(seed=557, n_clutter=10, n_eval=40, weighting_md=10.0, weights=(0.8, 1.0))
    out = os.path.join(str(tmp_path), "w.npy")
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import __graft_entry__ as g; pkg = g.load_package(); pkg.engine.LIB = %r; "
            "sc = pkg.scenarios; scen = sc.make_scenario(96, 200, 50, **%r); f = pkg.RBPHDFilter(96, gm_capacity=448); sc.load_scenario(f, scen); "
            "f.update_map(scen['Z']); f.importance_weighting(); np.save(%r, f.get_weights()); print('cold ok')" % (root, cold, kw, out))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "cold ok" in p.stdout, p.stderr[-3000:]
    w_cold = np.load(out)
    # the same search with only 16 positions of the open-node array in LDS (the rest in the job's arena: the overflow path): same bits
    w_smallq = None
    smallq = pkg.build_mod.SMALLQ_LIB
    if os.path.exists(smallq):
        p = subprocess.run([sys.executable, "-c", code.replace(cold, smallq)], capture_output=True, text=True, timeout=600)
        assert p.returncode == 0 and "cold ok" in p.stdout, p.stderr[-3000:]
        w_smallq = np.load(out)
    scen = sc.make_scenario(96, 200, 50, **kw)
    dev, orc = make_pair(pkg, ob, sc, scen, cap=448)
    for f in (dev, orc):
        f.update_map(scen["Z"])
        f.importance_weighting()
    assert orc.murty_calls() > 50, "scenario does not reach the Murty path"
    np.testing.assert_allclose(dev.get_weights(), w_cold, rtol=1e-12, atol=0.0)
    if w_smallq is not None:
        assert np.array_equal(dev.get_weights(), w_smallq)
    np.testing.assert_allclose(w_cold / w_cold.sum(), orc.get_weights() / orc.get_weights().sum(), rtol=1e-9, atol=1e-300)
    compare_weights(dev, orc)


def test_randomised_differential_run():
    """tools/fuzz_parity.py: random shapes (1..330 landmarks around the 64-entry chunk boundaries), ranges, SC-PHD / multi-
    feature, tied / quantised / sub-fp32 weights, two update cycles each, fused and three-kernel path, device vs oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "40", "2024"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert ", 0 failures" in out.stdout


def test_randomised_differential_run_fastslam():
    """tools/fuzz_fastslam.py: random FastSLAM / MH-FastSLAM set-ups (1..5 hypotheses, likelihood windows, candidate thresholds),
    four cycles with particle growth and resample(nParticles_init), device vs oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_fastslam.py"), "40", "2025"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert ", 0 failures" in out.stdout


def _vp_cpp(pkg, *args):
    import os
    import subprocess
    pkg.build_mod.build_host()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [pkg.build_mod.SIM_VP, "-c", os.path.join(root, "tests", "golden", "rbphdslam_VictoriaPark_c4.xml"),
           "-d", os.path.join(root, "tests", "golden", "vp_extract")] + [str(a) for a in args]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-500:]
    return out.stdout


def test_cpp_victoria_park_driver_matches_the_python_loop(pkg, sc, tmp_path):
    """host/rbphdslam_vp (C++: XML configuration of the reference's cfg/rbphdslam_VictoriaPark_artificialClutter.xml, the
    dataset's text files, Ackerman model, rfs_amd::RBPHDFilterVP) against rfs_slam_amd.vp_driver on the same 900 messages
    with the host randomness switched off (no input noise, no artificial clutter: all particles stay identical, nothing
    depends on a random stream): the best particle's final pose and map must agree."""
    import os
    import re
    n = 16
    _vp_cpp(pkg, "-n", n, "-e", n / 2.0, "-o", tmp_path, "--no-input-noise", "--no-clutter")
    rows = [ln.split() for ln in open(os.path.join(tmp_path, "finalMap.dat")) if not ln.startswith("#")]
    cpp = np.array(rows, dtype=np.float64)
    pose = np.array(re.search(r"# pose (.*)", open(os.path.join(tmp_path, "finalMap.dat")).read()).group(1).split(), dtype=np.float64)
    data = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "victoria_park_extract.npz"))
    P = dict(sc.VP_PARAMS)
    f = pkg.RBPHDFilter(n, gm_capacity=192, model=pkg.capi.MODEL_VICTORIAPARK_3D)
    sc.apply_vp_params(f, P, np.full(361, 70.0))
    run = pkg.vp_driver.VictoriaParkRun(f, data, P, seed=1, var_uv=0.0, var_ur=0.0, added_clutter=0.0).run(n_messages=900)
    w, wp, mean, cov = f.export_gm(0)
    assert cpp.shape[0] == len(w) and len(w) > 3
    np.testing.assert_allclose(pose, run.x[0], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(cpp[:, 0], w, rtol=1e-9)
    np.testing.assert_allclose(cpp[:, 1:4], mean, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(cpp[:, 4:], np.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], 1), rtol=1e-8, atol=1e-12)


def test_cpp_victoria_park_driver_end_to_end(pkg, tmp_path):
    """The same driver with everything on (input noise, Poisson(3) artificial clutter, resampling), 200 particles: runs the 900
    messages, resamples, keeps a map, writes the reference's log formats."""
    import os
    import re
    out = _vp_cpp(pkg, "-n", 200, "-s", 3, "-o", tmp_path)
    m = re.search(r"RESULT lidar=(\d+) resamples=(\d+) best=(\d+) map=(\d+) strong=(\d+)", out)
    assert m, out[-1500:]
    lidar, resamples, _, nmap, strong = [int(g) for g in m.groups()]
    assert lidar > 50 and resamples > 0 and nmap > 3 and strong > 0
    pose_rows = open(os.path.join(tmp_path, "particlePose.dat")).read().splitlines()
    assert len(pose_rows) == lidar * 200 and len(pose_rows[0].split()) == 6
    assert len(open(os.path.join(tmp_path, "landmarkEst.dat")).readline().split()) == 8


def test_vp_pd_probe_matches_oracle_incl_indefinite_covariances(pkg, ob, sc):
    """MeasurementModel_VictoriaPark::probabilityOfDetection per Gaussian, device (rfsgpu_vp_probe_pd: what the kernels evaluate)
    vs oracle: thin and uncertain landmarks (dozens of laterally shifted copies), and INDEFINITE covariances, where the
    reference's `std::max(3 * sqrt(negative), 0.2)` keeps the NaN (no shifted copies at all) -- `fmax` would not."""
    scen = sc.make_vp_scenario(6, 70, 9, seed=77, scan="ragged")
    scen["mean"][:, ::3, 2] = 0.06                     # thin
    scen["cov"][:, ::3, 0, 0] *= 300.0                 # uncertain
    scen["cov"][:, ::3, 1, 1] *= 300.0
    scen["cov"][:, 1::7, 0, 0] = -0.7                  # indefinite: perp' Sigma perp < 0 for most directions
    scen["cov"][:, 1::7, 1, 1] = -0.3
    dev, orc = make_vp_pair(pkg, ob, sc, scen)
    differs_from_fmax = 0
    for i in range(scen["n"]):
        pd_d, cl_d = dev.vp_probe_pd(i)
        pd_o, cl_o = orc.vp_probe_pd(i)
        assert len(pd_d) == 70
        assert np.array_equal(pd_d, pd_o), (i, np.nonzero(pd_d != pd_o)[0])
        assert np.array_equal(cl_d, cl_o), (i, np.nonzero(cl_d != cl_o)[0])
        differs_from_fmax += int(np.any(pd_d[1::7] != pd_d[1::7][0]))
    assert np.any(dev.vp_probe_pd(0)[0] > 0)
