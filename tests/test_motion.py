"""Device-side ParticleFilter::propagate for the Ackerman model (csrc/motion.h, rfsgpu_propagate_ackerman_async).

The normal deviates come from Philox4x32-10 (Salmon et al., SC'11).  The numpy restatement below is pinned to the published
known-answer vectors of the Random123 distribution (CPU test); the GPU test then predicts the device's poses from it."""
import numpy as np
import pytest

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32_10(ctr, key):
    c = [np.asarray(x, dtype=np.uint64) for x in ctr]
    k0, k1 = int(key[0]), int(key[1])
    for _ in range(10):
        p0 = np.uint64(M0) * c[0]
        p1 = np.uint64(M1) * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0), p1 & np.uint64(MASK), (p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1), p0 & np.uint64(MASK)]
        k0 = (k0 + W0) & MASK
        k1 = (k1 + W1) & MASK
    return c


def u01(a, b):
    m = ((a << np.uint64(32)) | b) >> np.uint64(11)
    return (m.astype(np.float64) + 1.0) * (1.0 / 9007199254740992.0)


def ackerman_reference(x, u, var, dt, geom, seed, call):
    """numpy restatement of motion.h (MotionModel_Ackerman2d::step, reference src/ProcessModel_Ackerman2D.cpp:47-78)."""
    n = x.shape[0]
    uv = np.full(n, u[0], dtype=np.float64)
    ur = np.full(n, u[1], dtype=np.float64)
    if var is not None and (var[0] != 0 or var[1] != 0):
        i = np.arange(n, dtype=np.uint64)
        z = np.zeros(n, dtype=np.uint64)
        r = philox4x32_10([i, z, z + np.uint64(call & MASK), z + np.uint64(call >> 32)], (seed & MASK, seed >> 32))
        u1, u2 = u01(r[0], r[1]), u01(r[2], r[3])
        rad, ang = np.sqrt(-2.0 * np.log(u1)), 2.0 * np.pi * u2
        uv = uv + np.sqrt(var[0]) * (rad * np.cos(ang))
        ur = ur + np.sqrt(var[1]) * (rad * np.sin(ang))
    h, l, dx, dy = geom
    c, s, t = np.cos(x[:, 2]), np.sin(x[:, 2]), np.tan(ur)
    v = uv / (1 - t * h / l)
    out = x.copy()
    out[:, 0] = x[:, 0] + dt * (v * c - v / l * t * (dx * s + dy * c))
    out[:, 1] = x[:, 1] + dt * (v * s + v / l * t * (dx * c - dy * s))
    th = x[:, 2] + dt * v / l * t
    th = np.where(th > np.pi, th - 2 * np.pi, np.where(th < -np.pi, th + 2 * np.pi, th))
    out[:, 2] = th
    return out


def test_philox_known_answers():
    """Random123's kat_vectors for philox4x32, 10 rounds."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((MASK, MASK, MASK, MASK), (MASK, MASK), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = philox4x32_10([np.array([c], dtype=np.uint64) for c in ctr], key)
        assert tuple(int(g[0]) for g in got) == want


GEOM = (0.76, 2.83, 3.78, 0.50)


@pytest.mark.gpu
def test_device_propagation_matches_the_restatement():
    from __graft_entry__ import load_package
    pkg = load_package()
    n = 5000
    rng = np.random.default_rng(3)
    x0 = np.column_stack([rng.uniform(-50, 50, n), rng.uniform(-50, 50, n), rng.uniform(-np.pi, np.pi, n)])
    f = pkg.RBPHDFilter(n, gm_capacity=64, model=pkg.capi.MODEL_VICTORIAPARK_3D)
    f.set_poses(x0)
    # noise-free: the model alone (device sin / cos / tan against numpy's)
    f.propagate_ackerman_async((3.1, 0.07), None, 0.025, GEOM, seed=1, call=0)
    f.synchronize()
    np.testing.assert_allclose(f.get_poses(), ackerman_reference(x0, (3.1, 0.07), None, 0.025, GEOM, 1, 0), rtol=1e-13, atol=1e-13)
    # with input noise: every particle's two deviates are Philox block (particle, call) under the seed
    f.set_poses(x0)
    seed, var = 0x1234567890ABCDEF, (0.2, 0.025)
    want = x0
    for call in (0, 1, 7, 1 << 33):
        f.propagate_ackerman_async((3.1, 0.07), var, 0.025, GEOM, seed=seed, call=call)
        want = ackerman_reference(want, (3.1, 0.07), var, 0.025, GEOM, seed, call)
    f.synchronize()
    got = f.get_poses()
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-10)
    # and the deviates are standard normal: mean / spread of the noisy speed recovered from the pose increments of one call
    f.set_poses(np.zeros((n, 3)))
    f.propagate_ackerman_async((5.0, 0.0), (0.2, 0.0), 1.0, (0.0, 2.83, 0.0, 0.0), seed=99, call=5)
    f.synchronize()
    v = f.get_poses()[:, 0]          # x = dt * v * cos(0): the drawn speeds
    assert abs(v.mean() - 5.0) < 4 * np.sqrt(0.2 / n) and abs(v.var() - 0.2) < 0.02
    # another seed, another draw
    f.set_poses(np.zeros((n, 3)))
    f.propagate_ackerman_async((5.0, 0.0), (0.2, 0.0), 1.0, (0.0, 2.83, 0.0, 0.0), seed=100, call=5)
    f.synchronize()
    assert np.abs(f.get_poses()[:, 0] - v).max() > 0.1


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["vp", "2d"])
def test_static_steps_in_one_launch_give_the_same_bits(model):
    """rfsgpu_static_steps_async(n, Q[n]) == n times (set_lmk_process_noise(Q_k); predict_map_async(0)): Sigma += Q_k in order, bit for bit."""
    from __graft_entry__ import load_package
    pkg = load_package()
    sc = pkg.scenarios
    fs = []
    for batched in (False, True):
        if model == "vp":
            f = pkg.RBPHDFilter(24, gm_capacity=64, model=pkg.capi.MODEL_VICTORIAPARK_3D)
            sc.apply_vp_params(f, dict(sc.VP_PARAMS), np.full(361, 70.0))
            rng = np.random.default_rng(1)
            for i in range(24):
                n = 5 + i % 7
                mean = rng.normal(0, 20, (n, 3))
                A = rng.normal(0, 1, (n, 3, 3))
                cov = A @ A.transpose(0, 2, 1) * 0.01 + np.eye(3) * 1e-3
                f.import_gm(i, rng.uniform(0.1, 1, n), mean, cov)
        else:
            scen = sc.make_scenario(24, 30, 6, seed=3)
            f = pkg.RBPHDFilter(24, gm_capacity=64)
            sc.load_scenario(f, scen)
        D = 3 if model == "vp" else 2
        Qs = [np.diag(np.arange(1, D + 1) * 1e-4 * (1 + 0.37 * k)) + 1e-6 * (k % 3) * (np.ones((D, D)) - np.eye(D)) for k in range(21)]
        if batched:
            f.static_steps_async(19, np.array(Qs[:19]))          # longer than one kernel-argument block: two launches
            f.static_steps_async(0, None)
            f.set_lmk_process_noise(Qs[19])
            f.static_steps_async(2, None)                        # the noise that is set now, twice
        else:
            for k in range(19):
                f.set_lmk_process_noise(Qs[k])
                f.predict_map_async(False)
            f.set_lmk_process_noise(Qs[19])
            f.predict_map_async(False)
            f.predict_map_async(False)
        f.synchronize()
        fs.append(f)
    a, b = fs
    for i in range(24):
        for x, y in zip(a.export_gm(i), b.export_gm(i)):
            assert np.array_equal(x, y)


@pytest.mark.gpu
def test_a_run_of_propagations_in_one_launch_gives_the_same_bits():
    """rfsgpu_propagate_ackerman_run_async(n, ...) == n calls of rfsgpu_propagate_ackerman_async with calls call0, call0 + 1, ..."""
    from __graft_entry__ import load_package
    pkg = load_package()
    n = 3000
    rng = np.random.default_rng(11)
    x0 = np.column_stack([rng.uniform(-50, 50, n), rng.uniform(-50, 50, n), rng.uniform(-np.pi, np.pi, n)])
    steps = 21                                                   # more than one kernel-argument block
    u = np.column_stack([rng.uniform(0, 6, steps), rng.uniform(-0.3, 0.3, steps)])
    var = np.column_stack([np.full(steps, 0.2), np.full(steps, 0.025)])
    var[4] = 0.0                                                 # a noise-free step in between
    dt = rng.uniform(0.01, 0.05, steps)
    out = []
    for batched in (False, True):
        f = pkg.RBPHDFilter(n, gm_capacity=64, model=pkg.capi.MODEL_VICTORIAPARK_3D)
        f.set_poses(x0)
        if batched:
            f.propagate_ackerman_run_async(u, var, dt, GEOM, seed=5, call0=40)
        else:
            for k in range(steps):
                f.propagate_ackerman_async(u[k], var[k], dt[k], GEOM, seed=5, call=40 + k)
        f.synchronize()
        out.append(f.get_poses())
    assert np.array_equal(out[0], out[1])
    want = x0
    for k in range(steps):
        want = ackerman_reference(want, u[k], var[k], dt[k], GEOM, 5, 40 + k)
    np.testing.assert_allclose(out[1], want, rtol=1e-9, atol=1e-9)
