"""Seeded synthetic inputs for the RB-PHD update path (configs C1-like / C2 / C3 / C5).

Input generation only -- shared by tests, smoke() and bench.py.  A scenario is the state the reference
driver would hand to RBPHDFilter::update(): per-particle poses (+ pose covariance), per-particle
Gaussian-mixture maps, one measurement set, and the three config structs.
"""
import numpy as np

from . import capi

# shipped cfg/rbphdslam2dSim.xml values (reference cfg/rbphdslam2dSim.xml:33-86)
C1_PARAMS = dict(
    R=np.diag([5e-4, 5e-5]) * 10.0, Pd=0.99, clutter=1e-4, rmax=2.5, rmin=0.5, rbuf=0.05,
    kf_range=1.0, kf_bearing=0.2, new_gaussian_md=3.0, n_eval=15, min_weight=0.75, weighting_md=3.0,
    merge_thr=0.5, merge_infl=1.5, prune_thr=0.01, birth_w=0.01, use_cluster=0,
    Q_lm=np.diag([2e-4, 2e-4]) * 0.01, pose_cov=np.diag([3e-5, 3e-5, 3e-5]),
)


def make_scenario(n_particles, n_landmarks, n_z, seed=12345, n_clutter=None, frac_in_fov=1.0, params=None,
                  rmax=None, weights=(0.3, 1.0), n_eval=None, weighting_md=None, use_cluster=None, per_particle_pose_cov=False):
    """C2-style state: a shared ground-truth set uniform by area in the sensing annulus;
    per particle n_landmarks Gaussians mu = gt + N(0,0.02^2), Sigma = diag(s^2), s ~ U[0.02,0.1],
    w ~ U(weights); poses = origin + N(0, diag(.02,.02,.01)^2); measurements = noisy detections of a
    ground-truth subset + uniform clutter.  frac_in_fov < 1 places the remaining landmarks outside rmax
    (Pd = 0, untouched by the update: the C2b steady-state shape)."""
    P = dict(C1_PARAMS)
    if params:
        P.update(params)
    if rmax is not None:
        P["rmax"] = rmax
    if n_eval is not None:
        P["n_eval"] = n_eval
    if weighting_md is not None:
        P["weighting_md"] = weighting_md
    if use_cluster is not None:
        P["use_cluster"] = int(use_cluster)
    rng = np.random.default_rng(seed)
    rmin, rmx = P["rmin"], P["rmax"]
    n_in = int(round(n_landmarks * frac_in_fov))
    # keep ground truth away from the buffer zones so Pd/near-limit decisions have margin
    lo, hi = rmin + 3 * P["rbuf"], rmx - 3 * P["rbuf"]
    r = np.sqrt(rng.uniform(lo * lo, hi * hi, n_in))
    a = rng.uniform(-np.pi, np.pi, n_in)
    gt_in = np.stack([r * np.cos(a), r * np.sin(a)], 1)
    n_out = n_landmarks - n_in
    r2 = np.sqrt(rng.uniform((rmx + 1.0) ** 2, (rmx + 4.0) ** 2, n_out))
    a2 = rng.uniform(-np.pi, np.pi, n_out)
    gt = np.concatenate([gt_in, np.stack([r2 * np.cos(a2), r2 * np.sin(a2)], 1)], 0)
    order = rng.permutation(n_landmarks)
    gt = gt[order]
    in_fov = np.zeros(n_landmarks, bool)
    in_fov[np.nonzero(order < n_in)[0]] = True

    poses = rng.normal(0, 1, (n_particles, 3)) * np.array([0.02, 0.02, 0.01])
    means = gt[None] + rng.normal(0, 0.02, (n_particles, n_landmarks, 2))
    s = rng.uniform(0.02, 0.1, (n_particles, n_landmarks, 2))
    covs = np.zeros((n_particles, n_landmarks, 2, 2))
    covs[..., 0, 0] = s[..., 0] ** 2
    covs[..., 1, 1] = s[..., 1] ** 2
    rho = rng.uniform(-0.3, 0.3, (n_particles, n_landmarks))
    covs[..., 0, 1] = covs[..., 1, 0] = rho * s[..., 0] * s[..., 1]
    w = rng.uniform(weights[0], weights[1], (n_particles, n_landmarks))

    if n_clutter is None:
        n_clutter = max(1, n_z // 5) if n_z > 1 else 0
    n_det = min(n_z - n_clutter, int(in_fov.sum()))
    n_clutter = n_z - n_det
    det_idx = rng.choice(np.nonzero(in_fov)[0], n_det, replace=False) if n_det > 0 else np.zeros(0, int)
    Rm = np.asarray(P["R"])
    zr = np.hypot(gt[det_idx, 0], gt[det_idx, 1]) + rng.normal(0, np.sqrt(Rm[0, 0]), n_det)
    zb = np.arctan2(gt[det_idx, 1], gt[det_idx, 0]) + rng.normal(0, np.sqrt(Rm[1, 1]), n_det)
    cr = rng.uniform(rmin, rmx, n_clutter)
    cb = rng.uniform(-np.pi, np.pi, n_clutter)
    Z = np.concatenate([np.stack([zr, zb], 1), np.stack([cr, cb], 1)], 0)
    Z = Z[rng.permutation(n_z)] if n_z > 0 else Z.reshape(0, 2)

    pose_cov = np.asarray(P["pose_cov"], dtype=np.float64)
    if per_particle_pose_cov:
        pose_cov = pose_cov[None] * rng.uniform(0.5, 2.0, (n_particles, 1, 1))
    return dict(n=n_particles, nM=n_landmarks, poses=poses, pose_cov=pose_cov, w=w, mean=means, cov=covs, Z=Z,
                particle_w=np.ones(n_particles), params=P, gt=gt, in_fov=in_fov)


def apply_params(f, P):
    """Push the reference's three config structs through the ABI (src/rbphdslam2dSim.cpp:444-492)."""
    cfg = f.default_filter_config()
    cfg.birthGaussianWeight = P["birth_w"]
    cfg.newGaussianCreateInnovMDThreshold = P["new_gaussian_md"]
    cfg.importanceWeightingEvalPointCount = P["n_eval"]
    cfg.importanceWeightingEvalPointGuassianWeight = P["min_weight"]
    cfg.importanceWeightingMeasurementLikelihoodMDThreshold = P["weighting_md"]
    cfg.gaussianMergingThreshold = P["merge_thr"]
    cfg.gaussianMergingCovarianceInflationFactor = P["merge_infl"]
    cfg.gaussianPruningThreshold = P["prune_thr"]
    cfg.useClusterProcess = P["use_cluster"]
    cfg.minUpdatesBeforeResample = P.get("min_updates", 2)
    f.set_filter_config(cfg)
    if hasattr(f, "config"):
        f.config = cfg
    f.set_model_rngbrg(P["R"], P["Pd"], P["clutter"], P["rmax"], P["rmin"], P["rbuf"])
    f.set_kf_config(P["kf_range"], P["kf_bearing"])
    f.set_lmk_process_noise(P["Q_lm"])


def load_scenario(f, scen, maps=True):
    """(Re-)inject a scenario's state into a filter handle."""
    if scen.get("model") == "vp":
        apply_vp_params(f, scen["params"], scen["scan"])
    else:
        apply_params(f, scen["params"])
    f.set_poses(scen["poses"], scen["pose_cov"])
    f.set_weights(scen["particle_w"])
    if maps:
        for i in range(scen["n"]):
            f.import_gm(i, scen["w"][i], scen["mean"][i], scen["cov"][i])


def match_gm(a, b, rtol=1e-10, atol=1e-12):
    """Multiset comparison of two exported mixtures (w, w_prev, mean, cov): returns max abs/rel error or raises."""
    wa, _, ma, ca = a
    wb, _, mb, cb = b
    assert wa.size == wb.size, f"GM size {wa.size} != {wb.size}"
    if wa.size == 0:
        return 0.0
    ca = (ca + np.swapaxes(ca, -1, -2)) / 2   # the device stores Sigma packed-symmetric
    cb = (cb + np.swapaxes(cb, -1, -2)) / 2
    A = np.concatenate([wa[:, None], ma, ca.reshape(wa.size, -1)], 1)
    B = np.concatenate([wb[:, None], mb, cb.reshape(wb.size, -1)], 1)
    ia = np.lexsort(np.round(A, 9).T[::-1])
    ib = np.lexsort(np.round(B, 9).T[::-1])
    A, B = A[ia], B[ib]
    if not np.allclose(A, B, rtol=rtol, atol=atol):
        # fall back to greedy nearest matching (near-tied sort keys)
        used = np.zeros(len(B), bool)
        for row in A:
            d = np.max(np.abs(B - row) / (atol + rtol * np.abs(row)), axis=1)
            d[used] = np.inf
            j = int(np.argmin(d))
            assert d[j] <= 1.0, f"unmatched Gaussian {row}"
            used[j] = True
        return 1.0
    return float(np.max(np.abs(A - B)))


def assert_gm_close(a, b, rtol=1e-10, atol=1e-12, ordered=False):
    if ordered:
        for x, y in zip((a[0], a[2], a[3]), (b[0], b[2], b[3])):
            assert x.shape == y.shape
            np.testing.assert_allclose(x, y, rtol=rtol, atol=atol)
    else:
        match_gm(a, b, rtol, atol)


# ---- Victoria Park (config 4): parameter values of the reference's cfg/rbphdslam_VictoriaPark_artificialClutter.xml ----
VP_PARAMS = dict(
    R=np.diag([0.025, 2.5e-5, 0.002]) * 40.0, Slb=1e-5, pd_table=[0.0, 0.05, 0.35, 0.76, 0.89, 0.90], expected_clutter=6.0,
    rmax=70.0, rmin=5.0, bmax=np.deg2rad(177.0), bmin=np.deg2rad(6.3025), buffer_pd=0.4,
    kf_range=7.5, kf_bearing=0.2, new_gaussian_md=3.0, n_eval=15, min_weight=0.75, weighting_md=3.0,
    merge_thr=1.0, merge_infl=1.5, prune_thr=0.01, birth_w=0.01, use_cluster=0,
    birth_count_thr=5, birth_check_thr=10, birth_support_dist=2.0, birth_cur_thr=2,
    Q_lm=np.diag([5e-4, 5e-4, 1e-4]) * 0.025 ** 2, min_updates=2, min_measurements=15,
)


def make_vp_scenario(n_particles, n_landmarks, n_z, seed=4242, scan="const", params=None, weights=(0.3, 1.0), frac_in_fov=0.8):
    """Victoria-Park-shaped state: landmarks (x, y, trunk diameter) in front of the vehicle, measurements
    (range, bearing, diameter); the laser scan is the synthetic constant-70 m one of SURVEY §8d (the dataset's LASER.txt is
    missing) or a ragged one that exercises the occlusion count."""
    P = dict(VP_PARAMS)
    if params:
        P.update(params)
    rng = np.random.default_rng(seed)
    n_in = int(round(n_landmarks * frac_in_fov))
    r = rng.uniform(8.0, 60.0, n_landmarks)
    b = rng.uniform(np.deg2rad(15), np.deg2rad(165), n_landmarks)
    out = np.arange(n_landmarks) >= n_in
    r[out] = rng.uniform(80.0, 120.0, out.sum())          # beyond the range limit
    d = rng.uniform(0.3, 1.5, n_landmarks)
    th0 = 0.3                                              # vehicle heading; the sensor frame is heading - pi/2
    gt = np.stack([r * np.cos(th0 - np.pi / 2 + b), r * np.sin(th0 - np.pi / 2 + b), d], 1)
    perm = rng.permutation(n_landmarks)
    gt, r, b, out = gt[perm], r[perm], b[perm], out[perm]
    poses = np.array([0.0, 0.0, th0]) + rng.normal(0, 1, (n_particles, 3)) * np.array([0.05, 0.05, 0.004])
    means = gt[None] + rng.normal(0, 1, (n_particles, n_landmarks, 3)) * np.array([0.08, 0.08, 0.03])
    A = rng.normal(0, 1, (n_particles, n_landmarks, 3, 3)) * np.array([0.25, 0.25, 0.08])[None, None, :, None]
    covs = A @ np.swapaxes(A, -1, -2) + np.diag([0.01, 0.01, 0.002])
    w = rng.uniform(weights[0], weights[1], (n_particles, n_landmarks))
    n_clutter = max(1, n_z // 4) if n_z > 1 else 0
    n_det = min(n_z - n_clutter, int((~out).sum()))
    n_clutter = n_z - n_det
    det = rng.choice(np.nonzero(~out)[0], n_det, replace=False) if n_det > 0 else np.zeros(0, int)
    Rm = np.asarray(P["R"]) / 40.0
    Zd = np.stack([r[det] + rng.normal(0, np.sqrt(Rm[0, 0]), n_det), b[det] + rng.normal(0, np.sqrt(Rm[1, 1]), n_det),
                   gt[det, 2] + rng.normal(0, np.sqrt(Rm[2, 2]), n_det)], 1)
    Zc = np.stack([rng.uniform(P["rmin"], P["rmax"], n_clutter), rng.uniform(P["bmin"], P["bmax"], n_clutter), np.ones(n_clutter)], 1)
    Z = np.concatenate([Zd, Zc], 0)
    Z = Z[rng.permutation(n_z)] if n_z > 0 else Z.reshape(0, 3)
    if isinstance(scan, str) and scan == "const":
        scan = np.full(361, 70.0)
    elif isinstance(scan, str):
        scan = rng.uniform(20.0, 80.0, 361)
        scan[rng.integers(0, 361, 20)] = 0.0               # no-return beams count as visible
    return dict(n=n_particles, nM=n_landmarks, poses=poses, pose_cov=np.zeros((3, 3)), w=w, mean=means, cov=covs, Z=Z,
                particle_w=np.ones(n_particles), params=P, gt=gt, scan=np.asarray(scan, dtype=np.float64), model="vp")


def apply_vp_params(f, P, scan):
    cfg = f.default_filter_config()
    cfg.birthGaussianWeight = P["birth_w"]
    cfg.birthGaussianMeasurementCountThreshold = P["birth_count_thr"]
    cfg.birthGaussianMeasurementCheckThreshold = P["birth_check_thr"]
    cfg.birthGaussianMeasurementSupportDist = P["birth_support_dist"]
    cfg.birthGaussianCurrentMeasurementCountThreshold = P["birth_cur_thr"]
    cfg.newGaussianCreateInnovMDThreshold = P["new_gaussian_md"]
    cfg.importanceWeightingEvalPointCount = P["n_eval"]
    cfg.importanceWeightingEvalPointGuassianWeight = P["min_weight"]
    cfg.importanceWeightingMeasurementLikelihoodMDThreshold = P["weighting_md"]
    cfg.gaussianMergingThreshold = P["merge_thr"]
    cfg.gaussianMergingCovarianceInflationFactor = P["merge_infl"]
    cfg.gaussianPruningThreshold = P["prune_thr"]
    cfg.useClusterProcess = P["use_cluster"]
    cfg.minUpdatesBeforeResample = P["min_updates"]
    cfg.minMeasurementsBeforeResample = P["min_measurements"]
    f.set_filter_config(cfg)
    if hasattr(f, "config"):
        f.config = cfg
    f.set_model_victoriapark(P["R"], P["Slb"], P["pd_table"], P["expected_clutter"], P["rmax"], P["rmin"], P["bmax"], P["bmin"], P["buffer_pd"])
    f.set_kf_config(P["kf_range"], P["kf_bearing"])
    f.set_lmk_process_noise(P["Q_lm"])
    f.set_laser_scan(scan)
