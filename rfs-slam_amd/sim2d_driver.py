"""The 2-D simulator's host loop (config C1) over one or MORE filter handles in lock-step.

Mirrors the reference driver `src/rbphdslam2dSim.cpp`: the data generation (`generateTrajectory` :146-206, `generateOdometry`
:209-246, `generateLandmarks` :249-284, `generateMeasurements` :287-366, numpy's generator instead of drand48 / boost: a seed
does not name the same realisation as in the reference) and the filter loop `run()` :540-643 -- `predict` (births at the
pre-propagation pose, host propagation with additive process noise, static landmark step), `setParticlePose` to the ground truth
for k <= 100, the measurements of the time step, `update`, and the resample-or-normalise tail of RBPHDFilter::update
(include/RBPHDFilter.hpp:524-539) with ParticleFilter::resample's N_eff test and systematic plan.

Everything random (process noise, the resampling draw) is drawn ONCE per step by this loop and applied to every handle, and the
particle poses live here, so that the device engine and the CPU oracle can be driven through the same realisation and compared
after every call (tests/test_gpu_parity.py::test_c1_trajectory_device_vs_oracle).  The filters only have to offer the C-ABI
methods of capi.CFilter.
"""
import numpy as np

from .engine import systematic_resample_plan

# values of the shipped cfg/rbphdslam2dSim.xml (tests/golden/rbphdslam2dSim_c1.xml)
C1_SIM = dict(kmax=3000, dt=0.1, n_segments=20, max_dx=0.30, max_dy=0.0, max_dz=0.50, min_dx=0.10, vardx=0.002, vardy=0.002, vardz=0.002,
              n_landmarks=50, varlmx=2e-4, varlmy=2e-4, rmax=2.5, rmin=0.5, rbuf=0.05, Pd=0.99, clutter=1e-4, varzr=5e-4, varzb=5e-5,
              p_noise_inflation=1.5, z_noise_inflation=10.0, birth_w=0.01, kf_range=1.0, kf_bearing=0.2, new_gaussian_md=3.0,
              n_eval=15, min_weight=0.75, weighting_md=3.0, use_cluster=0, eff_n=100.0, min_updates=2, merge_thr=0.5, merge_infl=1.5,
              prune_thr=0.01)


def odometry_step(x, u):
    """MotionModel_Odometry2d::step (src/ProcessModel_Odometry2D.cpp:40-90), vectorised: x [N,3] or [3], u [3] or [N,3]."""
    x = np.asarray(x, dtype=np.float64)
    u = np.asarray(u, dtype=np.float64)
    th = x[..., 2]
    ct, st = np.cos(th), np.sin(th)
    out = np.empty(np.broadcast(x, u).shape)
    out[..., 0] = x[..., 0] + (ct * u[..., 0] - st * u[..., 1])       # p_k = p_km + C_km^T dp
    out[..., 1] = x[..., 1] + (st * u[..., 0] + ct * u[..., 1])
    cd, sd = np.cos(u[..., 2]), np.sin(u[..., 2])
    # C_k = C_d C_km with C = [c s; -s c];  theta_k = atan2(C_k(0,1), C_k(0,0))
    out[..., 2] = np.arctan2(cd * st + sd * ct, cd * ct - sd * st)
    return out


def generate(P=C1_SIM, traj_seed=1, kmax=None):
    """gt poses [K,3], odometry [K,3], landmarks [L,2], per-step measurement lists."""
    K = int(kmax or P["kmax"])
    K_full = int(P["kmax"])           # segment / landmark spacing follows the configured length even when fewer steps are run
    dt = P["dt"]
    rng = np.random.default_rng(traj_seed)
    Qd = np.array([P["vardx"], P["vardy"], P["vardz"]])
    gt = np.zeros((K, 3))
    disp = np.zeros((K, 3))
    seg = 0
    u = np.zeros(3)
    for k in range(1, K):
        if k <= 50:
            u = np.zeros(3)
        elif k >= K_full // P["n_segments"] * seg:
            seg += 1
            dx = rng.random() * P["max_dx"] * dt
            while dx < P["min_dx"] * dt:
                dx = rng.random() * P["max_dx"] * dt
            dy = (rng.random() * P["max_dy"] * 2 - P["max_dy"]) * dt
            dz = (rng.random() * P["max_dz"] * 2 - P["max_dz"]) * dt
            u = np.array([dx, dy, dz])
        disp[k] = u
        gt[k] = odometry_step(gt[k - 1], u)
    # landmarks: inverse measurement of a random (r, b) from the pose at regular intervals (:249-284)
    lm = []
    for k in range(1, K_full):
        if k >= K_full // P["n_landmarks"] * len(lm) and k < K:
            r, b = rng.random() * P["rmax"], rng.random() * 2 * np.pi
            x = gt[k]
            lm.append([x[0] + r * np.cos(x[2] + b), x[1] + r * np.sin(x[2] + b)])
    lm = np.array(lm).reshape(-1, 2)
    # odometry = displacement + N(0, Q dt^2) (:209-246)
    odom = np.zeros((K, 3))
    for k in range(1, K):
        odom[k] = disp[k] + np.sqrt(Qd) * dt * rng.standard_normal(3)
    # measurements (:287-366): noisy range-bearing of the landmarks in range with probability Pd + Poisson clutter
    mean_clutter = P["clutter"] * (2 * np.pi) * (P["rmax"] - P["rmin"])      # MeasurementModel_RngBrg::clutterIntensityIntegral
    sr, sb = np.sqrt(P["varzr"]), np.sqrt(P["varzb"])
    Z = [np.zeros((0, 2))]
    for k in range(1, K):
        x = gt[k]
        zs = []
        for m in range(len(lm)):
            dxm, dym = lm[m, 0] - x[0], lm[m, 1] - x[1]
            r = np.hypot(dxm, dym) + sr * rng.standard_normal()
            b = np.arctan2(dym, dxm) - x[2] + sb * rng.standard_normal()
            b = (b + np.pi) % (2 * np.pi) - np.pi
            if P["rmin"] <= r <= P["rmax"] and rng.random() <= P["Pd"]:
                zs.append([r, b])
        for _ in range(rng.poisson(mean_clutter)):
            r = rng.random() * P["rmax"]
            while r < P["rmin"]:
                r = rng.random() * P["rmax"]
            zs.append([r, rng.random() * 2 * np.pi - np.pi])
        Z.append(np.array(zs, dtype=np.float64).reshape(-1, 2))
    return dict(gt=gt, odom=odom, landmarks=lm, Z=Z, K=K)


def configure(f, P=C1_SIM):
    """setupRBPHDFilter (:444-492) through the C ABI."""
    dt = P["dt"]
    cfg = f.default_filter_config()
    cfg.birthGaussianWeight = P["birth_w"]
    cfg.minUpdatesBeforeResample = P["min_updates"]
    cfg.newGaussianCreateInnovMDThreshold = P["new_gaussian_md"]
    cfg.importanceWeightingMeasurementLikelihoodMDThreshold = P["weighting_md"]
    cfg.importanceWeightingEvalPointCount = P["n_eval"]
    cfg.importanceWeightingEvalPointGuassianWeight = P["min_weight"]
    cfg.gaussianMergingThreshold = P["merge_thr"]
    cfg.gaussianMergingCovarianceInflationFactor = P["merge_infl"]
    cfg.gaussianPruningThreshold = P["prune_thr"]
    cfg.useClusterProcess = P["use_cluster"]
    f.set_filter_config(cfg)
    f.set_model_rngbrg(np.diag([P["varzr"], P["varzb"]]) * P["z_noise_inflation"], P["Pd"], P["clutter"], P["rmax"], P["rmin"], P["rbuf"])
    f.set_kf_config(P["kf_range"], P["kf_bearing"])
    f.set_lmk_process_noise(np.diag([P["varlmx"], P["varlmy"]]) * dt * dt)
    return cfg


class Sim2dRun:
    """run() :540-643 over `filters` (all driven through the same realisation); `on_step(k, run)` is called after every update."""

    def __init__(self, filters, data, P=C1_SIM, seed=1, eff_n=None):
        self.filters = list(filters)
        self.n = self.filters[0].n
        assert all(f.n == self.n for f in self.filters)
        self.data, self.P = data, P
        self.rng = np.random.default_rng(seed)
        self.x = np.zeros((self.n, 3))
        self.cov = np.zeros((3, 3))          # pose covariance shared by all particles: 0 (ground-truth poses) or Q (after sample())
        self.Q = np.diag([P["vardx"], P["vardy"], P["vardz"]]) * P["p_noise_inflation"] * P["dt"] ** 2
        self.eff_n = float(P["eff_n"] if eff_n is None else eff_n)
        self.cfgs = [configure(f, P) for f in self.filters]
        self.n_updates_since = 0
        self.n_meas_since = 0
        self.n_resamples = 0
        self.n_updates = 0
        self.resample_steps = []
        self.z_of_step = None

    def _each(self, fn):
        return [fn(f) for f in self.filters]

    def step(self, k):
        P, d = self.P, self.data
        # predict (:588): births at the poses the last update used, then the static landmark step; host propagation
        self._each(lambda f: (f.set_poses(self.x, self.cov), f.predict_map(True)))
        noise = self.rng.standard_normal((self.n, 3)) * np.sqrt(np.diag(self.Q))     # s_k.setCov(Q); s_k.sample()  (ProcessModel.hpp:143-149)
        self.x = odometry_step(self.x, d["odom"][k]) + noise
        self.cov = self.Q.copy()
        if k <= 100:                                                                    # :590-593
            self.x = np.tile(d["gt"][k], (self.n, 1))
            self.cov = np.zeros((3, 3))
        Z = d["Z"][k]
        self.z_of_step = Z
        self.n_updates_since += 1
        if len(Z) == 0:                                                                 # RBPHDFilter.hpp:450-452
            return False
        self.n_meas_since += len(Z)
        self.n_updates += 1
        self._each(lambda f: (f.set_poses(self.x, self.cov), f.update(Z)))
        fired = False
        if self.n_updates_since >= P["min_updates"] and self.n_meas_since >= 1:        # (minMeasurementsBeforeResample_ = 1, :381)
            fired = self._resample()
        if fired:
            self.n_updates_since = self.n_meas_since = 0
        else:
            self._each(lambda f: f.normalize_weights(f.weight_sums()[0]))
        return fired

    def _resample(self):
        self._each(lambda f: f.normalize_weights(f.weight_sums()[0]))
        ws = self._each(lambda f: f.get_weights())
        w = ws[0]
        neff = 1.0 / float(np.sum(w * w))
        if neff > self.eff_n and neff / self.n > self.eff_n / self.n:
            return False
        u01 = float(self.rng.random())
        plans = [systematic_resample_plan(wi, u01) for wi in ws]
        for p in plans[1:]:
            if not np.array_equal(p, plans[0]):
                raise AssertionError("the handles' weights lead to different resampling plans")
        self._each(lambda f: f.resample_apply(plans[0]))
        self.x = self.x[plans[0]]
        self.n_resamples += 1
        self.resample_steps.append(self.n_updates)
        return True

    def run(self, k_from=1, k_to=None, on_step=None):
        for k in range(k_from, int(k_to or self.data["K"])):
            fired = self.step(k)
            if on_step is not None:
                on_step(k, self, fired)
        return self


def map_error(f, i, landmarks, w_min=0.5, cutoff=0.5):
    """Matched landmarks / mean error of particle i's strong Gaussians against the ground truth (greedy nearest, as the C++ driver's summary)."""
    w, _, mean, _ = f.export_gm(i)
    strong = mean[w >= w_min]
    taken = np.zeros(len(landmarks), dtype=bool)
    errs = []
    for m in strong:
        dist = np.linalg.norm(landmarks - m, axis=1)
        dist[taken] = np.inf
        j = int(np.argmin(dist)) if len(dist) else -1
        if j >= 0 and dist[j] < cutoff:
            taken[j] = True
            errs.append(dist[j])
    return int(taken.sum()), (float(np.mean(errs)) if errs else float("nan")), len(strong)
