"""Event-driven Victoria Park host loop over a filter handle (device engine in production).

Mirrors the reference driver's run() (src/rbphdslam_VictoriaPark.cpp:471-628): `Input` messages -> predict only;
`Lidar` messages -> predict, optional artificial clutter (:555-580), setLaserScan (:582), update (:583); the Ackerman
motion model (src/ProcessModel_Ackerman2D.cpp:47-78) and the input-noise sampling stay on the host like every RNG-bound
piece of the reference; resampling follows ParticleFilter::resample through engine.systematic_resample_plan.
The dataset's raw LASER.txt is missing from the reference, so the scan is the synthetic constant-70 m one.
"""
import numpy as np

from .engine import systematic_resample_plan

ACKERMAN = dict(h=0.76, l=2.83, dx=3.78, dy=0.50)   # cfg/rbphdslam_VictoriaPark*.xml <AckermanModel>


def ackerman_step(x, u_v, u_r, dt, a=ACKERMAN):
    """MotionModel_Ackerman2d::step, vectorised over particles (x: [N,3], u_v/u_r: [N])."""
    r = x[:, 2]
    c, s, t = np.cos(r), np.sin(r), np.tan(u_r)
    v = u_v / (1 - t * a["h"] / a["l"])
    out = x.copy()
    out[:, 0] += dt * (v * c - v / a["l"] * t * (a["dx"] * s + a["dy"] * c))
    out[:, 1] += dt * (v * s + v / a["l"] * t * (a["dx"] * c - a["dy"] * s))
    out[:, 2] += dt * v / a["l"] * t
    th = out[:, 2]
    out[:, 2] = np.where(th > np.pi, th - 2 * np.pi, np.where(th < -np.pi, th + 2 * np.pi, th))
    return out


class VictoriaParkRun:
    def __init__(self, f, data, params, seed=1, var_uv=0.2, var_ur=0.025, noise_inflation=20.0, added_clutter=3.0, eff_n=None,
                 filter="rbphd"):
        # filter = "rbphd": src/rbphdslam_VictoriaPark.cpp; "fastslam": src/fastslam_VictoriaPark.cpp (same event loop around
        # FastSLAM::predict / update: no births in predict, rfsgpu_fastslam_update)
        self.f, self.P = f, params
        self.fastslam = filter == "fastslam"
        self.mgr, self.inputs, self.meas = data["manager"], data["inputs"], data["measurements"]
        self.rng = np.random.default_rng(seed)
        self.n = f.n
        self.x = np.zeros((self.n, 3))
        self.var = np.array([var_uv, var_ur]) * noise_inflation
        self.added_clutter = added_clutter
        self.eff_n = eff_n if eff_n is not None else self.n / 2.0
        self.scan = np.full(361, 70.0)
        self.n_updates_since = 0
        self.n_meas_since = 0
        self.n_resamples = 0
        self.n_lidar = 0

    def _predict(self, u, dt, stationary, birth):
        f = self.f
        f.set_lmk_process_noise(np.diag([5e-4, 5e-4, 1e-4]) * dt * dt)   # varlm* x dt^2 (:497-500)
        f.set_poses(self.x, None)                                          # poses BEFORE propagation: birth uses them
        f.predict_map(False if self.fastslam else birth)
        if stationary:
            uv = np.full(self.n, u[0]); ur = np.full(self.n, u[1])
        else:                                                              # predict(u, dt, false, true): noise from the input
            uv = u[0] + np.sqrt(self.var[0]) * self.rng.standard_normal(self.n)
            ur = u[1] + np.sqrt(self.var[1]) * self.rng.standard_normal(self.n)
        self.x = ackerman_step(self.x, uv, ur, dt)

    def run(self, n_messages=None):
        f, P = self.f, self.P
        t_km, u_km, stationary, birth, z_idx = 0.0, np.zeros(2), True, True, 0
        msgs = self.mgr if n_messages is None else self.mgr[:n_messages]
        for t_k, typ, idx in msgs:
            idx = int(idx) - 1
            dt = t_k - t_km
            if typ == 2:          # Input
                self._predict(u_km, dt, stationary, birth)
                birth = False
                u_km = self.inputs[idx, 1:3].copy()
                if u_km[0] != 0:
                    stationary = False
                t_km = t_k
            elif typ == 3:        # Lidar
                self._predict(u_km, dt, stationary, birth)
                birth = False
                Z = []
                while z_idx < len(self.meas) and abs(self.meas[z_idx, 0] - t_k) < 1e-9:
                    Z.append(self.meas[z_idx, 1:4])
                    z_idx += 1
                if self.added_clutter > 0:
                    for _ in range(self.rng.poisson(self.added_clutter)):
                        r = self.rng.uniform() * (P["rmax"] - P["rmin"]) + P["rmin"]
                        b = self.rng.uniform() * ((np.rad2deg(P["bmax"]) - np.rad2deg(P["bmin"])) + np.rad2deg(P["bmin"])) * np.pi / 180  # sic (:563)
                        Z.append(np.array([r, b, 1.0]))
                Z = np.array(Z).reshape(-1, 3)[:60]
                f.set_laser_scan(self.scan)
                f.set_poses(self.x, None)
                self.n_updates_since += 1
                if len(Z):
                    self.n_meas_since += len(Z)
                    if self.fastslam:
                        f.fastslam_update(Z)
                    else:
                        f.update(Z)
                    self.n_lidar += 1
                    fired = False
                    if self.n_updates_since >= P["min_updates"] and self.n_meas_since >= P["min_measurements"]:
                        fired = self._resample()
                    if fired:
                        self.n_updates_since = self.n_meas_since = 0
                    else:
                        s = f.weight_sums()
                        f.normalize_weights(s[0])
                birth = True
                t_km = t_k
        return self

    def _resample(self):
        f = self.f
        s = f.weight_sums()
        f.normalize_weights(s[0])
        w = f.get_weights()
        neff = 1.0 / float(np.sum(w * w))
        if neff > self.eff_n and neff / self.n > self.eff_n / self.n:
            return False
        plan = systematic_resample_plan(w, float(self.rng.uniform()))
        f.resample_apply(plan)
        self.x = self.x[plan]
        self.n_resamples += 1
        return True
