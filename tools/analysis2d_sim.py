#!/usr/bin/env python3
"""analysis2dSim (reference src/analysis2dSim.cpp) on the log files of host/rbphdslam2d_sim (or of the reference's own
rbphdslam2dSim -- same formats): per time step the dead-reckoning error, the weighted-mean pose error of the particle set
and the map error of the highest-weight particle.

    python tools/analysis2d_sim.py LOG_DIR/

reads   gtPose.dat `t x y th` | gtLandmark.dat `x y t_first_in_range` | particlePose.dat `t i x y th w` |
        landmarkEst.dat `t i mu_x mu_y S_xx S_xy S_yy w` | deadReckoning.dat `t x y th`
writes  deadReckoningError.dat `t ex ey er ed` | poseEstError.dat `t ex ey er ed` | landmarkEstError.dat `t nObservable cardEst cola`
        (src/analysis2dSim.cpp:392-422), and prints a summary line.

Map error: COLA (include/COLA.hpp:91-98) = OSPA * n^(1/p) / c with OSPA (include/OSPA.hpp:122-203: Euclidean distances cut
at c, padded with c to a square of n = max(n1, n2), optimal assignment, (sum C^p / n)^(1/p)), cutoff 0.20, order 1
(:232-233), between the estimated landmarks with weight >= 0.75 (:184) and the ground-truth landmarks that have been in
sensor range so far (:222-228).  The optimal assignment comes from scipy (the reference runs its Hungarian method; the
optimum is the same number).  This is a post-processing tool: it touches neither the device library nor the oracle."""
import os
import sys
import numpy as np
from scipy.optimize import linear_sum_assignment

W_THRESHOLD, CUTOFF, ORDER = 0.75, 0.20, 1.0


def ospa(est, truth, cutoff, order):
    n1, n2 = len(est), len(truth)
    n = max(n1, n2)
    if n == 0:
        return 0.0, 0
    Cm = np.full((n, n), float(cutoff))
    if n1 and n2:
        d = np.linalg.norm(np.asarray(est)[:, None, :] - np.asarray(truth)[None, :, :], axis=2)
        Cm[:n1, :n2] = np.minimum(d, cutoff)
    r, c = linear_sum_assignment(Cm)
    return float((np.sum(Cm[r, c] ** order) / n) ** (1.0 / order)), n


def cola(est, truth, cutoff=CUTOFF, order=ORDER):
    e, n = ospa(est, truth, cutoff, order)
    return e * n ** (1.0 / order) / cutoff


def wrap(a):
    return np.where(a > np.pi, a - 2 * np.pi, np.where(a < -np.pi, a + 2 * np.pi, a))


def analyse(log_dir, write=True):
    ld = lambda name: np.loadtxt(os.path.join(log_dir, name), ndmin=2)
    gt_pose, gt_lm, pp, le, dr = ld("gtPose.dat"), ld("gtLandmark.dat"), ld("particlePose.dat"), ld("landmarkEst.dat"), ld("deadReckoning.dat")
    dr_at = {round(float(t), 9): k for k, t in enumerate(dr[:, 0])}
    rows_dr, rows_pose, rows_map = [], [], []
    # particlePose.dat / landmarkEst.dat are grouped by time stamp
    pt = np.round(pp[:, 0], 9)
    lt = np.round(le[:, 0], 9) if le.size else np.zeros(0)
    for k in range(gt_pose.shape[0]):
        t, rx, ry, rz = gt_pose[k]
        key = round(float(t), 9)
        P = pp[pt == key]
        if P.shape[0] == 0:
            continue
        if key in dr_at:
            d = dr[dr_at[key]]
            ex, ey, er = d[1] - rx, d[2] - ry, float(wrap(d[3] - rz))
            rows_dr.append((t, ex, ey, er, np.hypot(ex, ey)))
        w = P[:, 5]
        ws = w.sum()
        ex, ey, er = P[:, 2] - rx, P[:, 3] - ry, wrap(P[:, 4] - rz)
        rows_pose.append((t, (ex * w).sum() / ws, (ey * w).sum() / ws, (er * w).sum() / ws, (np.hypot(ex, ey) * w).sum() / ws))
        i_hi = int(P[np.argmax(w), 1])                      # first particle with the highest weight (:163-166)
        L = le[(lt == key) & (le[:, 1] == i_hi)] if le.size else np.zeros((0, 8))
        est = L[L[:, 7] >= W_THRESHOLD][:, 2:4]
        card = float(L[:, 7].sum())
        seen = gt_lm[gt_lm[:, 2] <= t][:, :2]               # (:222-228; a landmark never in range carries -1 and so counts from the start, as in the reference)
        rows_map.append((t, seen.shape[0], card, cola(est, seen)))
    if write:
        for name, rows, fmt in (("deadReckoningError.dat", rows_dr, "%f   %f   %f   %f   %f\n"), ("poseEstError.dat", rows_pose, "%f   %f   %f   %f   %f\n"),
                                ("landmarkEstError.dat", rows_map, "%f   %d   %f   %f\n")):
            with open(os.path.join(log_dir, name), "w") as fh:
                for r in rows:
                    fh.write(fmt % r)
    return np.array(rows_dr), np.array(rows_pose), np.array(rows_map)


if __name__ == "__main__":
    if len(sys.argv) < 2:
        sys.exit("Usage: analysis2d_sim.py DATA_DIR/")
    d, p, m = analyse(sys.argv[1])
    print("steps %d: final dead-reckoning error %.3f m, final pose error %.3f m (mean %.3f m); landmarks in range so far %d, cardinality estimate %.2f, "
          "final COLA error %.2f (mean %.2f)" % (p.shape[0], d[-1, 4] if d.size else float("nan"), p[-1, 4], p[:, 4].mean(), int(m[-1, 1]), m[-1, 2], m[-1, 3], m[:, 3].mean()))
