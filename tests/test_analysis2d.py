"""tools/analysis2d_sim.py (analysis2dSim, reference src/analysis2dSim.cpp) on hand-made log files with known answers."""
import importlib.util
import os

import numpy as np


def _tool():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("analysis2d_sim", os.path.join(root, "tools", "analysis2d_sim.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_ospa_and_cola_known_answers():
    a = _tool()
    gt = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 2.0]])
    assert a.ospa(gt, gt, 0.2, 1.0) == (0.0, 3)
    assert a.cola(gt, gt) == 0.0
    # one landmark missing: OSPA = c / n, COLA = 1 (one cardinality error)
    e, n = a.ospa(gt[:2], gt, 0.2, 1.0)
    assert n == 3 and abs(e - 0.2 / 3) < 1e-15 and abs(a.cola(gt[:2], gt) - 1.0) < 1e-12
    # a 5 cm offset on one landmark: COLA = 0.05 / 0.2; beyond the cutoff it saturates at 1
    est = gt.copy(); est[1, 0] += 0.05
    assert abs(a.cola(est, gt) - 0.25) < 1e-12
    est[1, 0] += 10
    assert abs(a.cola(est, gt) - 1.0) < 1e-12
    # the assignment is optimal, not greedy: swapping the estimates' order changes nothing
    assert abs(a.cola(est[::-1], gt) - 1.0) < 1e-12
    assert a.ospa(np.zeros((0, 2)), np.zeros((0, 2)), 0.2, 1.0) == (0.0, 0)


def test_log_files_round_trip(tmp_path):
    a = _tool()
    d = str(tmp_path) + "/"
    T = [0.0, 0.1, 0.2]
    with open(d + "gtPose.dat", "w") as f:
        for t in T:
            f.write("%f   %f   %f   %f\n" % (t, 10 * t, 0.0, 3.1))
    with open(d + "deadReckoning.dat", "w") as f:
        for t in T:
            f.write("%f   %f   %f   %f\n" % (t, 10 * t + t, 0.5 * t, -3.1))       # heading error wraps: -3.1 - 3.1 + 2 pi
    with open(d + "gtLandmark.dat", "w") as f:
        f.write("1.0   1.0   0.1\n2.0   2.0   0.2\n3.0   3.0   7.0\n")
    with open(d + "particlePose.dat", "w") as f:
        for t in T[1:]:
            f.write("%f   0   %f   %f   %f   %f\n" % (t, 10 * t + 0.3, 0.0, 3.1, 0.25))
            f.write("%f   1   %f   %f   %f   %f\n" % (t, 10 * t - 0.1, 0.0, 3.1, 0.75))   # the highest weight: its map is judged
            f.write("\n")
    with open(d + "landmarkEst.dat", "w") as f:
        f.write("0.1   1   1.02   1.0   0.01   0.0   0.01   0.9\n")
        f.write("0.1   1   5.0    5.0   0.01   0.0   0.01   0.3\n")                       # below 0.75: counts for the cardinality only
        f.write("0.1   0   9.0    9.0   0.01   0.0   0.01   0.9\n")                       # another particle's: ignored
        f.write("0.2   1   1.0    1.0   0.01   0.0   0.01   0.95\n")
    dr, pose, mp = a.analyse(d)
    assert dr.shape == (2, 5) and pose.shape == (2, 5) and mp.shape == (2, 4)
    np.testing.assert_allclose(dr[0], [0.1, 0.1, 0.05, -6.2 + 2 * np.pi, np.hypot(0.1, 0.05)], atol=1e-9)
    np.testing.assert_allclose(pose[0], [0.1, 0.25 * 0.3 - 0.75 * 0.1, 0.0, 0.0, 0.25 * 0.3 + 0.75 * 0.1], atol=1e-9)
    # t = 0.1: one landmark in range so far, estimate 2 cm off -> COLA 0.1; cardinality estimate 0.9 + 0.3
    np.testing.assert_allclose(mp[0], [0.1, 1, 1.2, 0.02 / 0.2], atol=1e-9)
    # t = 0.2: two landmarks in range, one estimated exactly -> one cardinality error
    np.testing.assert_allclose(mp[1], [0.2, 2, 0.95, 1.0], atol=1e-9)
    rows = open(d + "landmarkEstError.dat").read().split("\n")
    assert len(rows[0].split()) == 4 and len(open(d + "poseEstError.dat").readline().split()) == 5
