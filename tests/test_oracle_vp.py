"""Victoria Park model in the oracle, cross-checked against independent numpy formulations of
src/MeasurementModel_VictoriaPark.cpp (measure / probabilityOfDetection(2) / setLaserScan) and of the 3-D KF map
update; plus the birth-candidate list logic (RBPHDFilter.hpp:1000-1084) against a Python restatement.  CPU only."""
import math

import numpy as np
import pytest


def wrap(a):
    return (a + np.pi) % (2 * np.pi) - np.pi


def np_vp_measure(P, pose, mu, Sig):
    th = pose[2] - np.pi / 2
    d = mu[:2] - pose[:2]
    r2 = d @ d
    r = np.sqrt(r2)
    H2 = np.array([[d[0] / r, d[1] / r], [-d[1] / r2, d[0] / r2]])
    R = np.asarray(P["R"])
    S = np.zeros((3, 3))
    S[:2, :2] = H2 @ Sig[:2, :2] @ H2.T + R[:2, :2]
    S[2, 2] = Sig[2, 2] + R[2, 2] + r * r * P["Slb"]
    H = np.zeros((3, 3))
    H[:2, :2] = H2
    H[2, 2] = 1
    return np.array([r, wrap(np.arctan2(d[1], d[0]) - th), mu[2]]), S, H


def np_vp_pd2(P, scan, pose, mu, Sig):
    z, _, _ = np_vp_measure(P, pose, mu, Sig)
    close = False
    if z[1] > P["bmax"] or z[1] < P["bmin"] or z[0] < P["rmin"] or z[0] > P["rmax"]:
        return 0.0, close
    rad = z[2] / 2
    gamma = math.atan(rad / z[0])
    tab = P["pd_table"]
    maxp = int(math.floor(2 * gamma * 720.0 / (2 * np.pi)))
    if len(tab) > maxp and tab[maxp] == 0:
        return 0.0, close
    if len(tab) > maxp and tab[maxp] < P["buffer_pd"]:
        close = True
    minb = int(math.ceil((z[1] - gamma) * 720.0 / (2 * np.pi))) % 720
    cnt = 0
    for k in range(maxp):
        s = scan[(minb + k) % 720] if (minb + k) % 720 < len(scan) else 0.0
        if s > z[0] - rad - 0.18 or s == 0:
            cnt += 1
    cnt = min(cnt, len(tab) - 1)
    if tab[cnt] == 0:
        close = False
    return tab[cnt], close


def np_vp_pd(P, scan, pose, mu, Sig):
    z, _, _ = np_vp_measure(P, pose, mu, Sig)
    ang = math.atan2(z[1], z[0]) + pose[2]
    perp = np.array([-math.sin(ang), math.cos(ang)])
    sd = max(3 * math.sqrt(perp @ Sig[:2, :2] @ perp), 0.2)
    vals = []
    close = False
    i = 1
    while (i - 1) * (2 * mu[2]) < sd:
        for sgn in (1, -1):
            m2 = mu.copy()
            m2[:2] = mu[:2] + sgn * i * 2 * mu[2] * perp
            v, close = np_vp_pd2(P, scan, pose, m2, Sig)
            vals.append(v)
        i += 1
    v, close = np_vp_pd2(P, scan, pose, mu, Sig)
    vals.append(v)
    if min(vals) == 0 and max(vals) > 0:
        close = True
    return max(vals), close


@pytest.mark.parametrize("scan", ["const", "ragged"])
def test_vp_measure_and_pd_vs_numpy(ob, sc, pkg, scan):
    scen = sc.make_vp_scenario(3, 40, 8, seed=7, scan=scan)
    o = ob.OracleFilter(3, model=pkg.capi.MODEL_VICTORIAPARK_3D)
    sc.load_scenario(o, scen)
    P = scen["params"]
    seen = set()
    for i in range(3):
        for m in range(40):
            mu, Sg = scen["mean"][i, m], scen["cov"][i, m]
            z, S, H = ob.vp_measure(o, scen["poses"][i], mu, Sg)
            z2, S2, H2 = np_vp_measure(P, scen["poses"][i], mu, Sg)
            np.testing.assert_allclose(z, z2, rtol=1e-12, atol=1e-13)
            np.testing.assert_allclose(S, S2, rtol=1e-11, atol=1e-15)
            np.testing.assert_allclose(H, H2, rtol=1e-12, atol=1e-15)
            pd, close = ob.vp_pd(o, scen["poses"][i], mu, Sg)
            pd2, close2 = np_vp_pd(P, scen["scan"], scen["poses"][i], mu, Sg)
            assert pd == pd2 and close == close2, (i, m, pd, pd2, close, close2)
            seen.add(pd)
    assert len(seen) >= 2            # in-FOV and out-of-FOV landmarks both occur
    area = (np.sum(scen["scan"][1:] * scen["scan"][:-1]) + scen["scan"][0] * scen["scan"][-1]) * math.sin(np.pi / 360) / 2
    assert np.isclose(ob.vp_clutter(o), P["expected_clutter"] / area, rtol=1e-13)


def np_vp_update_map(P, scan, pose, w, mu, Sig, Z):
    nM, nZ = len(w), len(Z)
    W = np.zeros((nM, nZ))
    new = {}
    Pd = np.zeros(nM)
    close = np.zeros(nM, bool)
    area = (np.sum(scan[1:] * scan[:-1]) + scan[0] * scan[-1]) * math.sin(np.pi / 360) / 2
    clutter = P["expected_clutter"] / area
    for m in range(nM):
        Pd[m], close[m] = np_vp_pd(P, scan, pose, mu[m], Sig[m])
        if close[m]:
            Pd[m] = 1.0
        if Pd[m] == 0:
            continue
        zexp, S, H = np_vp_measure(P, pose, mu[m], Sig[m])
        Si = np.linalg.inv(S)
        K = Sig[m] @ H.T @ Si
        Pn = (np.eye(3) - K @ H) @ Sig[m]
        Pn = (Pn + Pn.T) / 2
        for z in range(nZ):
            e = Z[z] - zexp
            nu = np.array([e[0], wrap(e[1]), e[2]])
            if abs(nu[0]) > P["kf_range"] or abs(nu[1]) > P["kf_bearing"]:
                continue
            md2 = e @ Si @ e
            if md2 > P["new_gaussian_md"] ** 2:
                continue
            lik = np.exp(-0.5 * md2) / np.sqrt((2 * np.pi) ** 3 * np.linalg.det(S))
            if lik == 0:
                continue
            W[m, z] = Pd[m] * w[m] * lik
            new[(m, z)] = (mu[m] + K @ nu, Pn)
    Wn = W / (clutter + W.sum(0))
    ow, omu, oS = [], [], []
    for (m, z), (x, Pn) in sorted(new.items()):
        if Wn[m, z] > 0:
            ow.append(Wn[m, z]); omu.append(x); oS.append(Pn)
    wk = (1 - Pd) * w
    for m in range(nM):
        if close[m] and w[m] > P["birth_w"]:
            dw = Pd[m] * w[m] - Wn[m].sum()
            if dw > 0:
                wk[m] = min(wk[m] + dw, 1.0)
    unused = [z for z in range(nZ) if not np.any(Wn[:, z] != 0)]
    return np.concatenate([wk, ow]), np.array(list(mu) + omu), np.array(list(Sig) + oS), unused, int((Pd != 0).sum())


@pytest.mark.parametrize("scan", ["const", "ragged"])
def test_vp_update_map_vs_numpy(ob, sc, pkg, scan):
    scen = sc.make_vp_scenario(3, 30, 10, seed=11, scan=scan)
    o = ob.OracleFilter(3, model=pkg.capi.MODEL_VICTORIAPARK_3D)
    sc.load_scenario(o, scen)
    o.update_map(scen["Z"])
    created = 0
    for i in range(3):
        w, mu, Sg, unused, nfov = np_vp_update_map(scen["params"], scen["scan"], scen["poses"][i], scen["w"][i], scen["mean"][i], scen["cov"][i], scen["Z"])
        ow, _, omu, oS = o.export_gm(i)
        assert len(ow) == len(w)
        created += len(w) - 30
        np.testing.assert_allclose(ow, w, rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(omu, mu, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(oS, Sg, rtol=1e-7, atol=1e-12)
        assert list(o.get_unused(i)) == unused
        assert o.landmarks_in_fov(i) == nfov
    assert created > 0


def test_vp_full_update_runs_and_is_finite(ob, sc, pkg):
    scen = sc.make_vp_scenario(4, 45, 12, seed=13)
    o = ob.OracleFilter(4, model=pkg.capi.MODEL_VICTORIAPARK_3D)
    sc.load_scenario(o, scen)
    o.update(scen["Z"])
    w = o.get_weights()
    assert np.all(np.isfinite(w)) and np.all(w > 0)
    assert np.all(o.gm_sizes() > 0)


def py_birth(P, cands, unused, nfov, Z, measure_md2, correct, inverse):
    """Python restatement of addBirthGaussians (:1000-1084) incl. the ++end() wrap of the promotion loop."""
    births = []
    for zi in reversed(unused):
        z = Z[zi]
        hit = False
        for c in cands:
            if measure_md2(c, z) <= P["birth_support_dist"] ** 2:
                correct(c, z)
                c["sup"] += 1
                hit = True
                break
        if not hit:
            c = inverse(z)
            c.update(sup=1, chk=0)
            if P["birth_count_thr"] == 1 or nfov <= P["birth_cur_thr"]:
                births.append(c)
            else:
                cands.append(c)
    k = 0
    while k < len(cands):
        cands[k]["chk"] += 1
        at_end = False
        while cands[k]["sup"] >= P["birth_count_thr"] or cands[k]["chk"] > P["birth_check_thr"] or nfov <= P["birth_cur_thr"]:
            if cands[k]["sup"] >= P["birth_count_thr"] or nfov <= P["birth_cur_thr"]:
                births.append(cands[k])
            del cands[k]
            if k < len(cands):
                cands[k]["chk"] += 1
            else:
                at_end = True
                break
        k = 0 if at_end else k + 1
        if at_end and not cands:
            break
    return births


def test_birth_candidate_list_vs_python(ob, sc, pkg):
    """Several predict cycles with the Victoria Park birth thresholds (5 supporting measurements / 10 checks / dist 2):
    candidate counts, support/check counters and promoted births must follow the Python restatement."""
    scen = sc.make_vp_scenario(2, 6, 6, seed=17, params=dict(birth_count_thr=3, birth_check_thr=2))
    P = scen["params"]
    o = ob.OracleFilter(2, model=pkg.capi.MODEL_VICTORIAPARK_3D)
    sc.load_scenario(o, scen)
    rng = np.random.default_rng(5)
    cands = [[], []]
    for step in range(7):
        Z = scen["Z"] + rng.normal(0, 1, scen["Z"].shape) * np.array([0.05, 0.002, 0.01])
        o.update(Z)
        before = [o.export_gm(i) for i in range(2)]
        unused = [list(o.get_unused(i)) for i in range(2)]
        nfov = [o.landmarks_in_fov(i) for i in range(2)]
        o.predict_map(True)
        for i in range(2):
            pose = scen["poses"][i]

            def md2(c, z, pose=pose):
                zexp, S, _ = np_vp_measure(P, pose, c["x"], c["S"])
                e = z - zexp
                return e @ np.linalg.inv(S) @ e

            def correct(c, z, pose=pose):
                zexp, S, H = np_vp_measure(P, pose, c["x"], c["S"])
                e = z - zexp
                nu = np.array([e[0], wrap(e[1]), e[2]])
                if abs(nu[0]) > P["kf_range"] or abs(nu[1]) > P["kf_bearing"]:
                    return
                K = c["S"] @ H.T @ np.linalg.inv(S)
                Pn = (np.eye(3) - K @ H) @ c["S"]
                c["x"] = c["x"] + K @ nu
                c["S"] = (Pn + Pn.T) / 2

            def inverse(z, pose=pose):
                a = pose[2] - np.pi / 2 + z[1]
                x = np.array([pose[0] + z[0] * np.cos(a), pose[1] + z[0] * np.sin(a), z[2]])
                Hi = np.array([[np.cos(a), -z[0] * np.sin(a)], [np.sin(a), z[0] * np.cos(a)]])
                S = np.zeros((3, 3))
                S[:2, :2] = Hi @ np.asarray(P["R"])[:2, :2] @ Hi.T
                S[2, 2] = np.asarray(P["R"])[2, 2]
                return dict(x=x, S=S)

            births = py_birth(P, cands[i], unused[i], nfov[i], Z, md2, correct, inverse)
            mean, cov, sup, chk = o.export_birth_candidates(i)
            assert len(sup) == len(cands[i]), (step, i)
            assert list(sup) == [c["sup"] for c in cands[i]] and list(chk) == [c["chk"] for c in cands[i]]
            for k, c in enumerate(cands[i]):
                np.testing.assert_allclose(mean[k], c["x"], rtol=1e-9, atol=1e-10)
                np.testing.assert_allclose(cov[k], c["S"], rtol=1e-7, atol=1e-12)
            w, _, mu, Sg = o.export_gm(i)
            assert len(w) == len(before[i][0]) + len(births)
            for k, c in enumerate(births):
                j = len(before[i][0]) + k
                assert w[j] == P["birth_w"]
                np.testing.assert_allclose(mu[j], c["x"], rtol=1e-9, atol=1e-10)
                np.testing.assert_allclose(Sg[j], c["S"] + np.asarray(P["Q_lm"]), rtol=1e-7, atol=1e-12)
    assert any(len(c) for c in cands) or step > 0
