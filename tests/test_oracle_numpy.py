"""Cross-checks of the oracle's UNPINNED rows (updateMap, KF correct, RngBrg model, importanceWeighting, merge,
prune) against independent numpy/scipy formulations written from the equations, not from the oracle's code
(np.linalg for the algebra, scipy.stats for Gaussian densities, scipy.sparse.csgraph for components, itertools for
assignment sums).  CPU only.  These do not pin the oracle to the reference (no reference goldens can be produced in
this image) -- they guard against transcription errors in the restatement."""
import itertools

import numpy as np
import pytest
from scipy.sparse import csr_matrix
from scipy.sparse.csgraph import connected_components
from scipy.stats import multivariate_normal

DENORM = np.nextafter(0, 1)


def wrap(a):
    return (a + np.pi) % (2 * np.pi) - np.pi


def np_measure(P, pose, pose_cov, mu, Sig):
    d = mu - pose[:2]
    r2 = d @ d
    r = np.sqrt(r2)
    zexp = np.array([r, wrap(np.arctan2(d[1], d[0]) - pose[2])])
    H = np.array([[d[0] / r, d[1] / r], [-d[1] / r2, d[0] / r2]])
    Hr = np.array([[-d[0] / r, -d[1] / r, 0], [d[1] / r2, -d[0] / r2, -1]])
    S = H @ Sig @ H.T + Hr @ pose_cov @ Hr.T + np.asarray(P["R"])
    return zexp, H, S, (P["rmin"] <= r <= P["rmax"]), r


def np_pd(P, r):
    if P["rmin"] <= r <= P["rmax"]:
        return P["Pd"], (r >= P["rmax"] - P["rbuf"] or r <= P["rmin"] + P["rbuf"])
    return 0.0, (P["rmin"] - P["rbuf"] <= r <= P["rmax"] + P["rbuf"])


def np_update_map(P, pose, pose_cov, w, mu, Sig, Z):
    nM, nZ = len(w), len(Z)
    W = np.zeros((nM, nZ))
    new = {}
    Pd = np.zeros(nM)
    close = np.zeros(nM, bool)
    for m in range(nM):
        zexp, H, S, ok, r = np_measure(P, pose, pose_cov, mu[m], Sig[m])
        Pd[m], close[m] = np_pd(P, r)
        if close[m]:
            Pd[m] = 1.0
        if Pd[m] == 0 or not ok:
            continue
        Si = np.linalg.inv(S)
        K = Sig[m] @ H.T @ Si
        Pn = (np.eye(2) - K @ H) @ Sig[m]
        Pn = (Pn + Pn.T) / 2
        for z in range(nZ):
            e = Z[z] - zexp
            if P["kf_range"] > 0 and abs(e[0]) > P["kf_range"]:
                continue
            nu = np.array([e[0], wrap(e[1])])
            if P["kf_bearing"] > 0 and abs(nu[1]) > P["kf_bearing"]:
                continue
            md2 = e @ Si @ e                                  # raw difference on purpose
            if md2 > P["new_gaussian_md"] ** 2:
                continue
            lik = np.exp(-0.5 * md2) / np.sqrt((2 * np.pi) ** 2 * np.linalg.det(S))
            if lik == 0:
                continue
            W[m, z] = Pd[m] * w[m] * lik
            new[(m, z)] = (mu[m] + K @ nu, Pn)
    colsum = P["clutter"] + W.sum(0)
    Wn = W / colsum
    out_w, out_mu, out_S = [], [], []
    for (m, z), (x, Pn) in sorted(new.items()):
        if Wn[m, z] > 0:
            out_w.append(Wn[m, z]); out_mu.append(x); out_S.append(Pn)
    wk = (1 - Pd) * w
    for m in range(nM):
        if close[m] and w[m] > P["birth_w"]:
            dw = Pd[m] * w[m] - Wn[m].sum()
            if dw > 0:
                wk[m] = min(wk[m] + dw, 1.0)
    unused = [z for z in range(nZ) if not np.any(Wn[:, z] != 0)]
    return (np.concatenate([wk, out_w]), np.concatenate([w, np.zeros(len(out_w))]),
            np.array(list(mu) + out_mu), np.array(list(Sig) + out_S), unused, int((Pd != 0).sum()), colsum)


@pytest.mark.parametrize("seed,kw", [(1, {}), (2, dict(frac_in_fov=0.5)), (3, dict(n_landmarks=70, n_z=20))])
def test_update_map_vs_numpy(ob, sc, seed, kw):
    args = dict(n_particles=3, n_landmarks=25, n_z=8, seed=seed, per_particle_pose_cov=True)
    args.update(kw)
    scen = sc.make_scenario(**args)
    o = ob.OracleFilter(scen["n"])
    sc.load_scenario(o, scen)
    o.update_map(scen["Z"])
    P = scen["params"]
    for i in range(scen["n"]):
        w, wp, mu, Sg, unused, nfov, _ = np_update_map(P, scen["poses"][i], scen["pose_cov"][i], scen["w"][i], scen["mean"][i], scen["cov"][i], scen["Z"])
        ow, owp, omu, oS = o.export_gm(i)
        assert len(ow) == len(w)
        np.testing.assert_allclose(ow, w, rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(owp, wp, rtol=0, atol=0)
        np.testing.assert_allclose(omu, mu, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(oS, Sg, rtol=1e-8, atol=1e-14)
        assert list(o.get_unused(i)) == unused
        assert o.landmarks_in_fov(i) == nfov


def np_partition_likelihood(L, pd, c):
    """Per connected component, with the reference's semantics: isolated rows/cols are pooled into
    ONE zero partition that contributes prod(Pd)*prod(c) -- and, as the caller indexes partitions, merged-away
    singletons are visited again as ordinary partitions while the same number of trailing components is dropped."""
    nE, nZ = L.shape
    V = nE + nZ
    rows, cols = np.nonzero(L)
    g = csr_matrix((np.ones(len(rows)), (rows, cols + nE)), shape=(V, V))
    _, lab = connected_components(g, directed=False)
    order = []                                   # component ids in order of their smallest vertex
    for v in range(V):
        if lab[v] not in order:
            order.append(lab[v])
    comps = [([e for e in range(nE) if lab[e] == cid], [n for n in range(nZ) if lab[nE + n] == cid]) for cid in order]
    zero = [k for k, (r, cc) in enumerate(comps) if not r or not cc]
    combined = zero[0] if zero else None
    nP = len(comps) - max(len(zero) - 1, 0)
    total = 1.0
    for p in range(nP):
        r, cc = comps[p]
        if p == combined:
            zr = [e for k in zero for e in comps[k][0]]
            zc = [n for k in zero for n in comps[k][1]]
            total *= np.prod([pd[e] for e in zr]) * c ** len(zc)
            continue
        s = 0.0
        for k in range(0, min(len(r), len(cc)) + 1):
            for rs in itertools.combinations(range(len(r)), k):
                for cs in itertools.permutations(range(len(cc)), k):
                    t = 1.0
                    for a, b in zip(rs, cs):
                        t *= L[r[a], cc[b]]
                    for a in range(len(r)):
                        if a not in rs:
                            t *= 1 - pd[r[a]]
                    s += t * c ** (len(cc) - k)
        total *= s
    return total


def np_importance_weight(P, pose, pose_cov, w, wp, mu, Sg, Z, prev_weight):
    order = np.argsort(-w, kind="stable")
    w, wp, mu, Sg = w[order], wp[order], mu[order], Sg[order]
    ev, evpd = [], []
    for m in range(len(w)):
        if w[m] < P["min_weight"]:
            break
        pdm, _ = np_pd(P, np.linalg.norm(mu[m] - pose[:2]))
        if pdm > 0:
            ev.append(m); evpd.append(pdm)
        if len(ev) >= min(P["n_eval"], len(w)):
            break
    pb = pa = 1.0
    for e in ev:
        dens = np.array([multivariate_normal.pdf(mu[e], mean=mu[m], cov=Sg[m]) for m in range(len(w))])
        pb *= DENORM + np.sum(wp * dens)
        pa *= DENORM + np.sum(w * dens)
    L = np.zeros((len(ev), len(Z)))
    for k, e in enumerate(ev):
        zexp, _, S, _, _ = np_measure(P, pose, pose_cov, mu[e], np.zeros((2, 2)))
        Si = np.linalg.inv(S)
        for n in range(len(Z)):
            d = Z[n] - zexp
            md2 = d @ Si @ d
            L[k, n] = 0.0 if md2 > P["weighting_md"] ** 2 else evpd[k] * np.exp(-0.5 * md2) / np.sqrt((2 * np.pi) ** 2 * np.linalg.det(S))
    ml = np_partition_likelihood(L, evpd, P["clutter"]) / (P["clutter"] * 2 * np.pi * (P["rmax"] - P["rmin"]))
    return ml * pb / pa * np.exp(w.sum() - wp.sum()) * prev_weight


@pytest.mark.parametrize("seed", [4, 5, 6])
def test_importance_weighting_vs_numpy(ob, sc, seed):
    scen = sc.make_scenario(3, 30, 9, seed=seed, per_particle_pose_cov=True, n_eval=6)
    o = ob.OracleFilter(scen["n"])
    sc.load_scenario(o, scen)
    o.update_map(scen["Z"])
    maps = [o.export_gm(i) for i in range(scen["n"])]
    o.importance_weighting()
    assert o.murty_calls() == 0
    got = o.get_weights()
    for i in range(scen["n"]):
        w, wp, mu, Sg = maps[i]
        want = np_importance_weight(scen["params"], scen["poses"][i], scen["pose_cov"][i], w, wp, mu, Sg, scen["Z"], 1.0)
        assert np.isclose(got[i], want, rtol=1e-8), (got[i], want)


def np_merge(w, mu, Sg, t, f):
    w, mu, Sg = w.copy(), mu.copy(), Sg.copy()
    alive = np.ones(len(w), bool)
    for a in range(len(w)):
        if not alive[a]:
            continue
        for b in range(a + 1, len(w)):
            if not alive[b]:
                continue
            e = mu[b] - mu[a]
            if e @ np.linalg.solve(Sg[a], e) > t * t and e @ np.linalg.solve(Sg[b], e) > t * t:
                continue
            wm = w[a] + w[b]
            if wm == 0:
                continue
            xm = (mu[a] * w[a] + mu[b] * w[b]) / wm
            d1, d2 = xm - mu[a], xm - mu[b]
            Sg[a] = (w[a] * (Sg[a] + f * np.outer(d1, d1)) + w[b] * (Sg[b] + f * np.outer(d2, d2))) / wm
            mu[a], w[a], alive[b] = xm, wm, False
    return w[alive], mu[alive], Sg[alive]


@pytest.mark.parametrize("seed", [7, 8])
def test_merge_and_prune_vs_numpy(ob, sc, seed):
    scen = sc.make_scenario(3, 40, 12, seed=seed)
    o = ob.OracleFilter(scen["n"])
    sc.load_scenario(o, scen)
    o.update_map(scen["Z"])
    o.importance_weighting()
    before = [o.export_gm(i) for i in range(scen["n"])]
    o.merge()
    P = scen["params"]
    merged_any = False
    for i in range(scen["n"]):
        w, _, mu, Sg = before[i]
        mw, mmu, mS = np_merge(w, mu, Sg, P["merge_thr"], P["merge_infl"])
        merged_any |= len(mw) < len(w)
        ow, _, omu, oS = o.export_gm(i)
        assert len(ow) == len(mw)
        np.testing.assert_allclose(ow, mw, rtol=1e-12)
        np.testing.assert_allclose(omu, mmu, rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(oS, mS, rtol=1e-9, atol=1e-15)
    assert merged_any
    after_merge = [o.export_gm(i) for i in range(scen["n"])]
    o.prune()
    for i in range(scen["n"]):
        w, _, mu, Sg = after_merge[i]
        keep = w >= P["prune_thr"]
        order = np.argsort(-w[keep], kind="stable")
        ow, _, omu, oS = o.export_gm(i)
        np.testing.assert_array_equal(ow, w[keep][order])
        np.testing.assert_array_equal(omu, mu[keep][order])


def test_cluster_process_weight_vs_numpy(ob, sc):
    scen = sc.make_scenario(2, 20, 6, seed=9, use_cluster=True)
    o = ob.OracleFilter(scen["n"])
    sc.load_scenario(o, scen)
    o.update_map(scen["Z"])
    for i in range(scen["n"]):
        *_, colsum = np_update_map(scen["params"], scen["poses"][i], scen["pose_cov"], scen["w"][i], scen["mean"][i], scen["cov"][i], scen["Z"])
        want = np.exp(DENORM + scen["w"][i].sum()) * np.prod(colsum)
        assert np.isclose(o.get_weights()[i], want, rtol=1e-10)


def test_resample_decision_properties(ob):
    rng = np.random.default_rng(3)
    w = rng.uniform(0, 1, 50) ** 8
    fired, wn, src = ob.resample_decide(w, 25.0, 0.3)
    assert abs(wn.sum() - 1) < 1e-12
    neff = 1 / np.sum(wn ** 2)
    assert fired == (not (neff > 25.0))
    if fired:
        counts = np.bincount(src, minlength=50)
        # systematic resampling: offspring counts within 1 of N*w
        assert np.all(np.abs(counts - 50 * wn) < 1 + 1e-9)
    fired2, _, _ = ob.resample_decide(np.ones(50), 25.0, 0.3)
    assert not fired2


# ---- FastSLAM 1.0 update (include/FastSLAM.hpp:424-706), written from the equations with scipy's assignment solver ------

def np_fastslam_update(P, F, pose, pose_cov, w_particle, lw, mu, Sig, Z):
    """One particle, no candidate queue (count threshold 1): returns (new particle weight, log-odds, means, covariances)
    before pruning order is applied (sorted by log-odds descending, entries below the prune threshold dropped)."""
    from scipy.optimize import linear_sum_assignment
    nZ = len(Z)
    lim = F["minLog"]
    rows = []
    for m in range(len(lw)):
        zexp, H, S, ok, r = np_measure(P, pose, pose_cov, mu[m], Sig[m])
        pd, close = np_pd(P, r)
        if pd != 0 or close:
            rows.append((m, pd, zexp, H, S, ok))
    nM = len(rows)
    n = max(nM, nZ)
    T = np.full((n, n), lim)
    for k, (m, pd, zexp, H, S, ok) in enumerate(rows):
        if ok:
            for z in range(nZ):
                T[k, z] = max(lim, multivariate_normal(zexp, S).logpdf(Z[z]))      # raw difference, like the reference
    r_idx, c_idx = linear_sum_assignment(T, maximize=True)
    da = dict(zip(r_idx, c_idx))
    pfa = P["clutter"] * 2 * np.pi * (P["rmax"] - P["rmin"]) / nZ
    prior = F["prior"]
    lw, mu, Sig = lw.copy(), mu.copy(), Sig.copy()
    used = np.zeros(nZ, bool)
    logw = 0.0
    for k, (m, pd, zexp, H, S, ok) in enumerate(rows):
        z = da.get(k, -1)
        upd = False
        if 0 <= z < nZ and T[k, z] > lim:
            e = Z[z] - zexp
            nu = np.array([e[0], wrap(e[1])])
            if not ((P["kf_range"] > 0 and abs(nu[0]) > P["kf_range"]) or (P["kf_bearing"] > 0 and abs(nu[1]) > P["kf_bearing"])):
                K = Sig[m] @ H.T @ np.linalg.inv(S)
                Pn = (np.eye(2) - K @ H) @ Sig[m]
                mu[m] = mu[m] + K @ nu
                Sig[m] = (Pn + Pn.T) / 2
                upd = True
        if upd:
            used[z] = True
            logw += T[k, z]
            pe = ((1 - pd) * pfa * prior + pd * prior) / (pfa + (1 - pfa) * pd * prior)
        else:
            pe = ((1 - pd) * prior) / ((1 - prior) + (1 - pd) * prior)
            if lw[m] > F["lock"]:
                pe = 0.5
        lw[m] = lw[m] + np.log(pe / (1 - pe))
    keep = lw >= F["prune"]
    lw, mu, Sig = lw[keep], mu[keep], Sig[keep]
    order = np.argsort(-lw, kind="stable")
    lw, mu, Sig = lw[order], mu[order], Sig[order]
    new_w = np.log(prior / (1 - prior))
    for z in range(nZ):
        if not used[z]:
            a = pose[2] + Z[z][1]
            x = pose[:2] + Z[z][0] * np.array([np.cos(a), np.sin(a)])
            Hi = np.array([[np.cos(a), -Z[z][0] * np.sin(a)], [np.sin(a), Z[z][0] * np.cos(a)]])
            lw = np.append(lw, new_w)
            mu = np.vstack([mu, x]) if len(mu) else x[None]
            Sig = np.concatenate([Sig, (Hi @ np.asarray(P["R"]) @ Hi.T)[None]]) if len(Sig) else (Hi @ np.asarray(P["R"]) @ Hi.T)[None]
    return w_particle * np.exp(logw), lw, mu, Sig


@pytest.mark.parametrize("seed,nm,nz,rmax", [(3, 12, 6, 5.0), (4, 40, 14, 10.0), (5, 5, 12, 5.0)])
def test_fastslam_update_against_numpy(ob, sc, seed, nm, nz, rmax):
    """The oracle's FastSLAM restatement (CostMatrix::reduce + Murty's first = Hungarian optimum on the reduced table) against
    a direct formulation: scipy's assignment on the FULL table, numpy KF / log-odds algebra, immediate landmark creation."""
    n = 6
    scen = sc.make_scenario(n, nm, nz, seed=seed, rmax=rmax)
    P = scen["params"]
    orc = ob.OracleFilter(n)
    sc.load_scenario(orc, scen)
    lw0 = np.random.default_rng(seed).uniform(-1.0, 2.0, (n, nm))
    for i in range(n):
        orc.import_gm(i, lw0[i], scen["mean"][i], scen["cov"][i])
    cfg = orc.default_fastslam_config()
    orc.set_fastslam_config(cfg)
    orc.fastslam_update(scen["Z"])
    F = dict(minLog=cfg.minLogMeasurementLikelihood, prior=cfg.landmarkExistencePrior, lock=cfg.landmarkLockWeight, prune=cfg.mapExistencePruneThreshold)
    Pn = dict(R=np.asarray(P["R"]).reshape(2, 2), rmin=P["rmin"], rmax=P["rmax"], rbuf=P["rbuf"], Pd=P["Pd"], clutter=P["clutter"],
              kf_range=P["kf_range"], kf_bearing=P["kf_bearing"])
    pose_cov = np.asarray(scen["pose_cov"]).reshape(3, 3)
    w = orc.get_weights()
    for i in range(n):
        wi, lw, mu, Sig = np_fastslam_update(Pn, F, scen["poses"][i], pose_cov, 1.0, lw0[i], scen["mean"][i], scen["cov"][i], scen["Z"])
        np.testing.assert_allclose(w[i], wi, rtol=1e-9)
        g = orc.export_gm(i)
        assert len(g[0]) == len(lw)
        np.testing.assert_allclose(g[0], lw, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(g[2], mu, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(g[3], Sig, rtol=1e-8, atol=1e-14)
