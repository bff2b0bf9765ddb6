"""The listing rule of the sparse intensity sums (rfs-slam_amd/csrc/weighting.h, step 3b; DESIGN.md section 4, deviation 9), restated in numpy
with the kernel's fp32 arithmetic, against the dense fp64 sums of importanceWeighting (reference include/RBPHDFilter.hpp:776-800).

What the device claims, checked here without a GPU: whatever the rule leaves out of an evaluation point's sum is below 2^-55 of that sum
(after the update: always; before the update: whenever the a-posteriori check on the listed part passes -- and where it fails the kernel
falls back to the dense loop), and a Gaussian whose fp32 image cannot be bounded is listed for every point.  The GPU tests
(test_gpu_parity.py::test_intensity_sums_over_listed_pairs) check the kernel itself against the oracle."""
import numpy as np
import pytest

BITS = np.float32(66.0)          # WEIGHT_SPARSE_BITS
PRIOR_BITS = np.float32(24.0)    # WEIGHT_SPARSE_PRIOR_BITS
LOG2E_HALF = np.float32(0.72134752044448170368)
LOG2_2PI = np.float32(2.65149612947231879804)
f32 = np.float32


def dense_terms(ev, mean, cov, w):
    """w_m N(e; mu_m, Sigma_m) for every (evaluation point, Gaussian): fp64, the reference's formula."""
    d = ev[:, None, :] - mean[None, :, :]
    det = cov[:, 0, 0] * cov[:, 1, 1] - cov[:, 0, 1] * cov[:, 1, 0]
    inv = np.empty_like(cov)
    inv[:, 0, 0] = cov[:, 1, 1] / det
    inv[:, 1, 1] = cov[:, 0, 0] / det
    inv[:, 0, 1] = -cov[:, 0, 1] / det
    inv[:, 1, 0] = -cov[:, 1, 0] / det
    md2 = np.einsum("emi,mij,emj->em", d, inv, d)
    with np.errstate(all="ignore"):
        lik = np.exp(-0.5 * md2) / np.sqrt((2 * np.pi) ** 2 * det)[None, :]
    lik = np.where(np.isnan(lik), 0.0, lik)
    return w[None, :] * lik


def log2_f64_as_f32(x):
    """log2 of a double of any magnitude as the kernel forms it: frexp, fp32 log2 of the mantissa + the exponent."""
    m, e = np.frexp(x)
    with np.errstate(divide="ignore"):
        return np.log2(m.astype(f32)).astype(f32) + e.astype(f32)


def listing(pose, ev, own, mean, cov, w, wp):
    """The kernel's sweep: which (evaluation point, Gaussian) pairs are listed.  own[e] = index of the Gaussian the point is the mean of."""
    gx = (ev[:, 0] - pose[0]).astype(f32)
    gy = (ev[:, 1] - pose[1]).astype(f32)
    g_scale = f32(2.0) * np.max(np.maximum(np.abs(gx), np.abs(gy)))
    det_own = cov[own, 0, 0] * cov[own, 1, 1] - cov[own, 0, 1] ** 2
    with np.errstate(all="ignore"):
        own_term = w[own] / np.sqrt((2 * np.pi) ** 2 * det_own)
    ok = (own_term > 0) & (own_term < 1e300)
    nth = np.where(ok, BITS - log2_f64_as_f32(np.where(ok, own_term, 1.0)), f32(3.0e38)).astype(f32)
    a, b, c = cov[:, 0, 0].astype(f32), cov[:, 0, 1].astype(f32), cov[:, 1, 1].astype(f32)
    fx = (mean[:, 0] - pose[0]).astype(f32)
    fy = (mean[:, 1] - pose[1]).astype(f32)
    with np.errstate(all="ignore"):
        ac = a * c
        det = (ac - b * b).astype(f32)
        rdet = (f32(1.0) / det).astype(f32)
        kr = (-LOG2E_HALF * rdet).astype(f32)
        j00, j01, j11 = c * kr, -b * kr, a * kr
        base = (-LOG2_2PI - f32(0.5) * np.log2(det)).astype(f32)
        cA = log2_f64_as_f32(w) + base
        cB = log2_f64_as_f32(wp) + base + PRIOR_BITS
        cS = np.fmax(cA, cB)
        S = g_scale + np.abs(fx) + np.abs(fy)
        sane = (a > 0) & (c > 0) & (det > f32(1e-3) * ac) & ((a + c) * rdet * S * S < f32(6.0e10)) & (w >= 0) & (wp >= 0) & (S < f32(1.0e6))
    j00, j01, j11 = (np.where(sane, v, f32(0)) for v in (j00, j01, j11))
    fx, fy = np.where(sane, fx, f32(0)), np.where(sane, fy, f32(0))
    cS = np.where(sane, cS, f32(3.0e38)).astype(f32)
    d0 = gx[:, None] - fx[None, :]
    d1 = gy[:, None] - fy[None, :]
    with np.errstate(all="ignore"):
        t0 = d0 * j00[None, :] + d1 * j01[None, :]
        t1 = d0 * j01[None, :] + d1 * j11[None, :]
        h = (t0 * d0 + t1 * d1).astype(f32)
        u = (h + cS[None, :]) + nth[:, None]
    return u > 0, sane, own_term


def check_rule(pose, mean, cov, w, wp, n_eval=15):
    order = np.argsort(-w, kind="stable")
    own = order[:n_eval]
    ev = mean[own]
    listed, sane, own_term = listing(pose, ev, own, mean, cov, w, wp)
    tA = dense_terms(ev, mean, cov, w)
    tB = dense_terms(ev, mean, cov, wp)
    sumA, sumB = tA.sum(1), tB.sum(1)
    assert listed[:, ~sane].all(), "a Gaussian the fp32 image of which is not trusted must be listed everywhere"
    n_checked = 0
    for e in range(len(own)):
        if not (own_term[e] > 0 and np.isfinite(own_term[e])):
            assert listed[e, sane].all()                       # everything listed -> overflow -> the dense loop
            continue
        leftA = tA[e, ~listed[e]].sum()
        assert leftA <= 2.0 ** -55 * sumA[e], (e, leftA / sumA[e])
        assert tA[e, ~listed[e]].max(initial=0.0) <= 2.0 ** -63.5 * sumA[e]        # term by term: 64 bits (66 - the 2 of margin), half a bit for this check
        partB = tB[e, listed[e]].sum()
        if partB >= own_term[e] * 2.0 ** -24 or not (wp > 0).any():    # the kernel's check passes (or there is no prior weight at all)
            leftB = tB[e, ~listed[e]].sum()
            assert leftB <= 2.0 ** -55 * max(sumB[e], 5e-324), (e, leftB, sumB[e])
            assert tB[e, ~listed[e]].max(initial=0.0) <= 2.0 ** -63.5 * max(sumB[e], 5e-324)
            n_checked += 1
    return listed, n_checked


def c2_like(rng, n, spread=2.5, smin=0.02, smax=0.1):
    r = np.sqrt(rng.uniform(0.4, spread ** 2, n))
    a = rng.uniform(-np.pi, np.pi, n)
    mean = np.stack([r * np.cos(a), r * np.sin(a)], 1)
    s = rng.uniform(smin, smax, (n, 2))
    rho = rng.uniform(-0.3, 0.3, n)
    cov = np.zeros((n, 2, 2))
    cov[:, 0, 0], cov[:, 1, 1] = s[:, 0] ** 2, s[:, 1] ** 2
    cov[:, 0, 1] = cov[:, 1, 0] = rho * s[:, 0] * s[:, 1]
    return mean, cov


@pytest.mark.parametrize("seed", range(6))
def test_rule_on_c2_like_mixtures(seed):
    """200 landmarks (w_prev > 0, faded to 1 % where detected) + 80 updated copies (w_prev = 0) near their parents: the shape of configs[1]."""
    rng = np.random.default_rng(seed)
    mean, cov = c2_like(rng, 200)
    wp = rng.uniform(0.3, 1.0, 200)
    w = wp.copy()
    det = rng.choice(200, 80, replace=False)
    w[det] *= 0.01
    cmean = mean[det] + rng.normal(0, 0.02, (80, 2))
    ccov = cov[det] * rng.uniform(0.2, 0.6, (80, 1, 1))
    mean, cov = np.concatenate([mean, cmean]), np.concatenate([cov, ccov])
    w = np.concatenate([w, rng.uniform(0.76, 0.99, 80)])
    wp = np.concatenate([wp, np.zeros(80)])
    listed, n_checked = check_rule(rng.normal(0, 0.02, 2), mean, cov, w, wp)
    assert n_checked == 15                                        # the prior check passes at every point of such a mixture
    assert listed.mean() < 0.25                                   # and the lists are short (the kernel: 301 of 4215 pairs at configs[1])


@pytest.mark.parametrize("kind", ["tiny_sigma", "tight_pairs_at_the_threshold", "correlated", "far_away", "faint_parents", "huge_range_of_weights", "indefinite"])
def test_rule_on_adversarial_mixtures(kind):
    rng = np.random.default_rng(sum(map(ord, kind)))
    mean, cov = c2_like(rng, 260)
    wp = rng.uniform(0.3, 1.0, 260)
    w = rng.uniform(0.3, 1.0, 260)
    pose = np.zeros(2)
    if kind == "tiny_sigma":          # standard deviations down to 1e-6 m: beyond what the fp32 difference resolves -> listed everywhere
        cov[::3] *= 10.0 ** rng.uniform(-9, -3, (len(cov[::3]), 1, 1))
    elif kind == "tight_pairs_at_the_threshold":   # pairs of Gaussians with sigma 3e-8 ... 1e-6 m, 8 ... 9.4 sigma apart, 3 m from the pose: the pairs whose
        sg = 10.0 ** rng.uniform(-7.5, -6, 40)       # log2 term sits AT the listing threshold and whose fp32 difference is off by per cents of sigma --
        k = rng.uniform(8, 9.4, 40)                # without the guard on tr(Sigma^-1) S^2 the rule drops terms it must keep
        mean[40:80] = mean[:40] + np.stack([k * sg, np.zeros(40)], 1)
        for blk in (slice(0, 40), slice(40, 80)):
            cov[blk] = 0.0
            cov[blk, 0, 0] = cov[blk, 1, 1] = sg ** 2
        w[:80] = rng.uniform(2.0, 3.0, 80)         # the evaluation points
        wp[:80] = 0.0                              # (no prior weight: only the 66-bit criterion lists them)
    elif kind == "correlated":        # correlation up to 1 - 1e-6
        k = rng.choice(260, 60, replace=False)
        rho = 1 - 10.0 ** rng.uniform(-6, -1, 60)
        cov[k, 0, 1] = cov[k, 1, 0] = rho * np.sqrt(cov[k, 0, 0] * cov[k, 1, 1])
    elif kind == "far_away":          # the map 3 km from the pose: the fp32 differences lose 10 bits
        mean += 3000.0
    elif kind == "faint_parents":     # prior weights 1e-9: the a-posteriori check decides
        wp[::2] = 1e-9
    elif kind == "huge_range_of_weights":
        w *= 10.0 ** rng.uniform(-60, 0, 260)
        wp *= 10.0 ** rng.uniform(-60, 0, 260)
    elif kind == "indefinite":
        cov[::5, 0, 0] *= -1
    check_rule(pose, mean, cov, w, wp)
