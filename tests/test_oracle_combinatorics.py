"""Pins the oracle's combinatorial restatements (CPU only).

 - mat_perm against the reference's own known-answer test (test/MatrixPermanentTest.hpp:55-87)
 - PermutationLexicographic against the REAL reference class compiled into oracle/_ref (when present)
   and against the committed golden fixture generated from it (tests/golden/permlex.json)
 - Hungarian against scipy.optimize.linear_sum_assignment
 - Murty (plain k-best) against the reference's BruteForceLinearAssignment (oracle/_ref + golden fixture)
 - Murty with the real-assignment block + lexicographic enumeration against an independent brute force
"""
import itertools
import json
import math
import os

import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment

GOLD = os.path.join(os.path.dirname(__file__), "golden")

DERANGEMENTS = {2: 1, 3: 2, 4: 9, 5: 44, 6: 265, 7: 1854, 8: 14833, 9: 133496, 10: 1334961, 11: 14684570, 12: 176214841}


def test_mat_perm_known_answers(ob):
    # reference test: ones - identity, n = 2..12, ASSERT_DOUBLE_EQ (4 ulp)
    for n, want in DERANGEMENTS.items():
        A = np.ones((n, n)) - np.eye(n)
        got = ob.mat_perm(A)[0]
        assert abs(got - want) <= 4 * np.spacing(float(want)), (n, got, want)


def brute_perm(A):
    n = A.shape[0]
    return sum(np.prod([A[i, p[i]] for i in range(n)]) for p in itertools.permutations(range(n)))


def test_mat_perm_random_vs_bruteforce(ob):
    rng = np.random.default_rng(0)
    for n in range(1, 8):
        A = rng.uniform(-1, 1, (3, n, n))
        got = ob.mat_perm(A)
        for b in range(3):
            assert np.isclose(got[b], brute_perm(A[b]), rtol=1e-10, atol=1e-12)


def test_permlex_matches_reference_build(ob):
    ref = ob.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    for nM in range(0, 5):
        for nZ in range(0, 5):
            if nM + nZ == 0:
                continue
            a = ob.permlex_all(nM, nZ)
            b = ob.permlex_all(nM, nZ, lib=ref, sym="rfsref_permlex_all")
            assert a.shape == b.shape and np.array_equal(a, b), (nM, nZ)


def test_permlex_matches_golden_fixture(ob):
    with open(os.path.join(GOLD, "permlex.json")) as fh:
        gold = json.load(fh)
    for key, perms in gold.items():
        nM, nZ = map(int, key.split("x"))
        a = ob.permlex_all(nM, nZ)
        assert a.tolist() == perms, key


def test_permlex_counts():
    # sum_k C(r,k) C(c,k) k!  (SURVEY a11): 7 for 2x2, 34 for 3x3, 209 for 4x4
    from oracle import binding as ob
    for r, c, want in [(2, 2, 7), (3, 3, 34), (4, 4, 209), (1, 0, 1), (0, 1, 1), (3, 5, 136)]:
        assert ob.permlex_all(r, c).shape[0] == want


def test_hungarian_vs_scipy(ob):
    rng = np.random.default_rng(1)
    for n in range(1, 12):
        for _ in range(5):
            Cm = rng.uniform(-5, 5, (n, n))
            ok, soln, cost, Cafter = ob.hungarian(Cm)
            assert ok
            r, c = linear_sum_assignment(Cm, maximize=True)
            assert np.isclose(cost, Cm[r, c].sum(), rtol=1e-12, atol=1e-12)
            assert sorted(soln.tolist()) == list(range(n))
            # the in-place offset is restored up to rounding (HungarianMethod.hpp:244-250)
            assert np.allclose(Cafter, Cm, rtol=0, atol=1e-12)


def test_murty_plain_vs_reference_bruteforce(ob):
    ref = ob.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(2)
    for n in (2, 3, 4, 5, 6):
        Cm = rng.uniform(-3, 0, (n, n))
        want, _ = ob.ref_bruteforce(Cm)
        k = min(200, math.factorial(n))
        got, assign = ob.murty(Cm, kmax=k)
        assert got.size == k
        assert np.allclose(got, want[:k], rtol=0, atol=1e-9)
        assert len({tuple(a) for a in assign.tolist()}) == k  # all distinct


def test_murty_plain_vs_golden_fixture(ob):
    with open(os.path.join(GOLD, "bruteforce_ranked.json")) as fh:
        gold = json.load(fh)
    for case in gold:
        Cm = np.array(case["C"])
        want = np.array(case["scores"])
        got, _ = ob.murty(Cm, kmax=len(want))
        assert np.allclose(got, want, rtol=0, atol=1e-9)


def _extended_cases():
    with open(os.path.join(GOLD, "murty_extended_ranked.json")) as fh:
        return json.load(fh)


def test_murty_with_the_real_assignment_block_vs_golden_fixture(ob):
    """Murty where the path uses it: extended tables of dimension 7 ... 9 built the way rfsMeasurementLikelihood
    builds them (include/RBPHDFilter.hpp:907-940), setRealAssignmentBlock(nR, nC) as at :942-947.  The fixture holds the ranking
    by the reference's OWN BruteForceLinearAssignment over all n! assignments, de-duplicated as the reference's Murty example does
    (src/examples/linearAssignment_MurtyAlgorithm.cpp:118-127).  The oracle's Murty must return exactly those scores, call by
    call, and run dry where the distinct assignments do (no duplicate through the dummy block, none missing)."""
    cases = _extended_cases()
    assert len(cases) >= 20 and {c["nR"] + c["nC"] for c in cases} == {7, 8, 9}
    for case in cases:
        Cm = np.array(case["C"])
        want = np.array(case["scores"])
        got, assign = ob.murty(Cm, case["nR"], case["nC"], kmax=200)
        got = got[got >= -1000.0]                       # (the caller's loop stops at the first score below BIG_NEG_NUM, :951-952)
        assert got.size == min(case["n_distinct"], 200), (case["nR"], case["nC"], got.size, case["n_distinct"])
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=0)
        assert np.all(np.diff(got) <= 0)                # ranked: never increasing (the premise of the device's early end of the loop)


def test_murty_with_the_real_assignment_block_vs_reference_bruteforce_live(ob):
    """The same comparison against the reference class executed here (fresh tables, incl. dimension 9 with > 200 distinct assignments)."""
    ref = ob.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(77)
    for nR, nC in ((4, 4), (5, 4), (4, 5), (3, 6)):
        n = nR + nC
        Cm = np.full((n, n), -1000.0)
        L = np.log(rng.uniform(1e-6, 1.0, (nR, nC)))
        L[rng.uniform(size=(nR, nC)) > 0.8] = -1000.0
        Cm[:nR, :nC] = L
        Cm[np.arange(nR), nC + np.arange(nR)] = np.log(1.0 - rng.uniform(0.5, 0.99, nR))
        Cm[nR + np.arange(nC), np.arange(nC)] = np.log(rng.uniform(1e-3, 1e-1, nC))
        Cm[nR:, nC:] = 0.0
        s, _ = ob.ref_bruteforce(Cm, kmax=math.factorial(n))
        keep = [s[i] for i in range(len(s)) if s[i] >= -1000.0 and (i == 0 or s[i] != s[i - 1])]
        got, _ = ob.murty(Cm, nR, nC, kmax=200)
        got = got[got >= -1000.0]
        assert got.size == min(len(keep), 200)
        np.testing.assert_allclose(got, keep[:got.size], rtol=1e-12, atol=0)


def enumerate_partial(Lp, pd, clutter):
    """Independent brute force of sum over partial assignments (rows -> distinct cols or miss)."""
    r, c = Lp.shape
    total = 0.0
    for k in range(0, min(r, c) + 1):
        for rows in itertools.combinations(range(r), k):
            for cols in itertools.permutations(range(c), k):
                t = 1.0
                for a, b in zip(rows, cols):
                    t *= Lp[a, b]
                for a in range(r):
                    if a not in rows:
                        t *= 1 - pd[a]
                t *= clutter ** (c - k)
                total += t
    return total


def test_partition_likelihood_small_connected(ob):
    rng = np.random.default_rng(3)
    for r, c in [(1, 1), (2, 2), (3, 3), (2, 4), (4, 3), (4, 4)]:
        L = rng.uniform(0.1, 2.0, (r, c))  # fully connected -> a single partition, r+c <= 8 -> lexicographic
        pd = rng.uniform(0.5, 0.99, r)
        cl, ci = 1e-2, 0.37
        got, mc, lr = ob.partition_likelihood(L, pd, cl, ci)
        want = enumerate_partial(L, pd, cl) / ci
        assert mc == 0
        assert np.isclose(got, want, rtol=1e-11)


def test_partition_likelihood_murty_matches_enumeration_when_few_terms(ob):
    # r+c > 8 -> Murty-200; sparse matrix with < 200 significant assignments -> equals the full sum
    rng = np.random.default_rng(4)
    r, c = 5, 5
    L = np.zeros((r, c))
    for i in range(r):       # a chain so the component is connected but sparse
        L[i, i] = rng.uniform(0.5, 2)
        if i + 1 < c:
            L[i, i + 1] = rng.uniform(0.5, 2)
    pd = np.full(r, 0.9)
    cl, ci = 1e-3, 1.0
    got, mc, lr = ob.partition_likelihood(L, pd, cl, ci)
    assert mc == 1
    want = enumerate_partial(L, pd, cl)
    assert np.isclose(got, want, rtol=1e-9), (got, want)


def test_partition_indexing_quirk(ob):
    """SURVEY §7 hard part 2 probe: rows {0,1} isolated, 2<->col0, 3<->col1.
    Visited: p0 = zero partition [rows 0,1] -> Pd0*Pd1 ; p1 = lone row 1 as NON-zero -> (1-Pd1) ;
    p2 = [row2,col0] ; the component [row3,col1] is never visited."""
    L = np.zeros((4, 2))
    L[2, 0] = 0.7
    L[3, 1] = 0.4
    pd = np.array([0.9, 0.8, 0.95, 0.85])
    cl, ci = 1e-2, 1.0
    got, mc, lr = ob.partition_likelihood(L, pd, cl, ci)
    p2 = L[2, 0] + (1 - pd[2]) * cl
    want = (pd[0] * pd[1]) * (1 - pd[1]) * p2
    assert lr == 1
    assert np.isclose(got, want, rtol=1e-12)


def test_partition_no_eval_points(ob):
    # nE = 0: all measurements clutter -> c^nZ / integral
    L = np.zeros((0, 3))
    got, _, _ = ob.partition_likelihood(L, np.zeros(0), 0.5, 2.0)
    assert np.isclose(got, 0.5 ** 3 / 2.0)


def test_cost_matrix_reduce_keeps_the_optimum(ob):
    """CostMatrix::reduce (src/CostMatrix.cpp:263-340) as restated for FastSLAM: pairs fixed by the reduction are mutual
    single possibilities, and the optimum of the full table equals the fixed score plus the optimum of the reduced
    table (scipy's Hungarian as the independent solver)."""
    import ctypes as C
    from scipy.optimize import linear_sum_assignment
    lib = ob.load()
    rng = np.random.default_rng(3)
    lim = -10.0
    for trial in range(60):
        n = int(rng.integers(1, 12))
        T = np.full((n, n), lim)
        k = int(rng.integers(0, 2 * n + 1))
        for _ in range(k):
            T[rng.integers(0, n), rng.integers(0, n)] = rng.uniform(-9.5, 2.0)
        T0 = T.copy()
        a_fixed = np.zeros(n, np.int32); iRed = np.zeros(n, np.int32); jRed = np.zeros(n, np.int32)
        Tc = np.ascontiguousarray(T)
        nRed = lib.rfsor_cost_matrix_reduce(Tc.ctypes.data_as(C.c_void_p), C.c_int(n), C.c_double(lim),
                                            a_fixed.ctypes.data_as(C.c_void_p), iRed.ctypes.data_as(C.c_void_p), jRed.ctypes.data_as(C.c_void_p))
        match = T0 > lim
        fixed_score = 0.0
        for i in range(n):
            j = a_fixed[i]
            if j >= 0 and match[i, j]:
                assert match[i].sum() == 1 and match[:, j].sum() == 1
            if j >= 0:
                fixed_score += T0[i, j] if match[i, j] else lim
        r, c = linear_sum_assignment(T0, maximize=True)
        full = T0[r, c].sum()
        if nRed > 0:
            sub = T0[np.ix_(iRed[:nRed], jRed[:nRed])]
            rr, cc = linear_sum_assignment(sub, maximize=True)
            red = sub[rr, cc].sum()
        else:
            red = 0.0
        np.testing.assert_allclose(fixed_score + red, full, rtol=0, atol=1e-9)
