"""Generates tests/golden/permlex.json and tests/golden/bruteforce_ranked.json by EXECUTING the reference's own
classes (oracle/_ref/librfs_ref.so = unmodified /root/reference/src/{PermutationLexicographic,BruteForceAssignment}.cpp).
Run in the build container (needs /root/reference):  python tests/golden/make_combinatorics_fixtures.py
The fixtures are data (inputs + reference outputs); no reference source text is stored."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402

ob.build(force=True)
ref = ob.load_ref()
assert ref is not None, "oracle/_ref missing: run `make -C oracle ref` with /root/reference present"
here = os.path.dirname(os.path.abspath(__file__))

perm = {}
for nM in range(0, 5):
    for nZ in range(0, 5):
        if nM + nZ == 0:
            continue
        perm[f"{nM}x{nZ}"] = ob.permlex_all(nM, nZ, lib=ref, sym="rfsref_permlex_all").tolist()
with open(os.path.join(here, "permlex.json"), "w") as fh:
    json.dump(perm, fh, separators=(",", ":"))

rng = np.random.default_rng(20260928)
cases = []
for n in (2, 3, 4, 5):
    for _ in range(3):
        Cm = np.round(rng.uniform(-4, 0, (n, n)), 6)
        s, a = ob.ref_bruteforce(Cm)
        cases.append(dict(C=Cm.tolist(), scores=s[:50].tolist(), assignments=a[:50].tolist()))
with open(os.path.join(here, "bruteforce_ranked.json"), "w") as fh:
    json.dump(cases, fh, separators=(",", ":"))

# ---- Murty where the path uses it: EXTENDED tables as rfsMeasurementLikelihood builds them
# (include/RBPHDFilter.hpp:907-940: log-likelihood block floored at -1000, miss-detection diagonal, clutter diagonal, zero block),
# ranked by the reference's BruteForceLinearAssignment over all n! assignments and de-duplicated the way the reference's own
# Murty example validates the real-assignment block (src/examples/linearAssignment_MurtyAlgorithm.cpp:118-127: equal consecutive
# scores are one assignment; stop below -1000).  Kept per case: the table, nR, nC, the number of distinct scores >= -1000 and the
# first 200 of them -- what Murty with setRealAssignmentBlock(nR, nC) must return, call by call.
import math


def extended_table(rng, nR, nC, gate_frac):
    n = nR + nC
    Cm = np.full((n, n), -1000.0)
    L = np.log(rng.uniform(1e-6, 1.0, (nR, nC)))
    L[rng.uniform(size=(nR, nC)) > gate_frac] = -1000.0          # cells outside the gate: likelihood 0 -> BIG_NEG_NUM (:907-916)
    Cm[:nR, :nC] = L
    for r in range(nR):
        Cm[r, nC + r] = math.log(1.0 - rng.uniform(0.5, 0.99))   # log(1 - Pd) (:923-930)
    for c in range(nC):
        Cm[nR + c, c] = math.log(rng.uniform(1e-3, 1e-1))         # log(clutter) (:932-939)
    Cm[nR:, nC:] = 0.0
    return Cm


rng = np.random.default_rng(20260930)
ext = []
for nR, nC in ((3, 4), (4, 3), (4, 4), (5, 3), (3, 5), (2, 6), (4, 5), (5, 4), (6, 3), (3, 6)):   # extended dimension 7, 8, 9
    for gate_frac in (0.5, 0.9):
        Cm = extended_table(rng, nR, nC, gate_frac)
        s, _ = ob.ref_bruteforce(Cm, kmax=math.factorial(nR + nC))
        keep = []
        for i in range(len(s)):
            if s[i] < -1000.0:
                break
            if i == 0 or s[i] != s[i - 1]:
                keep.append(float(s[i]))
        ext.append(dict(nR=nR, nC=nC, C=Cm.tolist(), n_distinct=len(keep), scores=keep[:200]))
with open(os.path.join(here, "murty_extended_ranked.json"), "w") as fh:
    json.dump(ext, fh, separators=(",", ":"))
print("wrote", len(perm), "permlex cases,", len(cases), "ranked-assignment cases and", len(ext), "extended-table cases (n = 7 ... 9)")
