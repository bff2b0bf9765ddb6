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
print("wrote", len(perm), "permlex cases and", len(cases), "ranked-assignment cases")
