#!/usr/bin/env python3
"""Does the order of EXACTLY EQUAL weights change the filter's results?  (CPU only: oracle against oracle.)

The reference sorts a mixture with `std::sort(gList_.begin(), gList_.end(), weightCompare)`
(include/GaussianMixture.hpp:523-534): unstable, so Gaussians of exactly equal weight come out in an order that depends on
libstdc++'s introsort.  The device (and the oracle in its `stable_sort=True` mode, which every device parity test uses) orders
ties by their index before the sort.  Equal weights are common: every birth Gaussian enters with `birthGaussianWeight`
(include/RBPHDFilter.hpp:1000-1084) and keeps it while it is out of the field of view.  The order matters in three places:
the choice of evaluation points (RBPHDFilter.hpp:747-761), the greedy merge (GaussianMixture.hpp:394-416: row i absorbs
every later row that passes the test, in order), and the order of floating-point sums over the mixture.

This tool drives `OracleFilter(stable_sort=False)` (= the reference's std::sort) and `OracleFilter(stable_sort=True)`
(= the device's order) through ONE realisation and compares after EVERY update:
  * normalised particle weights, rel 1e-9;
  * every particle's mixture as a MULTISET of (w, mu, Sigma), rel 1e-10 / abs 1e-12;
  * mixture sizes, unused-measurement lists, resampling decisions.
It reports how many updates differ, how many of the compared mixtures held tied weights at all (so that "no difference"
cannot mean "no ties"), and the largest deviations seen.

    python tools/tie_order_study.py c1 [steps=3000] [particles=200] [seed=1]
    python tools/tie_order_study.py vp [messages=900] [particles=64] [seed=5]
    python tools/tie_order_study.py fuzz [cases=2000] [seed=1]
    python tools/tie_order_study.py all          (the three above with their defaults; JSON summary on the last line)
"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
sc = pkg.scenarios
ob = importlib.import_module("oracle.binding")

W_RTOL, GM_RTOL, GM_ATOL = 1e-9, 1e-10, 1e-12


class Tally:
    def __init__(self, name):
        self.name = name
        self.updates = 0
        self.updates_differing = 0
        self.mixtures = 0
        self.mixtures_with_ties = 0          # mixtures (after the update) holding >= 2 exactly equal non-zero weights
        self.mixtures_reordered = 0          # same multiset, different order between the two modes
        self.mixtures_differing = 0
        self.size_mismatch = 0
        self.weight_mismatch = 0
        self.unused_mismatch = 0
        self.plan_mismatch = 0
        self.max_w_rel = 0.0
        self.max_gm_abs = 0.0
        self.first_diffs = []
        self.t0 = time.time()

    def compare(self, a, b, tag, particles=None):
        """a = reference order (std::sort), b = device order (ties by index)."""
        n = a.n
        self.updates += 1
        differs = False
        wa, wb = a.get_weights(), b.get_weights()
        wa, wb = wa / wa.sum(), wb / wb.sum()
        with np.errstate(divide="ignore", invalid="ignore"):
            rel = np.where(wa == wb, 0.0, np.abs(wa - wb) / np.maximum(np.abs(wa), 1e-300))
        self.max_w_rel = max(self.max_w_rel, float(rel.max()))
        if rel.max() > W_RTOL:
            self.weight_mismatch += 1
            differs = True
            self._note(tag, "weights rel %.3g" % rel.max())
        sa, sb = a.gm_sizes(), b.gm_sizes()
        if not np.array_equal(sa, sb):
            self.size_mismatch += 1
            differs = True
            self._note(tag, "sizes differ at particles %s" % np.nonzero(sa != sb)[0][:5].tolist())
        if not np.array_equal(a.get_unused_masks(), b.get_unused_masks()):
            self.unused_mismatch += 1
            differs = True
            self._note(tag, "unused-measurement lists differ")
        for i in (range(n) if particles is None else particles):
            ga, gb = a.export_gm(i), b.export_gm(i)
            self.mixtures += 1
            w = ga[0][ga[0] != 0]
            if w.size != np.unique(w).size:
                self.mixtures_with_ties += 1
            if ga[0].size != gb[0].size:
                self.mixtures_differing += 1
                continue
            same_order = all(np.array_equal(x, y) for x, y in zip((ga[0], ga[2], ga[3]), (gb[0], gb[2], gb[3])))
            if same_order:
                continue
            try:
                err = sc.match_gm(ga, gb, GM_RTOL, GM_ATOL)
                self.mixtures_reordered += 1
                if err < 1.0:
                    self.max_gm_abs = max(self.max_gm_abs, err)
            except AssertionError as e:
                self.mixtures_differing += 1
                differs = True
                self._note(tag, "particle %d: %s" % (i, str(e)[:120]))
        if differs:
            self.updates_differing += 1
        return differs

    def _note(self, tag, msg):
        if len(self.first_diffs) < 12:
            self.first_diffs.append("%s: %s" % (tag, msg))

    def summary(self):
        d = {k: getattr(self, k) for k in ("name", "updates", "updates_differing", "mixtures", "mixtures_with_ties", "mixtures_reordered",
                                          "mixtures_differing", "size_mismatch", "weight_mismatch", "unused_mismatch", "plan_mismatch",
                                          "max_w_rel", "max_gm_abs", "first_diffs")}
        d["seconds"] = round(time.time() - self.t0, 1)
        return d


def run_c1(steps=3000, n=200, seed=1):
    """The shipped C1 run (cfg/rbphdslam2dSim.xml: 3000 steps, 50 landmarks) at `n` particles."""
    sd = pkg.sim2d_driver
    data = sd.generate(traj_seed=seed, kmax=steps)
    ref = ob.OracleFilter(n, stable_sort=False)
    dev = ob.OracleFilter(n, stable_sort=True)
    t = Tally("c1_%dsteps_%dparticles_seed%d" % (steps, n, seed))

    def check(k, run, fired):
        if len(run.z_of_step) == 0:
            return
        t.compare(ref, dev, "step %d" % k)

    try:
        run = sd.Sim2dRun([ref, dev], data, seed=seed + 10).run(on_step=check)
        t.resamplings = run.n_resamples
    except AssertionError as e:           # the two modes asked for different resampling plans
        t.plan_mismatch += 1
        t._note("run", str(e)[:120])
    return t


class Tee:
    """Forwards every call to both handles; returns the first one's result; `after_update` runs after every update()."""

    def __init__(self, a, b, after_update):
        self._a, self._b, self._hook = a, b, after_update
        self.n = a.n

    def __getattr__(self, name):
        fa, fb = getattr(self._a, name), getattr(self._b, name)

        def both(*args, **kw):
            r = fa(*args, **kw)
            fb(*args, **kw)
            if name == "update":
                self._hook()
            return r
        return both


def run_vp(messages=900, n=64, seed=5):
    """The Victoria Park extract (tests/golden/victoria_park_extract.npz) through the event-driven host loop."""
    data = np.load(os.path.join(ROOT, "tests", "golden", "victoria_park_extract.npz"))
    P = dict(sc.VP_PARAMS)
    mk = lambda stable: ob.OracleFilter(n, stable_sort=stable, model=pkg.capi.MODEL_VICTORIAPARK_3D)  # noqa: E731
    ref, dev = mk(False), mk(True)
    for f in (ref, dev):
        sc.apply_vp_params(f, P, np.full(361, 70.0))
    t = Tally("vp_%dmessages_%dparticles_seed%d" % (messages, n, seed))
    cnt = [0]

    def hook():
        cnt[0] += 1
        t.compare(ref, dev, "update %d" % cnt[0])

    run = pkg.vp_driver.VictoriaParkRun(Tee(ref, dev, hook), data, P, seed=seed).run(n_messages=messages)
    t.resamplings = run.n_resamples
    return t


def run_fuzz(cases=2000, seed=1):
    """Random 2-D scenarios in the style of tools/fuzz_parity.py (shapes, ranges, weight ties everywhere / a handful of distinct
    weights), two update cycles + predict (births at one common weight) + resampling + two more."""
    rng = np.random.default_rng(seed)
    aux = np.random.default_rng(999)
    t = Tally("fuzz_%dcases_seed%d" % (cases, seed))
    for case in range(cases):
        n = int(rng.integers(3, 8))
        nlm = int(rng.choice([5, 20, 40, 63, 64, 65, 100, 128, 200]))
        nz = int(rng.integers(1, 25))
        kw = dict(n_particles=n, n_landmarks=nlm, n_z=nz, seed=int(rng.integers(1 << 30)), rmax=float(rng.choice([2.5, 4.0, 6.0])),
                  frac_in_fov=float(rng.choice([1.0, 0.6, 0.2])))
        scen = sc.make_scenario(**kw)
        mode = int(rng.integers(0, 4))
        if mode == 1:
            scen["w"][:, ::2] = 0.5                                  # ties everywhere
        elif mode == 2:
            scen["w"][:] = np.round(scen["w"] * 4) / 4                # a handful of distinct weights
        elif mode == 3:
            scen["w"][:, ::3] = 0.01                                  # birth-weight ties (out-of-view births keep 0.01)
        ref, dev = ob.OracleFilter(n, stable_sort=False), ob.OracleFilter(n, stable_sort=True)
        for f in (ref, dev):
            sc.load_scenario(f, scen)
        for cyc in range(4):
            Z = scen["Z"] + 1e-3 * cyc
            for f in (ref, dev):
                f.update(Z)
            t.compare(ref, dev, "case %d cycle %d" % (case, cyc))
            for f in (ref, dev):
                f.normalize_weights(f.weight_sums()[0])
            if cyc == 1:
                w = dev.get_weights()
                plan = pkg.engine.systematic_resample_plan(w / w.sum(), float(aux.random()))
                for f in (ref, dev):
                    f.resample_apply(plan)
            for f in (ref, dev):
                f.predict_map(True)
        for f in (ref, dev):
            f.close()
        if case % 200 == 199:
            print("  fuzz: %d cases, %d updates differ so far" % (case + 1, t.updates_differing), file=sys.stderr, flush=True)
    return t


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what == "none":
        return
    a = [int(x) for x in sys.argv[2:]]
    out = []
    if what in ("c1", "all"):
        out.append(run_c1(*a[:3]) if what == "c1" else run_c1())
    if what in ("vp", "all"):
        out.append(run_vp(*a[:3]) if what == "vp" else run_vp())
    if what in ("fuzz", "all"):
        out.append(run_fuzz(*a[:2]) if what == "fuzz" else run_fuzz())
    res = []
    for t in out:
        s = t.summary()
        s["resamplings"] = getattr(t, "resamplings", None)
        res.append(s)
        print("%-40s updates %5d differing %4d | mixtures %7d with ties %7d reordered %6d differing %5d | max rel weight dev %.2e, max abs "
              "Gaussian dev %.2e | %.0f s" % (s["name"], s["updates"], s["updates_differing"], s["mixtures"], s["mixtures_with_ties"],
                                             s["mixtures_reordered"], s["mixtures_differing"], s["max_w_rel"], s["max_gm_abs"], s["seconds"]))
        for d in s["first_diffs"]:
            print("     ", d)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
