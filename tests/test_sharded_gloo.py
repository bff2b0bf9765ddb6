"""world_size-2 gloo tests (CPU) of the multi-GPU host logic in rfs-slam_amd/sharded.py: particle sharding, the
{sum w, sum w^2} all-reduce, global systematic resampling with cross-shard migration.  The oracle stands in for the
device engine on each rank (test infrastructure: the product's ShardedRBPHDFilter is backend-agnostic); the result
must equal a single-process filter holding all particles."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def shard_scen(scen, rank, world):
    n = scen["n"] // world
    sl = slice(rank * n, (rank + 1) * n)
    out = dict(scen)
    out.update(n=n, poses=scen["poses"][sl], w=scen["w"][sl], mean=scen["mean"][sl], cov=scen["cov"][sl], particle_w=scen["particle_w"][sl])
    return out


def make_scen(sc, n_total, big=False):
    if big:   # configs[2]-sized mixtures: 500 Gaussians per particle, 5 m range limit
        scen = sc.make_scenario(n_total, 500, 30, seed=77, rmax=5.0, params=dict(min_updates=1))
        scen["particle_w"] = np.random.default_rng(3).uniform(0.05, 1.0, n_total) ** 3
    else:
        scen = sc.make_scenario(n_total, 40, 12, seed=31, params=dict(min_updates=1))
        scen["particle_w"] = np.random.default_rng(1).uniform(0.2, 1.0, n_total)
    return scen


def z_stream(scen, k):
    """Measurement sets of the follow-up steps: the scenario's, jittered, with two measurements nobody has seen (-> unused)."""
    rng = np.random.default_rng(1000 + k)
    Z = scen["Z"] + rng.normal(0, 1e-3, scen["Z"].shape)
    Z[-2:, 0] = rng.uniform(1.0, 2.0, 2)
    return Z


def candidate_config(f):
    """A configuration that keeps birth-candidate lists (CountThreshold > 1): the inheritance walk must carry the lists."""
    cfg = f.get_filter_config()
    cfg.birthGaussianMeasurementCountThreshold = 3
    cfg.birthGaussianMeasurementCheckThreshold = 2
    cfg.birthGaussianMeasurementSupportDist = 2.0
    cfg.birthGaussianCurrentMeasurementCountThreshold = 0
    f.set_filter_config(cfg)


def import_some_candidates(f, lo, n_local):
    """Imported birth-candidate lists on an immediate-birth configuration (CountThreshold == 1 never builds any itself): global slot
    g gets g % 3 candidates whose contents name the slot, so that a list that travels with the inheritance walk can be told apart."""
    for i in range(n_local):
        g = lo + i
        k = g % 3
        if k == 0:
            continue
        mean = np.array([[10.0 + g, -3.0 + 0.25 * j] for j in range(k)])
        cov = np.array([[[0.01 * (j + 1), 0.0], [0.0, 0.02]] for j in range(k)])
        f.import_birth_candidates(i, mean, cov, np.arange(1, k + 1, dtype=np.int32), np.full(k, g % 5, dtype=np.int32))


def _worker(rank, world, port, n_total, force_resample, q, big=False, cycles=0, cand=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from __graft_entry__ import load_package
    from oracle import binding as ob
    pkg = load_package()
    sc = pkg.scenarios
    ob.set_threads(1)
    scen = make_scen(sc, n_total, big)
    local = ob.OracleFilter(n_total // world)
    sc.load_scenario(local, shard_scen(scen, rank, world))
    if cand == "imported":
        import_some_candidates(local, rank * (n_total // world), n_total // world)
    elif cand:
        candidate_config(local)
    sh = pkg.sharded.ShardedRBPHDFilter(local)
    sh.effNParticles_t = n_total + 1.0 if force_resample else 1e-9     # always / never resample
    # (the 500-landmark state drives the raw weights to 0 in one update: the big case resamples the loaded state directly)
    fired = sh.resample(u01=0.4321) if big else sh.update(scen["Z"], u01=0.4321)
    rows_total = sh.last_migration["rows_sent"]
    for k in range(cycles):       # predict (births; after a resampling: the reference's inheritance over GLOBAL slots) + update
        sh.predict_map(True)
        if k % 2 == 1:
            sh.predict_map(True)  # a second predict while resampleOccured_ is still set
        sh.update(z_stream(scen, k), u01=0.1 + 0.17 * k)
        rows_total += sh.last_migration["rows_sent"]
    res = dict(rank=rank, fired=fired, ids=(sh.pid.copy(), sh.ppid.copy()), rows_total=rows_total, w=local.get_weights(), sizes=local.gm_sizes(), poses=local.get_poses(),
               maps=[local.export_gm(i) for i in range(local.n)], unused=[local.get_unused(i) for i in range(local.n)],
               cands=[local.export_birth_candidates(i) for i in range(local.n)], migration=sh.last_migration)
    q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def run_world(n_total, force_resample, world=2, big=False, cycles=0, cand=False):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, force_resample, q, big, cycles, cand)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get() for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return sorted(out, key=lambda d: d["rank"])


def single_process(pkg, ob, n_total, force_resample, big=False):
    sc = pkg.scenarios
    scen = make_scen(sc, n_total, big)
    f = ob.OracleFilter(n_total)
    sc.load_scenario(f, scen)
    sh = pkg.sharded.ShardedRBPHDFilter(f)     # world 1: same host code path, no collectives
    assert sh.defer_normalisation is False and sh._handover is None   # (round 6: the trailing normalisation is opt-in; the default keeps the reference's order)
    sh.effNParticles_t = n_total + 1.0 if force_resample else 1e-9
    fired = sh.resample(u01=0.4321) if big else sh.update(scen["Z"], u01=0.4321)
    return f, fired


@pytest.mark.parametrize("force_resample", [False, True])
def test_two_rank_update_matches_single_process(pkg, ob, force_resample):
    n_total = 24
    ref, fired_ref = single_process(pkg, ob, n_total, force_resample)
    out = run_world(n_total, force_resample)
    assert fired_ref == force_resample
    w = np.concatenate([o["w"] for o in out])
    np.testing.assert_allclose(w, ref.get_weights(), rtol=1e-12)
    if not force_resample:
        assert abs(w.sum() - 1.0) < 1e-12          # global normalisation through the all-reduce
    sizes = np.concatenate([o["sizes"] for o in out])
    assert np.array_equal(sizes, ref.gm_sizes())
    maps = [m for o in out for m in o["maps"]]
    unused = [u for o in out for u in o["unused"]]
    sc = pkg.scenarios
    for i in range(n_total):
        assert all(o["fired"] == force_resample for o in out)
        sc.assert_gm_close(maps[i], ref.export_gm(i), 1e-13, 0, ordered=True)
        assert np.array_equal(unused[i], ref.get_unused(i))
    # the pose travels with the particle (Particle::copy): kept slots keep theirs, children carry their parent's
    np.testing.assert_array_equal(np.concatenate([o["poses"] for o in out]), ref.get_poses())
    if force_resample:
        assert np.array_equal(w, np.ones(n_total))     # weights reset to 1 (ParticleFilter.hpp:486-489)
        assert sum(o["migration"]["rows_sent"] for o in out) == sum(o["migration"]["rows_received"] for o in out) > 0
        # the plan really moved particles across the shard boundary (otherwise the test shows nothing)
        ref2, _ = single_process(pkg, ob, n_total, False)
        plan = pkg.engine.systematic_resample_plan(ref2.get_weights(), 0.4321)
        half = n_total // 2
        crossed = [(g, s) for g, s in enumerate(plan) if (g < half) != (s < half)]
        assert len(crossed) > 0


@pytest.mark.parametrize("world,cand", [(2, False), (3, False), (2, True), (3, True), (2, "imported"), (3, "imported")])
def test_sharded_resampling_cycles_inherit_birth_state_as_the_reference(pkg, ob, world, cand):
    """Forced global resamplings with predicts and updates in between, over 2 and 3 ranks: maps, unused lists, weights and particle
    ids equal those of ONE plain filter holding all particles in RFSGPU_INHERIT_REFERENCE mode (the engine's own, single-GPU
    implementation of include/RBPHDFilter.hpp:1005-1011) driven through the same decisions -- i.e. the sharded host's mask
    exchange over GLOBAL slots reproduces the reference's slot-ordered copy across shard boundaries.  cand: a configuration that
    keeps birth-candidate lists -- the sharded host then carries out the walk level by level (lists of parent slots on other
    ranks travel as objects), and the candidate lists (means, covariances, supports, checks) must match too.  cand == "imported":
    candidate lists put into an IMMEDIATE-birth configuration from outside: rfsgpu_has_birth_candidates is then 1 on
    the shards, every rank must leave the closed form over the masks for the full walk (the lists move with :1005-1011 although
    CountThreshold == 1 never reads them), exactly as rfsgpu_group_predict_map does."""
    n_total, cycles = 24, 4
    sc = pkg.scenarios
    scen = make_scen(sc, n_total)
    ref = ob.OracleFilter(n_total)
    assert ref.get_birth_inheritance() == pkg.capi.INHERIT_REFERENCE
    sc.load_scenario(ref, scen)
    if cand == "imported":
        import_some_candidates(ref, 0, n_total)
        assert ref.has_birth_candidates()
    elif cand:
        candidate_config(ref)
    else:
        assert not ref.has_birth_candidates()

    def step(Z, u01):
        ref.update(Z)
        s = ref.weight_sums()
        ref.normalize_weights(s[0])
        fired, wn, src = ob.resample_decide(ref.get_weights(), n_total + 1.0, u01)
        assert fired
        ref.resample_apply(src)

    step(scen["Z"], 0.4321)
    for k in range(cycles):
        ref.predict_map(True)
        if k % 2 == 1:
            ref.predict_map(True)
        step(z_stream(scen, k), 0.1 + 0.17 * k)
    out = run_world(n_total, True, world=world, cycles=cycles, cand=cand)
    maps = [m for o in out for m in o["maps"]]
    unused = [u for o in out for u in o["unused"]]
    cands = [c for o in out for c in o["cands"]]
    assert np.array_equal(np.concatenate([o["sizes"] for o in out]), ref.gm_sizes())
    n_cand = 0
    for i in range(n_total):
        sc.assert_gm_close(maps[i], ref.export_gm(i), 1e-13, 0, ordered=True)
        assert np.array_equal(unused[i], ref.get_unused(i))
        mr, cr, sr, kr = ref.export_birth_candidates(i)
        assert list(cands[i][2]) == list(sr) and list(cands[i][3]) == list(kr), i
        np.testing.assert_array_equal(cands[i][0], mr)
        np.testing.assert_array_equal(cands[i][1], cr)
        n_cand += len(sr)
    assert (n_cand > 0) == (cand is True)     # (imported lists on CountThreshold == 1 are promoted by the first birth predict: they end up in the MAPS)
    np.testing.assert_array_equal(np.concatenate([o["poses"] for o in out]), ref.get_poses())
    ids, par = ref.get_particle_ids()
    for o in out:
        assert np.array_equal(o["ids"][0], ids) and np.array_equal(o["ids"][1], par)
    assert np.any(ids != np.arange(n_total))
    lvl = np.zeros(n_total, dtype=int)
    for i in range(n_total):
        lvl[i] = 0 if par[i] >= i else lvl[par[i]] + 1
    assert lvl.max() >= 1 and sum(o["rows_total"] for o in out) > 0


def test_two_rank_migration_of_c3_sized_mixtures(pkg, ob):
    """configs[2]'s shape: 500-Gaussian mixtures cross the shard boundary as packed rows in a forced global resampling; poses,
    maps, unused lists and weights equal the single-process filter's."""
    n_total = 8
    ref, fired = single_process(pkg, ob, n_total, True, big=True)
    assert fired
    out = run_world(n_total, True, big=True)
    sc = pkg.scenarios
    maps = [m for o in out for m in o["maps"]]
    assert np.array_equal(np.concatenate([o["sizes"] for o in out]), ref.gm_sizes())
    assert ref.gm_sizes().max() > 100
    for i in range(n_total):
        sc.assert_gm_close(maps[i], ref.export_gm(i), 1e-13, 0, ordered=True)
    np.testing.assert_array_equal(np.concatenate([o["poses"] for o in out]), ref.get_poses())
    assert sum(o["migration"]["rows_sent"] for o in out) > 0
    assert all(o["migration"]["bytes_sent"] == o["migration"]["rows_sent"] * ref.slab_row_bytes() for o in out)


def test_vectorised_resample_plan_equals_the_reference_loop(pkg):
    e = pkg.engine
    rng = np.random.default_rng(5)
    for t in range(60):
        n = int(rng.integers(256, 2500))
        w = rng.uniform(0, 1, n) ** int(rng.integers(1, 12))
        if t % 4 == 0:
            w[rng.integers(0, n, n // 3)] = 0.0
        w /= w.sum()
        u = float(rng.random())
        assert np.array_equal(e._systematic_resample_plan_loop(w, u), e._systematic_resample_plan_vectorised(w, u))
    w = np.full(20000, 1.0 / 20000)
    assert np.array_equal(e.systematic_resample_plan(w, 0.5), np.arange(20000))
