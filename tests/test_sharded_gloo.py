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


def _worker(rank, world, port, n_total, force_resample, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from __graft_entry__ import load_package
    from oracle import binding as ob
    pkg = load_package()
    sc = pkg.scenarios
    ob.set_threads(1)
    scen = sc.make_scenario(n_total, 40, 12, seed=31, params=dict(min_updates=1))
    scen["particle_w"] = np.random.default_rng(1).uniform(0.2, 1.0, n_total)
    local = ob.OracleFilter(n_total // world, stable_sort=True)
    sc.load_scenario(local, shard_scen(scen, rank, world))
    sh = pkg.sharded.ShardedRBPHDFilter(local)
    sh.effNParticles_t = n_total + 1.0 if force_resample else 1e-9     # always / never resample
    fired = sh.update(scen["Z"], u01=0.4321)
    res = dict(rank=rank, fired=fired, w=local.get_weights(), sizes=local.gm_sizes(),
               maps=[local.export_gm(i) for i in range(local.n)], unused=[local.get_unused(i) for i in range(local.n)])
    q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def run_world(n_total, force_resample, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, force_resample, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get() for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return sorted(out, key=lambda d: d["rank"])


def single_process(pkg, ob, n_total, force_resample):
    sc = pkg.scenarios
    scen = sc.make_scenario(n_total, 40, 12, seed=31, params=dict(min_updates=1))
    scen["particle_w"] = np.random.default_rng(1).uniform(0.2, 1.0, n_total)
    f = ob.OracleFilter(n_total, stable_sort=True)
    sc.load_scenario(f, scen)
    sh = pkg.sharded.ShardedRBPHDFilter(f)     # world 1: same host code path, no collectives
    sh.effNParticles_t = n_total + 1.0 if force_resample else 1e-9
    fired = sh.update(scen["Z"], u01=0.4321)
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
    if force_resample:
        assert np.array_equal(w, np.ones(n_total))     # weights reset to 1 (ParticleFilter.hpp:486-489)
        # the plan really moved particles across the shard boundary (otherwise the test shows nothing)
        ref2, _ = single_process(pkg, ob, n_total, False)
        plan = pkg.engine.systematic_resample_plan(ref2.get_weights(), 0.4321)
        half = n_total // 2
        crossed = [(g, s) for g, s in enumerate(plan) if (g < half) != (s < half)]
        assert len(crossed) > 0
