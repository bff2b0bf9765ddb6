"""The multi-GPU host logic of rfs-slam_amd/sharded.py with the DEVICE engine on every rank: two ranks share this box's
GPU (gloo for the collectives, as in bench.py's RFS_BENCH_SHARE_GPU hook), particles sharded in contiguous blocks, the
{sum w, sum w^2} all-reduce, global systematic resampling with cross-shard migration of packed mixtures.  The result
must equal a single device filter holding all particles.  (tests/test_sharded_gloo.py runs the same logic on CPU with
the oracle standing in for the engine.)"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _worker(rank, world, port, n_total, force_resample, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from test_sharded_gloo import shard_scen
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from __graft_entry__ import load_package
    pkg = load_package()
    sc = pkg.scenarios
    scen = sc.make_scenario(n_total, 40, 12, seed=31, params=dict(min_updates=1))
    scen["particle_w"] = np.random.default_rng(1).uniform(0.2, 1.0, n_total)
    local = pkg.RBPHDFilter(n_total // world, device_id=0, gm_capacity=192)
    sc.load_scenario(local, shard_scen(scen, rank, world))
    sh = pkg.sharded.ShardedRBPHDFilter(local)
    sh.effNParticles_t = n_total + 1.0 if force_resample else 1e-9     # always / never resample
    fired = sh.update(scen["Z"], u01=0.4321)
    q.put(dict(rank=rank, fired=fired, w=local.get_weights(), sizes=local.gm_sizes(), poses=local.get_poses(), migration=sh.last_migration,
               maps=[local.export_gm(i) for i in range(local.n)], unused=[local.get_unused(i) for i in range(local.n)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("force_resample", [False, True])
def test_two_ranks_device_engine_match_single_filter(pkg, force_resample):
    import torch.multiprocessing as mp
    from test_sharded_gloo import free_port
    n_total, world = 24, 2
    sc = pkg.scenarios
    scen = sc.make_scenario(n_total, 40, 12, seed=31, params=dict(min_updates=1))
    scen["particle_w"] = np.random.default_rng(1).uniform(0.2, 1.0, n_total)
    ref = pkg.RBPHDFilter(n_total, device_id=0, gm_capacity=192)
    sc.load_scenario(ref, scen)
    sh = pkg.sharded.ShardedRBPHDFilter(ref)            # world 1: same host code path, no collectives
    sh.effNParticles_t = n_total + 1.0 if force_resample else 1e-9
    assert sh.update(scen["Z"], u01=0.4321) == force_resample

    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, force_resample, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted([q.get() for _ in range(world)], key=lambda d: d["rank"])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(o["fired"] == force_resample for o in out)
    w = np.concatenate([o["w"] for o in out])
    np.testing.assert_allclose(w, ref.get_weights(), rtol=1e-12)
    assert np.array_equal(np.concatenate([o["sizes"] for o in out]), ref.gm_sizes())
    maps = [m for o in out for m in o["maps"]]
    unused = [u for o in out for u in o["unused"]]
    for i in range(n_total):
        sc.assert_gm_close(maps[i], ref.export_gm(i), 1e-13, 0, ordered=True)
        assert np.array_equal(unused[i], ref.get_unused(i))
    np.testing.assert_array_equal(np.concatenate([o["poses"] for o in out]), ref.get_poses())
    if force_resample:
        assert np.array_equal(w, np.ones(n_total))
        assert sum(o["migration"]["rows_sent"] for o in out) > 0     # packed rows really crossed the shard boundary
