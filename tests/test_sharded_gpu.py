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


@pytest.mark.parametrize("n_shards", [2, 3])
def test_group_of_shards_in_one_process_matches_single_filter(pkg, n_shards):
    """rfsgpu_group_* (the C-ABI's own multi-GPU form: one host thread, a shard per device id, hipMemcpyPeerAsync row transport) with
    every shard on this box's one GPU: update + normalise + forced global resampling (rows cross the shard boundaries), then a
    second predict/update cycle, equal a single handle holding all particles."""
    sc = pkg.scenarios
    n_total = 30
    scen = sc.make_scenario(n_total, 40, 12, seed=31, params=dict(min_updates=1))
    scen["particle_w"] = np.random.default_rng(1).uniform(0.2, 1.0, n_total)
    ref = pkg.RBPHDFilter(n_total, device_id=0, gm_capacity=192)
    sc.load_scenario(ref, scen)
    grp = pkg.FilterGroup(n_total, [0] * n_shards, gm_capacity=192)
    sc.load_scenario(grp, scen)
    assert [s.n for s in grp.shards] == [10] * 3 if n_shards == 3 else [15, 15]

    ref.update(scen["Z"])
    sums = grp.update(scen["Z"])
    np.testing.assert_allclose(sums, ref.weight_sums(), rtol=1e-13)
    np.testing.assert_array_equal(grp.get_weights(), ref.get_weights())
    # forced resampling: the same global plan, poses / maps / unused lists follow their particles across shards
    s_ref = ref.weight_sums()
    ref.normalize_weights(s_ref[0])
    plan_ref = pkg.engine.systematic_resample_plan(ref.get_weights(), 0.4321)
    x = ref.get_poses()
    ref.resample_apply(plan_ref)
    fired, plan = grp.resample(n_total + 1.0, 0.4321)
    assert fired
    rows, nbytes = grp.migration_stats()
    if np.array_equal(plan, plan_ref):      # (the group's sums are added per shard: the normalised weights may differ in the last bit)
        assert rows > 0 and nbytes == rows * grp.shards[0].slab_row_bytes()
        np.testing.assert_array_equal(grp.get_poses(), x[plan_ref])
        assert np.array_equal(grp.gm_sizes(), ref.gm_sizes())
        for i in range(n_total):
            for a, b in zip(grp.export_gm(i), ref.export_gm(i)):
                assert np.array_equal(a, b)
            assert np.array_equal(grp.get_unused(i), ref.get_unused(i))
        np.testing.assert_array_equal(grp.get_weights(), np.ones(n_total))
        # a second cycle on the resampled state
        ref.set_poses(x[plan_ref], scen["pose_cov"])
        ref.predict_map(True)
        grp.predict_map(True)
        ref.update(scen["Z"])
        grp.update(scen["Z"])
        np.testing.assert_array_equal(grp.get_weights(), ref.get_weights())
        for i in range(n_total):
            for a, b in zip(grp.export_gm(i), ref.export_gm(i)):
                assert np.array_equal(a, b)
    else:
        pytest.fail("the group's resampling plan differs from the single filter's")
    grp.close(); ref.close()
