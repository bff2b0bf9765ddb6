#!/usr/bin/env python3
"""bench.py -- PHD filter-update steps/s of the MI355X-native RB-PHD update engine.

One "step" = one RBPHDFilter::update() body (map update + particle weighting + GM merge + prune; reference
include/RBPHDFilter.hpp:444-523) over one batch of synthetic input + the weight normalisation
({sum w, sum w^2} reduction, RCCL all-reduce across ranks when N>1, divide).  Workload = BASELINE.json
configs[1] ("C2a"): 2000 particles x 200 GM landmarks x 30 measurements per GPU, all landmarks in
the field of view (worst case: 6000 landmark-measurement pairs per particle), fp64.  The C2a state collapses
after one update (Pd = 0.99), so every step starts from the same device-resident snapshot
(rfsgpu_restore_state, a device-to-device copy of the live Gaussians, INSIDE the timed region).

  python bench.py --gpus N --steps K --warmup W
  N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Particles shard across ranks with no data-path collective except the 2-double all-reduce (weak scaling:
2000 particles per GPU).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_PARTICLES, N_LANDMARKS, N_Z, CAP = 2000, 200, 30, 384
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
BG = 48                # packed 2-D Gaussian record: w, mu(2), Sigma upper triangle(3) doubles
KERNELS = ["phd_update_map", "phd_weight_multifeature", "gm_merge_prune"]   # rfsgpu_update runs merge+prune fused


def algorithmic_bytes(n_particles, nM, nNew, nKept, nZ):
    """ALGORITHMIC HBM bytes per launch of each hot-path kernel (DESIGN.md 'Kernels').
    nM / nNew / nKept are sums over particles of: Gaussians before the update, appended, surviving prune."""
    sweep = nM * BG + nNew * BG + nM * 8 + n_particles * (24 + 8) + nZ * 16
    weight = (nM + nNew) * (BG + 8) + (nM + nNew) * BG + n_particles * (24 + 8)   # read w,w_prev,mu,Sigma; write sorted; weight out
    merge_prune = (nM + nNew) * BG + nKept * BG + n_particles * 4                 # read once, write the compacted survivors
    return dict(zip(KERNELS, [sweep, weight, merge_prune]))


def cpu_baseline(sc, scen_full, seconds_budget=20.0):
    """The oracle (CPU restatement of the same path, oracle/, g++ -O2 -fopenmp, OpenMP `parallel for` over particles
    exactly like the reference) timed on this box's host cores on a bounded sample of the same workload, scaled to the
    full particle count (the path is embarrassingly parallel over particles).  Reported baseline only -- never part
    of the measured GPU path."""
    from oracle import binding as ob
    out, info = {}, {}
    cores = ob.max_threads()
    for label, threads, n_s in (("1thread", 1, 64), ("allcores", cores, min(scen_full["n"], max(64, 8 * cores)))):
        ob.set_threads(threads)
        scen = sc.make_scenario(n_s, N_LANDMARKS, N_Z, seed=12345)
        orc = ob.OracleFilter(n_s, stable_sort=False)
        reps, t_acc, times = 0, 0.0, []
        while t_acc < seconds_budget / 2 and reps < 30:
            sc.load_scenario(orc, scen)
            t0 = time.perf_counter()
            orc.update(scen["Z"])
            s = orc.weight_sums()
            orc.normalize_weights(s[0])
            times.append(time.perf_counter() - t0)
            t_acc += times[-1]
            reps += 1
        out[label] = 1.0 / (float(np.median(times)) / n_s * scen_full["n"])   # median repetition (the box's load varies)
        info[label] = n_s
        orc.close()
    ob.set_threads(cores)
    return dict(value=round(out["allcores"], 4), unit="steps/s", cores=cores, kind="port",
                single_thread_value=round(out["1thread"], 4),
                sample=f"update()+normalise on {info['allcores']} of {scen_full['n']} particles with {cores} OpenMP threads "
                       f"({info['1thread']} particles for the 1-thread figure), same 200-landmark x 30-measurement state, "
                       "scaled by particle count")


def pmc_traffic(kernel_name):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 PMC summary (profiles/pmc_latest.json, made by
    tools/profile_round.sh + tools/pmc_summary.py on this workload): (2*FETCH_SIZE + WRITE_SIZE)*1024."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
    except Exception:
        return None
    alias = {"phd_update_map": "phd_update_map_kernel", "phd_weight_multifeature": "phd_weight_multifeature_kernel",
             "gm_merge_prune": "gm_merge_kernel", "phd_step_fused": "phd_step_fused_kernel"}[kernel_name]
    best = None
    for k, v in d.items():
        if k.startswith(alias) and (best is None or v["calls"] > best["calls"]):
            best = v
    return int(best["hbm_bytes"]) if best else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--particles", type=int, default=N_PARTICLES, help="particles per GPU")
    ap.add_argument("--shards-per-gpu", type=int, default=1,
                    help="independent shards (handles, HIP streams) the GPU's particles are split into.  Measured on MI355X at C2a: "
                         "2 shards that never meet overlap each other's short serial kernels (+20 %% throughput), but the per-step "
                         "weight normalisation couples them again and the gain is lost (4390 vs 4780 steps/s), so the default is 1")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the device path)")
    # Test hook (tests/test_gpu_parity.py): RFS_BENCH_SHARE_GPU=1 lets several ranks share one GPU over gloo so that the
    # N>1 code path runs on a 1-GPU box; the judged runs use one GPU per rank over RCCL ("nccl").
    share = os.environ.get("RFS_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if share else "nccl", rank=rank, world_size=world)

    from __graft_entry__ import load_package
    pkg = load_package()
    sc = pkg.scenarios

    n_local = args.particles
    scen = sc.make_scenario(n_local, N_LANDMARKS, N_Z, seed=12345 + rank)
    # all ranks see the same measurement set (one sensor scan per step)
    scen["Z"] = sc.make_scenario(4, N_LANDMARKS, N_Z, seed=12345)["Z"] if rank else scen["Z"]
    if world > 1:
        zt = torch.from_numpy(scen["Z"].copy()).cuda()
        dist.broadcast(zt, 0)
        scen["Z"] = zt.cpu().numpy()

    # The GPU's particles are held by S shards (handles), each on its own HIP stream (default 1; see --shards-per-gpu).
    # Shards -- on one GPU or on different GPUs -- only meet in the weight normalisation: per-shard {sum w, sum w^2} ->
    # RCCL all-reduce (N > 1) -> on-device divide by the sum over all shards.
    S = args.shards_per_gpu
    if n_local % S:
        raise SystemExit("--particles must be divisible by --shards-per-gpu")
    n_sh = n_local // S
    dev = torch.device("cuda", local_rank)
    shards, streams, evs = [], [], []
    parts = torch.zeros(S, 2, dtype=torch.float64, device=dev)   # shard k's {sum w, sum w^2} lands in parts[k]
    for k in range(S):
        sub = dict(scen)
        lo, hi = k * n_sh, (k + 1) * n_sh
        sub.update(n=n_sh, poses=scen["poses"][lo:hi], w=scen["w"][lo:hi], mean=scen["mean"][lo:hi], cov=scen["cov"][lo:hi],
                   particle_w=scen["particle_w"][lo:hi])
        if np.ndim(scen["pose_cov"]) == 3:
            sub["pose_cov"] = scen["pose_cov"][lo:hi]
        fk = pkg.RBPHDFilter(n_sh, device_id=local_rank, gm_capacity=CAP)
        sc.load_scenario(fk, sub)
        tk = torch.cuda.Stream()
        fk.set_stream(tk.cuda_stream)      # engine kernels and the timing events of this shard order on this stream
        fk.bind_weight_sums_buffer(parts[k].data_ptr())
        fk.save_state()
        shards.append(fk); streams.append(tk); evs.append(torch.cuda.Event())
    ev_red = torch.cuda.Event()
    Z = scen["Z"]
    f = shards[0]
    parts_ptr = parts.data_ptr()

    def step():
        for k in range(S):
            fk = shards[k]
            fk.restore_state()
            fk.update_async(Z)   # stream-ordered: the host never waits inside a step; device errors surface at the final sync
            fk.weight_sums_async()
            if S > 1 or world > 1:
                evs[k].record(streams[k])
        # every shard divides by the sum over all shards of all GPUs: the per-shard pairs sit side by side in `parts`; across
        # GPUs they are all-reduced in place (the only collective on the path: 2*S doubles over xGMI); the divide adds the S
        # entries on the device.  Shard 0's stream carries the reduction.
        if world > 1:
            with torch.cuda.stream(streams[0]):
                for k in range(1, S):
                    streams[0].wait_event(evs[k])
                dist.all_reduce(parts)
                ev_red.record(streams[0])
            for k in range(1, S):
                streams[k].wait_event(ev_red)
        elif S > 1:
            for k in range(S):
                for j in range(S):
                    if j != k:
                        streams[k].wait_event(evs[j])
        for k in range(S):
            shards[k].normalize_weights(0.0, parts_ptr, S)   # divisor read on the device

    for _ in range(args.warmup):
        step()
    # shapes for the algorithmic byte counts (one instrumented step of every shard, untimed)
    nM = nAfter = nKept = 0
    for fk in shards:
        fk.restore_state()
        nM += int(fk.gm_sizes().sum())
        fk.update_map(Z)
        nAfter += int(fk.gm_sizes().sum())
        fk.importance_weighting(); fk.merge(); fk.prune()
        nKept += int(fk.gm_sizes().sum())
    bytes_k = algorithmic_bytes(n_local, nM, nAfter - nM, nKept, N_Z)   # per step of this GPU (all its shards)

    for fk in shards:
        fk.synchronize()
        fk.kernel_time_stats()       # discard the warm-up statistics
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # per-kernel HIP-event pairs recorded on each shard's stream inside every timed step, harvested after the region
    kern_ms = np.zeros(3)
    for fk in shards:
        fk.synchronize()             # raises if any step overflowed / hit an unsupported case
        ka, _ = fk.kernel_time_stats()
        kern_ms += np.array(ka) / 1e6
    kern_ms /= S                     # average duration of ONE launch (one shard); S launches run concurrently
    ms_per_step = dt / args.steps * 1e3
    wsum = sum(float(fk.get_weights().sum()) for fk in shards)
    assert np.isfinite(wsum) and (world > 1 or abs(wsum - 1.0) < 1e-6), "weights did not normalise"

    # rfsgpu_update_async runs the step as ONE kernel (step_fused.h) unless RFSGPU_FUSED_STEP=0: kernel_time_stats then
    # reports [fused step, 0, 0].  The per-phase breakdown (and the likelihood-sweep rate the north star asks for) comes
    # from the three stand-alone kernels on the whole GPU's particles in ONE handle, measured with the same HIP events in an
    # untimed pass after the region.
    fused = kern_ms[1] == 0.0 and kern_ms[2] == 0.0
    phase_ms = kern_ms
    if fused:
        for fk in shards[1:]:
            fk.close()
        fa = pkg.RBPHDFilter(n_local, device_id=local_rank, gm_capacity=CAP)
        sc.load_scenario(fa, scen)
        fa.save_state()
        acc = np.zeros(3)
        reps = 20
        for r in range(reps + 3):
            fa.restore_state()
            fa.update(Z)
            if r >= 3:
                acc += np.array(fa.last_kernel_ns()[:3], dtype=np.float64)
        phase_ms = acc / reps / 1e6

    if rank == 0:
        per_kernel = {}
        for k, name in enumerate(KERNELS):
            gbs = bytes_k[name] / (phase_ms[k] * 1e-3) / 1e9 if phase_ms[k] > 0 else 0.0
            per_kernel[name] = dict(ms=round(float(phase_ms[k]), 5), algorithmic_bytes=int(bytes_k[name]), achieved_GBps=round(gbs, 2))
        if fused:
            tot_bytes = int(sum(bytes_k[name] for name in KERNELS))
            dname = "phd_step_fused"
            per_launch = round((tot_bytes / S) / (kern_ms[0] * 1e-3) / 1e9, 2)
            if S == 1:
                achieved = per_launch
                how = "algorithmic bytes of the launch / its HIP-event duration"
            else:
                # The S launches of a step (one per shard, tot_bytes / S each) overlap on the device, so a single launch's
                # bytes / duration understates what the device moves while they run.  Reported instead: the bytes of ALL S
                # launches over the WHOLE step time (which also contains the short serial kernels) -- a lower bound on the
                # device-level rate during the fused kernels, never an overstatement.
                achieved = round(tot_bytes / (ms_per_step * 1e-3) / 1e9, 2)
                how = (f"algorithmic bytes of the {S} overlapping launches of a step / step time (lower bound); one launch alone: "
                       f"{per_launch} GB/s over {round(float(kern_ms[0]), 5)} ms")
            fused_entry = dict(ms=round(float(kern_ms[0]), 5), algorithmic_bytes=tot_bytes, achieved_GBps=achieved,
                               launches_per_step=S, algorithmic_bytes_per_launch=tot_bytes // S, per_launch_GBps=per_launch,
                               note="update_map + weighting + merge/prune of a particle in one workgroup; one launch per shard "
                                    "per step, measured inside the timed region")
            per_kernel = {"phd_step_fused": fused_entry,
                          "standalone_phases_untimed_pass": per_kernel}
            sweep = per_kernel["standalone_phases_untimed_pass"]["phd_update_map"]
        else:
            dom = int(np.argmax(kern_ms))
            dname = KERNELS[dom]
            achieved = per_kernel[dname]["achieved_GBps"]
            sweep = per_kernel["phd_update_map"]
        out = {
            "metric": "PHD filter-update steps/sec",
            # whole-job aggregate: every rank completes `steps` updates of its own 2000-particle shard in `dt`
            # (weak scaling: the filter grows with the GPUs); at N=1 this is the plain filter-update rate
            "value": round(world * args.steps / dt, 3),
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"C2a: {n_local} particles/GPU x {N_LANDMARKS} GM landmarks x {N_Z} measurements/step, all landmarks in FOV, "
                            "2D RngBrg model, multi-feature weighting (nEvalPt 15), state re-seeded from a device snapshot every step",
                "particles_total": n_local * world,
                "shards_per_gpu": S,
                "unit_definition": "one step = one update(Z) of one shard of %d particles; value sums the shard-steps of all ranks "
                                   "(global filter of %d particles: %.3f updates/s)" % (n_local, n_local * world, args.steps / dt),
                "parallelism": f"particle-sharded: {world} GPU(s) x {S} shard(s) per GPU on separate HIP streams, RCCL all-reduce of 2 doubles/step",
                "gm_after_update": nAfter // n_local, "gm_after_prune": nKept // n_local,
                "kernels": per_kernel,
                "likelihood_sweep": sweep,
            },
            "roofline": {"bound": "hbm", "kernel": dname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": pmc_traffic(dname), "achieved_definition": how if fused else
                         "algorithmic bytes of the launch / its HIP-event duration",
                         "traffic_source": "profiles/pmc_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload)"},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc, scen)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
