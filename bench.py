#!/usr/bin/env python3
"""bench.py -- PHD filter-update steps/s of the MI355X-native RB-PHD update engine (BASELINE.json metric).

One "step" = one RBPHDFilter::update() body (map update + particle weighting + GM merge + prune; reference
include/RBPHDFilter.hpp:444-523) over one batch of synthetic input + the weight normalisation ({sum w, sum w^2}
reduction, RCCL all-reduce across ranks when N>1, divide).  Workloads:

  c2a  BASELINE configs[1]: 2000 particles x 200 GM landmarks x 30 measurements, all landmarks in the field of view
       (worst case: 6000 landmark-measurement pairs per particle).  DEFAULT at every --gpus N (the same per-GPU work: weak scaling).
       The state collapses after one update (Pd = 0.99), so every step starts from a fresh copy of the saved state; the copies of all
       W + K steps are resident in HBM before the timed region (rfsgpu_state_ring_*, a pointer swap per step).
  c3   BASELINE configs[2]'s shard: 2500 particles x 500 GM landmarks x 30 measurements per GPU, range limit 5 m
       (--workload c3 --gpus 8 is configs[2]: 20 000 particles over 8 GPUs).  Same re-seeding.  With N > 1 one global resampling step
       with cross-shard mixture migration is timed separately after the region (`resample_migration`).
  c2b  the C2 map with ~30 landmarks inside the field of view, NOT re-seeded: predict (births) + update + normalise per step,
       a fresh noisy measurement set every step; median over the steps is reported beside the mean.
  c4   BASELINE configs[3]: Victoria Park model (3-D landmarks, scan-based Pd), 5000 particles x 40 landmarks x 12 measurements
       (the shapes of the artificial-clutter run; the raw laser scan is the synthetic one of SURVEY 8(d)), re-seeded.  One fused
       launch per update (vp_step_fused_kernel); packed record B_g = 80 B.
  c5   BASELINE configs[4]: 1000 particles x 200 landmarks x 50 measurements, 40 evaluation points, 10-sigma weighting gate: the
       partitions exceed 8 and go through Murty-200 (murty_jobs_kernel, the dominant kernel of this workload: latency-bound
       assignment searches, its "roofline" line is there for the record).  Re-seeded.

  python bench.py --gpus N --steps K --warmup W [--workload c2a|c2b|c3|c4|c5]
  N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Particles shard across ranks with no data-path collective except the 2-double all-reduce (weak scaling).  Rank 0 prints
ONE JSON line.  roofline.achieved = SURVEY 8(d)'s bytes_step / the fused step kernel's average HIP-event duration inside the
timed region; roofline.traffic = HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of this same
workload, collected live by two short child runs (N = 1, rank 0; --no-pmc skips them); roofline.valu_issue_frac (what binds) from a
third, SQ-counter pass; roofline.flop_frac = SURVEY 8(d)'s algorithmic fp64 flops / kernel time / 78.6 TFLOP/s.  cpu_baseline = the
oracle's restatement timed on the host cores (two labelled figures), boundary = what an update costs through the reference-side
binding and under the unmodified reference driver, config.distributed = world size / backend / PCI bus ids / the collective's form.
"""
import os
import argparse
import csv
import glob
import json
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "c2a": dict(n=2000, nm=200, nz=30, cap=384, rmax=None, frac=1.0, reseed=True,
                label="C2a (BASELINE configs[1]): {n} particles/GPU x 200 GM landmarks x 30 measurements/step, all landmarks in FOV, "
                      "2D RngBrg model, multi-feature weighting (nEvalPt 15), every step on the same saved state (config.state_reseed)"),
    "c3": dict(n=2500, nm=500, nz=30, cap=640, rmax=5.0, frac=1.0, reseed=True,
               label="C3 shard (BASELINE configs[2]: 20000 particles x 500 GM landmarks over 8 GPUs): {n} particles/GPU x 500 GM landmarks x "
                     "30 measurements/step, range limit 5 m, 2D RngBrg model, multi-feature weighting (nEvalPt 15), every step on the "
                     "same saved state (config.state_reseed)"),
    "c2b": dict(n=2000, nm=200, nz=30, cap=384, rmax=None, frac=0.15, reseed=False,
                label="C2b steady state: {n} particles/GPU x 200 GM landmarks (30 inside the FOV) x 30 measurements/step, NOT re-seeded: "
                      "predict (births) + update + normalise per step, fresh measurement noise and clutter every step"),
    "c4": dict(n=5000, nm=40, nz=12, cap=192, model="vp", reseed=True, cpu_sample=1024,
               label="C4 (BASELINE configs[3]): Victoria Park model (Ackerman2D poses, MeasurementModel_VictoriaPark: 3-D landmarks x, y, trunk "
                     "diameter; scan-based Pd), {n} particles x 40 landmarks x 12 measurements/update, multi-feature weighting (nEvalPt 15), "
                     "synthetic ragged 361-beam scan (the dataset's LASER.txt is not in the reference tree), every step on the same saved state (config.state_reseed)"),
    "c5": dict(n=1000, nm=200, nz=50, cap=448, rmax=None, frac=1.0, reseed=True, cpu_sample=64,
               scen_kw=dict(n_clutter=10, n_eval=40, weighting_md=10.0, weights=(0.8, 1.0)), seed=555,
               label="C5 (BASELINE configs[4]): SC-PHD / multi-feature weighting stress, {n} particles x 200 GM landmarks x 50 measurements/step "
                     "(40 detections + 10 clutter), 40 evaluation points, 10-sigma weighting gate -> partitions of extended dimension 9-15 -> "
                     "Murty-200 (bug-compatible with the reference's truncation), fp64, every step on the same saved state (config.state_reseed)"),
}


def make_scen(sc, wl, n, seed_offset=0):
    """The workload's seeded synthetic state for n particles (shared by the GPU run, the PMC child and the CPU baseline)."""
    if wl.get("model") == "vp":
        return sc.make_vp_scenario(n, wl["nm"], wl["nz"], seed=4321 + seed_offset, scan="ragged")
    return sc.make_scenario(n, wl["nm"], wl["nz"], seed=wl.get("seed", 12345) + seed_offset, rmax=wl.get("rmax"), frac_in_fov=wl.get("frac", 1.0),
                            **wl.get("scen_kw", {}))
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
KERNELS = ["phd_update_map", "phd_weight_multifeature", "gm_merge_prune"]   # the three stand-alone kernels of rfsgpu_update
KERNELS_VP = ["vp_update_map", "vp_weighting", "vp_merge_prune"]


def record_bytes(wl):
    """Packed Gaussian record of SURVEY 8(d): B_g = 8 (1 + d_m + d_m (d_m + 1) / 2): 48 B (2-D), 80 B (3-D)."""
    return 80 if wl.get("model") == "vp" else 48


def survey_bytes(n_particles, nM, nNew, nKept, nZ, BG=48, dz=2):
    """SURVEY 8(d), verbatim: bytes_sweep = sum_i[nM_i*B_g + nNew_i*B_g + nM_i*8] + N_p*(24+8) + nZ*8*d_z;
    bytes_step = bytes_sweep + sum_i (nM_i+nNew_i)*B_g + sum_i nKept_i*B_g + N_p*8.  nM/nNew/nKept are sums over particles."""
    sweep = nM * BG + nNew * BG + nM * 8 + n_particles * (24 + 8) + nZ * 8 * dz
    step = sweep + (nM + nNew) * BG + nKept * BG + n_particles * 8
    return sweep, step


FP64_VECTOR_PEAK_TFLOPS = 78.6     # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz (vector fp64, no MFMA on this path)


def survey_flops_totals(n_particles, nM, nNew, nZ, nE):
    """SURVEY 8(d)'s ALGORITHMIC fp64 flops of one launch, "for context": ~70 per (landmark, measurement) pair + ~120 per landmark in
    the sweep, ~15 N^2 / 2 in the merge (N = Gaussians after the map update), and one Gaussian evaluation (~25 flop: difference,
    2 x 2 quadratic form, exp) per (evaluation point, Gaussian) pair in the weighting.  nM / nNew are sums over the n_particles of the
    shard; the merge term uses the per-particle mean N (the mixtures of a synthetic state have the same size to a few per cent).  The
    constants are SURVEY's 2-D ones; the Victoria Park model's 3 x 3 algebra costs more per item (the figure is a lower bound there)."""
    n_after = (nM + nNew) / float(n_particles)
    sweep = 70.0 * nM * nZ + 120.0 * nM
    merge = 15.0 * n_after * n_after / 2.0 * n_particles
    weighting = 25.0 * min(nE, n_after) * (nM + nNew)
    return dict(sweep=sweep, merge=merge, weighting=weighting, total=sweep + merge + weighting)


def design_bytes(n_particles, nM, nNew, nKept, nZ, BG=48):
    """What THIS design's kernels move by construction (DESIGN.md 'Kernels'), beyond the SURVEY formula: the w_prev plane
    (56-byte records), the weighting phase's own read of the mixture and the merge phase's read of it."""
    sweep = nM * BG + nNew * (BG + 8) + nM * 16 + n_particles * (24 + 8) + nZ * 16
    weight = (nM + nNew) * (BG + 8) + n_particles * (24 + 8)
    merge_prune = (nM + nNew) * BG + nKept * (BG + 8) + n_particles * 4
    return dict(zip(KERNELS, [sweep, weight, merge_prune]))


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (reported beside the GPU number; never part of the measured path)
# ---------------------------------------------------------------------------------------------------------------------
def cpu_sample_size(wl, n_full):
    """Particles of the CPU sample: the SAME in every setting of a workload (r02 scaled it with the thread count, which made the
    settings incomparable)."""
    return min(n_full, wl.get("cpu_sample", 512 if wl["nm"] <= 200 else 256))


def cpu_child(sc, wl, n_full, threads, seconds_budget):
    """One measurement of the CPU baseline in a fresh process (fresh OpenMP runtime: the binding comes from the environment
    the parent set).  Prints one JSON object: steps/s scaled to the full particle count, the process CPU time per wall second
    during the timed region (how many CPUs the process really got), and the oracle's TimingInfo buckets."""
    import ctypes as C
    import resource
    from oracle import binding as ob
    so, flags = ob.build_fast()
    lib = C.CDLL(so)
    n_s = cpu_sample_size(wl, n_full)
    ob.set_threads(threads)
    lib.rfsor_set_threads(C.c_int(threads))
    scen = make_scen(sc, wl, n_s)
    kw = dict(model=1) if wl.get("model") == "vp" else {}

    def one():
        orc = ob.OracleFilter(n_s, stable_sort=False, lib=lib, **kw)
        sc.load_scenario(orc, scen)
        t0 = time.perf_counter()
        orc.update(scen["Z"])
        s_ = orc.weight_sums()
        orc.normalize_weights(s_[0])
        dt = time.perf_counter() - t0
        ti = orc.getTimingInfo()
        orc.close()
        return dt, ti

    one()                                   # starts the thread team, touches the allocator arenas
    times, buckets = [], np.zeros(4)
    ru0, w0 = resource.getrusage(resource.RUSAGE_SELF), time.perf_counter()
    t_acc = 0.0
    while t_acc < seconds_budget and len(times) < 30:
        dt, ti = one()
        times.append(dt)
        t_acc += dt
        buckets += np.array([ti.mapUpdate_wall, ti.particleWeighting_wall, ti.mapMerge_wall, ti.mapPrune_wall], dtype=np.float64)
    ru1, w1 = resource.getrusage(resource.RUSAGE_SELF), time.perf_counter()
    cpu = (ru1.ru_utime + ru1.ru_stime) - (ru0.ru_utime + ru0.ru_stime)
    rate = 1.0 / (float(np.median(times)) / n_s * n_full)     # median repetition (the box's load varies)
    tot = float(buckets.sum()) or 1.0
    print(json.dumps(dict(steps_per_s=rate, sample_particles=n_s, repetitions=len(times), cpu_seconds_per_wall_second=cpu / (w1 - w0),
                          timing_buckets_share=dict(mapUpdate=buckets[0] / tot, particleWeighting=buckets[1] / tot, mapMerge=buckets[2] / tot,
                                                    mapPrune=buckets[3] / tot, predict=0.0, particleResample=0.0),
                          flags=flags)), flush=True)


def cpu_baseline(wname, n_full, particles_arg):
    """The oracle (CPU restatement of the same path; OpenMP `parallel for` over particles in every phase exactly like the
    reference) compiled -O3 -march=native -fopenmp ON THIS HOST (oracle.binding.build_fast; the parity tests keep the strict
    -O2 -ffp-contract=off build), timed on a bounded sample of the same workload -- the same sample in every setting -- and
    scaled to the full particle count (the path is independent per particle).  A few (threads, binding) settings are tried, each
    in a fresh process; every setting reports the CPU time it consumed per wall second, i.e. how many CPUs the container really
    gave it (a cgroup quota may or may not be enforced on a given box), and a setting whose speed-up over one thread exceeds
    that number by more than 10 % is rejected as a measurement artefact.  The best accepted setting is the reported value."""
    from oracle import binding as ob
    wl = WORKLOADS[wname]
    model, logical, physical = ob.cpu_info()
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = logical
    quota = None
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] != "max":
            quota = float(q[0]) / float(q[1])
    except Exception:
        pass
    usable = int(min(physical, avail, quota if quota else physical))
    tried, flags, single = [], None, None
    settings = [("false", None, 1)]
    for thr in sorted({max(1, usable), max(1, physical // 2), physical}):
        if thr > 1:
            settings.append(("false", None, thr))
    settings.append(("close", "cores", physical))           # all physical cores, OMP_PROC_BIND=close
    if usable > 1 and usable != physical:
        settings.append(("close", "cores", usable))         # ... and the same binding at the thread count the quota allows
    for bind, places, thr in settings:
        env = dict(os.environ, OMP_PROC_BIND=bind, OMP_NUM_THREADS=str(thr))
        env.pop("OMP_PLACES", None)
        if places:
            env["OMP_PLACES"] = places
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-child", str(thr), "--workload", wname] + (["--particles", str(particles_arg)] if particles_arg else [])
        try:
            r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=180)
            d = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
        except Exception as e:   # noqa: BLE001
            tried.append(dict(omp_proc_bind=bind, omp_places=places, threads=thr, error=str(e)[:100]))
            continue
        flags = d["flags"]
        rec = dict(omp_proc_bind=bind, omp_places=places, threads=thr, steps_per_s=round(d["steps_per_s"], 4), sample_particles=d["sample_particles"],
                   repetitions=d["repetitions"], cpus_used=round(d["cpu_seconds_per_wall_second"], 2),
                   timing_buckets_share={k: round(v, 3) for k, v in d["timing_buckets_share"].items()})
        if thr == 1 and single is None:
            single = rec
        tried.append(rec)
    ok = [r for r in tried if "steps_per_s" in r]
    if not ok or single is None:
        return dict(value=None, unit="steps/s", cores=physical, kind="port", error="no CPU baseline run succeeded", settings_tried=tried)
    for r in ok:
        r["speedup_vs_1_thread"] = round(r["steps_per_s"] / single["steps_per_s"], 2)
        # more speed-up than CPUs consumed: not a parallel speed-up (cache / frequency / sampling artefact) -> not the reported value
        r["accepted"] = bool(r["threads"] == 1 or r["speedup_vs_1_thread"] <= 1.10 * max(1.0, r["cpus_used"]))
    best = max((r for r in ok if r["accepted"]), key=lambda r: r["steps_per_s"])
    # two labelled figures.  `value` stays the sustainable one (what this container's CPU quota lets the OpenMP path
    # hold); the burst figure is the fastest setting of all -- many threads finishing inside one scheduling period of the quota --, i.e.
    # what the box's physical cores can do when nothing throttles them.
    burst = max(ok, key=lambda r: r["steps_per_s"])
    figures = dict(
        quota_bound=dict(value=best["steps_per_s"], threads=best["threads"], cpus_used=best["cpus_used"], omp_proc_bind=best["omp_proc_bind"],
                         label="best setting whose speed-up is covered by the CPUs it consumed (cgroup quota %s CPUs)" % (quota if quota else "none")),
        unthrottled_burst=dict(value=burst["steps_per_s"], threads=burst["threads"], cpus_used=burst["cpus_used"], omp_proc_bind=burst["omp_proc_bind"],
                               accepted=burst["accepted"],
                               label="fastest setting of all (%d threads on %d physical cores); above the quota-bound figure only because a burst "
                                     "shorter than the quota's scheduling period is not throttled" % (burst["threads"], physical)))
    return dict(value=best["steps_per_s"], value_is="quota_bound", figures=figures, unit="steps/s", cores=best["threads"], kind="port",
                kind_note="port = the oracle's restatement of the reference path (oracle/rbphd_oracle.cpp), not the reference itself: the reference "
                          "needs Eigen3 + Boost, absent from this image (SURVEY 8(d) allows the port as the baseline)",
                single_thread_value=single["steps_per_s"], speedup_vs_1_thread=best["speedup_vs_1_thread"], cpus_used_by_best=best["cpus_used"],
                parallel_efficiency=round(best["speedup_vs_1_thread"] / best["threads"], 3),
                timing_buckets_share=best["timing_buckets_share"],
                cpu_model=model, logical_cpus=logical, physical_cores=physical, cpus_available_to_this_container=avail, cgroup_cpu_quota=quota,
                omp_num_threads=best["threads"], omp_proc_bind=best["omp_proc_bind"], omp_places=best["omp_places"],
                compiler_flags="g++ -std=c++17 " + " ".join(flags), settings_tried=tried,
                sample=f"update()+normalise on {best['sample_particles']} of {n_full} particles (the same sample in every setting; median of "
                       f"{best['repetitions']} repetitions), same landmark x measurement state, scaled by particle count; best ACCEPTED setting of "
                       "settings_tried (each a fresh process; cpus_used = process CPU seconds per wall second in the timed region; a setting is "
                       "rejected when its speed-up over one thread exceeds 1.1 x cpus_used -- a cgroup CPU quota is enforced per scheduling period, so a "
                       "many-thread burst shorter than the period can outrun it inside the timed call without being sustainable); timing_buckets_share = the oracle's TimingInfo "
                       "buckets (RBPHDFilter::TimingInfo; predict and resample are not part of a step)")


# ---------------------------------------------------------------------------------------------------------------------
# live HBM traffic (rocprofv3 PMC passes of a short child run of this same workload)
# ---------------------------------------------------------------------------------------------------------------------
def pmc_pass(counters, child_args, timeout_s):
    """Average of each counter in `counters` (a name or a list collected in ONE pass) per launch of every kernel over one
    `rocprofv3 --kernel-trace --pmc <counters>` child run.  One counter: {kernel: (avg, launches)}; a list: {kernel: {counter: (avg, launches)}}."""
    single = isinstance(counters, str)
    names = [counters] if single else list(counters)
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    d = tempfile.mkdtemp(prefix="rfs_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [exe, "--kernel-trace", "--pmc"] + names + ["--output-format", "csv", "-d", d, "-o", "pmc", "--",
           sys.executable, os.path.abspath(__file__), "--pmc-child"] + child_args
    try:
        r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        shutil.rmtree(d, ignore_errors=True)
        return None, f"rocprofv3 pass timed out after {timeout_s} s"
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if r.returncode != 0 or not files:
        tail = r.stdout.decode(errors="replace")[-300:]
        shutil.rmtree(d, ignore_errors=True)
        return None, f"rocprofv3 pass failed (rc {r.returncode}): {tail}"
    agg = {}
    for fpath in files:
        with open(fpath) as fh:
            for row in csv.DictReader(fh):
                cn = row.get("Counter_Name", names[0])
                if cn not in names:
                    continue
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                agg.setdefault(name, {}).setdefault(cn, []).append(float(row["Counter_Value"]))
    shutil.rmtree(d, ignore_errors=True)
    out = {k: {cn: (sum(v) / len(v), len(v)) for cn, v in cs.items()} for k, cs in agg.items()}
    if single:
        return {k: cs[names[0]] for k, cs in out.items() if names[0] in cs}, None
    return out, None


def live_traffic(kernel_prefix, child_args, timeout_s=240):
    """HBM bytes per launch of the kernel whose name starts with `kernel_prefix`: (2*FETCH_SIZE + WRITE_SIZE)*1024 -- rocprofv3
    reports both in KiB and gfx950's FETCH_SIZE tallies 128-byte read requests at 64 bytes (MI355X_MICROARCH.md, HBM section;
    confirmed on restore_state_kernel, a pure copy).  Two separate passes, as the guide prescribes."""
    out = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        agg, err = pmc_pass(counter, child_args, timeout_s)
        if agg is None:
            return None, err
        hit = [(k, v) for k, v in agg.items() if k.startswith(kernel_prefix)]
        if not hit:
            return None, f"no {kernel_prefix} launches in the {counter} pass"
        k, (avg, calls) = max(hit, key=lambda kv: kv[1][1])
        out[counter] = (avg, calls)
    b = (2.0 * out["FETCH_SIZE"][0] + out["WRITE_SIZE"][0]) * 1024.0
    return dict(bytes=int(b), fetch_size_kib=round(out["FETCH_SIZE"][0], 1), write_size_kib=round(out["WRITE_SIZE"][0], 1),
                launches=out["FETCH_SIZE"][1]), None


N_SIMD, N_SE = 1024, 32       # MI355X: 256 CUs x 4 SIMDs; 8 XCDs x 4 shader engines (SQ_BUSY_CYCLES is summed over the SEs)
SQ_COUNTERS = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_SALU", "SQ_WAIT_ANY", "SQ_WAVES"]


def live_issue(kernel_prefix, child_args, timeout_s=240, also=None):
    """The resource that binds a kernel HBM does not: the share of the vector ALUs' issue slots the dominant
    kernel uses.  One more rocprofv3 child pass of this workload (SQ counters only, its own run).  SQ_ACTIVE_INST_VALU counts
    quad-cycles (4 shader cycles) a SIMD spends issuing vector instructions, summed over the device; SQ_BUSY_CYCLES counts
    shader cycles the launch keeps a shader engine busy, summed over the 32 SEs (checked against the launch's duration:
    8.20 M / 32 = 256 k cycles for 122.9 us at 2.09 GHz, profiles/r04l_cut_profile_c2a.txt).  So
        valu_issue_frac = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * SQ_BUSY_CYCLES / 32),
    a ratio of counters of ONE pass -- the clock the profiled run happened to hold cancels."""
    agg, err = pmc_pass(SQ_COUNTERS, child_args, timeout_s)
    if agg is None:
        return None, err

    def figures(prefix):
        hit = [(k, v) for k, v in agg.items() if k.startswith(prefix) and "SQ_BUSY_CYCLES" in v and "SQ_ACTIVE_INST_VALU" in v]
        if not hit:
            return None
        k, cs = max(hit, key=lambda kv: kv[1]["SQ_BUSY_CYCLES"][1])
        v = {cn: cs[cn][0] for cn in cs}
        cyc = v["SQ_BUSY_CYCLES"] / N_SE
        o = dict(valu_issue_frac=round(4.0 * v["SQ_ACTIVE_INST_VALU"] / (N_SIMD * cyc), 4), launch_cycles=int(cyc), launches=cs["SQ_BUSY_CYCLES"][1],
                 counters={cn: int(x) for cn, x in sorted(v.items())})
        if "SQ_WAVE_CYCLES" in v and "SQ_WAIT_ANY" in v and v["SQ_WAVE_CYCLES"] > 0:
            o["wave_wait_frac"] = round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 4)          # share of the waves' resident time spent parked (s_waitcnt / barrier)
            o["waves_per_simd_time_avg"] = round(4.0 * v["SQ_WAVE_CYCLES"] / (N_SIMD * cyc), 3)
        return o
    out = figures(kernel_prefix)
    if out is None:
        return None, f"no {kernel_prefix} launches in the SQ pass"
    if also:      # a second kernel of the same child run (the stand-alone likelihood sweep of the untimed phase pass)
        out["also"] = {also: figures(also)}
    return out, None


# ---------------------------------------------------------------------------------------------------------------------
def boundary_cost(pkg, sc, wl, n_local, CAP, scen, local_rank):
    """What RBPHDFilter::update costs THROUGH the drop-in boundary, beside the stream-ordered step the headline
    times.  (a) The call sequence of integration/RBPHDFilter_rfsgpu.hpp::update (:213-235 there; reference include/RBPHDFilter.hpp:
    444-541) through the C ABI, synchronous, at this workload's shape: the three config structs + the model, rfsgpu_set_poses with a
    covariance per particle, rfsgpu_set_weights, rfsgpu_update (blocks until the GPU is done), rfsgpu_get_weights.  (b) The UNMODIFIED
    reference driver src/rbphdslam2dSim.cpp linked against the binding (tests/support/_build, built in the dev container where
    /root/reference exists), its shipped configuration with nParticles = 2000 and result logging off: the driver's own TimingInfo
    printout (:654-690) divided by its updates."""
    import re
    import subprocess
    import tempfile
    out = {}
    g = pkg.RBPHDFilter(n_local, device_id=local_rank, gm_capacity=CAP)
    sc.load_scenario(g, scen)
    g.save_state()
    Z = scen["Z"]
    x = np.ascontiguousarray(scen["poses"], dtype=np.float64)
    cov = np.ascontiguousarray(np.broadcast_to(np.asarray(scen["pose_cov"], dtype=np.float64), (n_local, 3, 3)))
    w1 = np.ones(n_local)
    P = scen["params"]
    cfg = g.get_filter_config()

    def push_config():
        g.set_filter_config(cfg)                                           # pushConfiguration(): the public config members may change any time
        g.set_kf_config(P["kf_range"], P["kf_bearing"])
        g.set_lmk_process_noise(P["Q_lm"])
        g.set_model_rngbrg(P["R"], P["Pd"], P["clutter"], P["rmax"], P["rmin"], P["rbuf"])

    def seq(mode):
        g.restore_state()
        if mode == "restore_only":
            g.synchronize()
            return
        push_config()
        if mode == "io":        # round 5: integration/RBPHDFilter_rfsgpu.hpp::update -- poses + covariances + weights in, weights out, ONE call, one wait
            g.update_io(Z, poses=x, pose_cov=cov, weights=w1)
            return
        g.set_poses(x, cov)                                                 # rounds 3-4: pushPoses(): mean + 3x3 covariance per particle
        g.set_weights(w1)                                                   # pushWeights()
        g.update(Z)                                                         # rfsgpu_update: synchronous
        g.get_weights()                                                     # pullWeights()
    for _ in range(300):
        seq("io")
    t = {}
    for name, mode in (("io", "io"), ("four_calls", "four"), ("restore_only", "restore_only"), ("io2", "io"), ("four_calls2", "four")):
        S = 300
        t0 = time.perf_counter()
        for _ in range(S):
            seq(mode)
        t[name] = (time.perf_counter() - t0) / S * 1e6
    per = min(t["io"], t["io2"]) - t["restore_only"]
    per4 = min(t["four_calls"], t["four_calls2"]) - t["restore_only"]
    # the parts, one at a time
    parts = {}
    for name, fn in (("config_structs_and_model", push_config),):
        t0 = time.perf_counter()
        for _ in range(300):
            fn()
        parts[name] = round((time.perf_counter() - t0) / 300 * 1e6, 2)
    parts["rfsgpu_update_io"] = round(per - sum(parts.values()), 2)
    out["binding_sequence"] = dict(us_per_update=round(per, 2), parts_us=parts, particles=n_local,
                                   four_call_sequence_us_per_update=round(per4, 2),
                                   note="ctypes calls through the C ABI in the order integration/RBPHDFilter_rfsgpu.hpp::update makes them (round 5: the "
                                        "config structs + rfsgpu_update_io; four_call_sequence = set_poses + set_weights + update + get_weights, what rounds 3-4 "
                                        "measured); the state re-seed between updates is measured alone and subtracted")
    g.close()
    # (b) the unmodified reference driver
    root = os.path.dirname(os.path.abspath(__file__))
    exe = os.path.join(root, "tests", "support", "_build", "rbphdslam2dSim")
    xml = os.path.join(root, "tests", "golden", "rbphdslam2dSim_c1.xml")
    if not (os.path.exists(exe) and os.path.exists(xml)):
        out["reference_driver"] = None
        out["reference_driver_note"] = "tests/support/_build/rbphdslam2dSim not built (needs /root/reference at build time)"
        return out
    try:
        with tempfile.TemporaryDirectory() as tmp:
            steps = 600
            txt = open(xml).read()
            txt = txt.replace("<config>", "<config>\n  <logging><logResultsToFile>0</logResultsToFile><logTimingToFile>0</logTimingToFile>"
                              "<logDirPrefix>%s/</logDirPrefix></logging>" % tmp, 1)
            txt = re.sub(r"<timesteps>\d+</timesteps>", "<timesteps>%d</timesteps>" % steps, txt)
            txt = re.sub(r"<nParticles>\d+</nParticles>", "<nParticles>2000</nParticles>", txt)
            cfgp = os.path.join(tmp, "cfg.xml")
            open(cfgp, "w").write(txt)
            def run_driver(extra_env):
                t0 = time.perf_counter()
                r = subprocess.run([exe, "-c", cfgp, "-t", "1", "-s", "1"], capture_output=True, text=True, timeout=300,
                                   env=dict(os.environ, RFSGPU_DEVICE=str(local_rank), RFSGPU_GM_CAPACITY="256", RFSGPU_BINDING_PROFILE="1", **extra_env))
                wall = time.perf_counter() - t0
                if r.returncode != 0:
                    raise RuntimeError(r.stderr[-300:])
                rows = dict((m.group(1).strip(), (int(m.group(2)), int(m.group(3))))
                            for m in re.finditer(r"^(Prediction|Map Update|Weighting|Map Merge|Map Prune|Resampling|Total)\s+wall:\s*(\d+)\s+cpu:\s*(\d+)", r.stdout, re.M))
                bd = re.search(r"flush \+ configuration ([0-9.]+) \| map part[^|]*\) ([0-9.]+) \| ParticleFilter::propagate[^|]*\) ([0-9.]+) \| lazy=(\d)", r.stderr)
                return rows, wall, (dict(flush_and_configuration=float(bd.group(1)), map_part=float(bd.group(2)), reference_propagate=float(bd.group(3)),
                                         lazy=bool(int(bd.group(4)))) if bd else None)
            n_upd = steps - 1
            rows, wall, bd = run_driver({})                                  # the shipped binding: lazy predict (round 6)
            rows_e, wall_e, bd_e = run_driver({"RFSGPU_LAZY_PREDICT": "0"})  # the eager binding of rounds 3-5, same binary
            out["reference_driver"] = dict(
                binary="src/rbphdslam2dSim.cpp (unmodified) + integration/RBPHDFilter_rfsgpu.hpp + librfsgpu.so", particles=2000, timesteps=steps,
                timing_info_us_per_step={k: round(v[0] / 1e3 / n_upd, 2) for k, v in rows.items()},
                predict_breakdown_us_per_call=bd,
                eager_predict=dict(timing_info_us_per_step={k: round(v[0] / 1e3 / n_upd, 2) for k, v in rows_e.items()}, predict_breakdown_us_per_call=bd_e,
                                   process_wall_s=round(wall_e, 2), env="RFSGPU_LAZY_PREDICT=0"),
                process_wall_s=round(wall, 2),
                note="the driver's own 'Elapsed Timing Information' (wall, ns) / (timesteps - 1); Map Update holds the whole fused device step "
                     "(HIP events) -- with the lazy predict (round 6) including the births + static step at its head --, Prediction and Resampling are "
                     "host timers around the binding's predict() / resample tail; predict_breakdown = the binding's own clock inside predict() "
                     "(RFSGPU_BINDING_PROFILE=1): configuration push | map part (lazy: pose compare + record; eager: pose push + predict launch + wait) | "
                     "ParticleFilter::propagate (the reference's host code, out of scope); shipped C1 scene: ~38 Gaussians x ~10 measurements per particle-update")
    except Exception as e:   # noqa: BLE001
        out["reference_driver"] = None
        out["reference_driver_note"] = "run failed: " + repr(e)[:200]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default=None, help="default: c2a (configs[1]'s shape per GPU) at every --gpus N; c3 = configs[2]'s shard (--workload c3 --gpus 8 is configs[2])")
    ap.add_argument("--particles", type=int, default=None, help="particles per GPU (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-boundary", action="store_true", help="skip the 'boundary' leg (synchronous binding sequence + the unmodified reference driver)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 PMC child runs (roofline.traffic = null)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-child", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.cpu_child:                    # CPU-baseline leg in its own process: no torch, no GPU
        from __graft_entry__ import load_package
        wl = WORKLOADS[args.workload or "c2a"]
        cpu_child(load_package().scenarios, wl, args.particles or wl["n"], args.cpu_child, seconds_budget=4.0)
        return

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
    # Second test hook: RFS_BENCH_FORCE_DIST=1 under a ONE-rank torchrun takes the N > 1 path (RCCL init, broadcast, the all-reduce
    # on the engine's stream, the device-side divide, the all_to_all row migration) with world_size 1 -- the "nccl" calls the
    # judged N > 1 runs make, executed on a 1-GPU box.
    multi = world > 1 or os.environ.get("RFS_BENCH_FORCE_DIST") == "1"
    if share:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if share else "nccl", rank=rank, world_size=world)

    from __graft_entry__ import load_package
    pkg = load_package()
    sc = pkg.scenarios

    # which physical device every rank drives (all-gathered; VERDICT r5 item 7: proof that RCCL saw N distinct devices)
    prop = torch.cuda.get_device_properties(local_rank)
    my_id = [int(getattr(prop, "pci_domain_id", -1)), int(getattr(prop, "pci_bus_id", -1)), int(getattr(prop, "pci_device_id", -1))]
    if multi:
        idt = torch.tensor(my_id, dtype=torch.int64, device="cuda")
        allid = torch.empty(3 * world, dtype=torch.int64, device="cuda")
        if share:
            lst = [torch.empty(3, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(lst, idt.cpu())
            allid = torch.cat(lst)
        else:
            dist.all_gather_into_tensor(allid, idt)
        allid = allid.cpu().numpy().reshape(world, 3)
    else:
        allid = np.array([my_id])
    pci_ids = ["%04x:%02x:%02x" % (d if d >= 0 else 0, b if b >= 0 else 0, v if v >= 0 else 0) for d, b, v in allid.tolist()]
    probe_note = None

    # The SAME per-GPU work at every N (weak scaling): configs[1]'s shape -- the configuration the metric is quoted on -- on each
    # rank, so that value(N) / (N * value(1)) compares like with like.  (Until late in round 3 the N > 1 default was configs[2]'s
    # shard, 2500 x 500 per rank: 1.9x the work of the N = 1 line per GPU.)  configs[2] itself: --workload c3 --gpus 8.
    wname = args.workload or "c2a"
    wl = WORKLOADS[wname]
    n_local = args.particles or wl["n"]
    CAP = wl["cap"]
    N_Z = wl["nz"]
    vp = wl.get("model") == "vp"
    DZ = 3 if vp else 2
    BG = record_bytes(wl)
    kernels = KERNELS_VP if vp else KERNELS
    fused_name = "vp_step_fused_kernel" if vp else "phd_step_fused_kernel"
    scen = make_scen(sc, wl, n_local, seed_offset=rank)
    # all ranks see the same measurement set (one sensor scan per step)
    if multi:
        zt = torch.from_numpy(np.ascontiguousarray(scen["Z"])).cuda()
        dist.broadcast(zt, 0)
        scen["Z"] = zt.cpu().numpy()
    dev = torch.device("cuda", local_rank)
    sums = torch.zeros(2, dtype=torch.float64, device=dev)     # this shard's {sum w, sum w^2}; all-reduced in place when N > 1
    f = pkg.RBPHDFilter(n_local, device_id=local_rank, gm_capacity=CAP, **(dict(model=pkg.capi.MODEL_VICTORIAPARK_3D) if vp else {}))
    sc.load_scenario(f, scen)
    stream = torch.cuda.Stream()
    f.set_stream(stream.cuda_stream)          # engine kernels, the RCCL all-reduce and the timing events order on this stream
    f.bind_weight_sums_buffer(sums.data_ptr())
    f.save_state()
    Z = scen["Z"]
    sums_ptr = sums.data_ptr()

    # N > 1: the one collective of the path -- the all-reduce of {sum w, sum w^2}, 2 doubles over xGMI -- runs on a side stream BESIDE the
    # next step's kernel: the post kernel of step k + 1 divides the weights by step k's total (rfsgpu_step_async_deferred) instead
    # of a divide kernel behind the collective at the end of step k (round 4: 21 us per step with one rank).  RFS_BENCH_INLINE_COLLECTIVE=1
    # restores the round-4 order (A/B).
    deferred = multi and os.environ.get("RFS_BENCH_INLINE_COLLECTIVE") != "1"
    if deferred:
        side = torch.cuda.Stream()
        tot = torch.ones(2, dtype=torch.float64, device=dev)
        ev_post, ev_tot = torch.cuda.Event(), torch.cuda.Event()
        ev_post.record(stream); ev_tot.record(side)          # (creates the handles)
        pend = {"have": False}

        # hand-over between the step's stream and the collective's: device sequence numbers (two words, no event on the step's stream), or
        # -- RFS_BENCH_COLLECTIVE=events, and by itself when the first steps below report a protocol time-out -- stream events
        mode = {"events": os.environ.get("RFS_BENCH_COLLECTIVE") == "events", "fell_back": False}
        # rfsgpu_collective_probe (round 6): the sequence-number form needs the two streams to make progress side by side; every rank plays
        # the hand-over once with nothing at stake and the ranks agree on the minimum (they must issue the same collectives)
        probe_note = "not probed (RFS_BENCH_COLLECTIVE=events)"
        if not mode["events"]:
            stream.synchronize()
            okp = torch.tensor([1 if f.collective_probe(side.cuda_stream) else 0], dtype=torch.int64, device=dev)
            if world > 1:
                dist.all_reduce(okp, op=dist.ReduceOp.MIN)
            probe_note = "side by side on every rank" if int(okp.item()) else "a rank's streams serialise (or its probe timed out): stream events"
            if not int(okp.item()):
                mode["events"], mode["fell_back"] = True, True

        def collective_step(Zk):
            use_events = mode["events"]
            if use_events:
                f.step_async_deferred(Zk, tot.data_ptr() if pend["have"] else None, ev_tot.cuda_event if pend["have"] else None)
                ev_post.record(stream)
            else:      # no event on the step's stream: the post kernel and the side stream's gate kernel meet through two device words
                f.step_async_trailing(Zk, tot.data_ptr(), pend["have"])
            with torch.cuda.stream(side):
                if use_events:
                    side.wait_event(ev_post)
                else:
                    f.collective_gate(side.cuda_stream)
                tot.copy_(sums)
                dist.all_reduce(tot)
                if not use_events and os.environ.get("RFS_BENCH_DROP_PUBLISH") != "1":   # (test hook: without the publish the post kernel's bounded wait runs out)
                    f.collective_publish(side.cuda_stream)
                ev_tot.record(side)
            pend["have"] = True

        def collective_flush():          # what a host does when it needs the normalised weights / N_eff (the resample test)
            if pend["have"]:
                stream.wait_event(ev_tot)
                f.normalize_weights(0.0, tot.data_ptr(), 1)
                pend["have"] = False

    # Re-seeding.  Every step of a `reseed` workload runs on the SAME saved state.  Until round 5 that state was copied back inside the
    # step (rfsgpu_restore_state: 45 MB and 8.8 us of a 124 us step at configs[1] -- benchmark scaffolding, not the path).  Now the inputs of
    # the W warm-up and K timed steps are resident in HBM before the timed region starts: a ring of W + K pre-seeded copies of the state
    # (rfsgpu_state_ring_create), and a step takes the next one by a pointer swap on the host.  RFS_BENCH_RESEED_IN_LOOP=1 (or a ring
    # that would not fit in half of the free memory) keeps the copy in the loop.
    # The untimed steps ahead of it (clock ramp, 0.25 s) go round the same ring and fill it again whenever it has been used up -- so that they
    # too read a state that has left the caches (220 slots x 43 MB), and the kernel average a profiler takes over the whole process is the
    # timed region's.
    ring = {"on": False, "slots": 0, "left": 0}

    def reseed():
        if ring["on"]:
            if ring["left"] == 0:
                f.state_ring_seed()
                ring["left"] = ring["slots"]
            f.state_ring_next()
            ring["left"] -= 1
        else:
            f.restore_state()

    def ring_start(n_slots):
        if not wl["reseed"] or os.environ.get("RFS_BENCH_RESEED_IN_LOOP") == "1":
            return
        slot_bytes = n_local * (11 if vp else 7) * CAP * 8 + n_local * 28
        free_b, _ = torch.cuda.mem_get_info()
        if n_slots * slot_bytes > free_b // 2:
            return
        f.synchronize()
        f.state_ring_create(n_slots)
        ring["on"], ring["slots"], ring["left"] = True, n_slots, n_slots

    def ring_refill():                    # every slot fresh: the next `slots` steps find their inputs resident
        if ring["on"]:
            f.synchronize()
            f.state_ring_seed()
            ring["left"] = ring["slots"]

    def ring_stop():
        if ring["on"]:
            f.synchronize()
            f.state_ring_create(0)
            ring["on"] = False

    if wl["reseed"]:
        def step(k):
            reseed()
            # stream-ordered: the host never waits inside a step; device errors surface at the final sync.  Two launches: the
            # fused step kernel (measurement set in its arguments) and the post kernel (Murty partitions if any, weight sums,
            # and -- one GPU -- the division).
            if deferred:
                collective_step(Z)
                return
            f.step_async(Z, not multi)
            if multi:                 # the only collective on the path: 2 doubles over xGMI
                with torch.cuda.stream(stream):
                    dist.all_reduce(sums)
                f.normalize_weights(0.0, sums_ptr, 1)   # divisor read on the device
    else:
        # C2b: a ring of measurement sets of the same scene (fresh detection noise and clutter), no re-seeding
        rngz = np.random.default_rng(999)
        gt, P = scen["gt"], scen["params"]
        vis = np.nonzero(scen["in_fov"])[0]
        Zring = []
        for _ in range(32):
            det = rngz.choice(vis, min(24, vis.size), replace=False)
            zr = np.hypot(gt[det, 0], gt[det, 1]) + rngz.normal(0, np.sqrt(P["R"][0, 0]), det.size)
            zb = np.arctan2(gt[det, 1], gt[det, 0]) + rngz.normal(0, np.sqrt(P["R"][1, 1]), det.size)
            ncl = N_Z - det.size
            Zk = np.concatenate([np.stack([zr, zb], 1), np.stack([rngz.uniform(P["rmin"], P["rmax"], ncl), rngz.uniform(-np.pi, np.pi, ncl)], 1)], 0)
            Zring.append(np.ascontiguousarray(Zk[rngz.permutation(N_Z)]))

        def step(k):
            # one submission per cycle (rfsgpu_cycle_async): the predict -- births from the previous step's unused measurements at the
            # poses that step used, Sigma += Q -- runs at the head of the fused step kernel; no separate launch, no host wait
            f.cycle_async(True, Zring[k % len(Zring)], normalize=not multi)
            if multi:
                with torch.cuda.stream(stream):
                    dist.all_reduce(sums)
                f.normalize_weights(0.0, sums_ptr, 1)

    if args.pmc_child:                    # the short run the parent profiles with rocprofv3 --pmc (no output, no baselines)
        ring_start(13)
        for k in range(3):
            step(k)
        f.synchronize()                   # (a filter that queues Murty partitions switches to its full post-kernel instance once the host has seen the flag)
        for k in range(3, 13):
            step(k)
        f.synchronize()
        if wl["reseed"]:                  # + the stand-alone likelihood sweep (phase-by-phase update), so that the SQ pass can give it its own issue-slot share
            ring_stop()
            f.set_phase_timing(True)
            for _ in range(6):
                f.restore_state()
                f.update(Z)
            f.set_phase_timing(False)
        return

    ring_start(max(args.warmup + args.steps, 16))
    try:
        for k in range(3):
            step(k)
        f.synchronize()
    except RuntimeError as exc:
        # The sequence-number hand-over needs the step's stream and the collective's stream to make progress side by side (the post kernel
        # waits, bounded, for a word the side stream publishes).  Where a runtime does not give that, the bounded wait raises the
        # protocol flag here, in the first untimed steps: go on with stream events (+8 us per step with one rank) instead of failing the run.
        if not (deferred and not mode["events"]):
            raise
        print(f"[bench rank {rank}] sequence-number hand-over timed out ({exc}); continuing with stream events", file=sys.stderr, flush=True)
        torch.cuda.synchronize()
        try:
            f.synchronize()           # (flags the side stream's gate kernels raised after the first read)
        except RuntimeError:
            pass
        mode["events"], mode["fell_back"] = True, True
        pend["have"] = False
        for k in range(3):
            step(k)
        f.synchronize()
    # shapes for the algorithmic byte counts (one instrumented step, untimed)
    if wl["reseed"]:
        f.restore_state()
    nM = int(f.gm_sizes().sum())
    f.update_map(Z)
    nAfter = int(f.gm_sizes().sum())
    f.importance_weighting(); f.merge(); f.prune()
    nKept = int(f.gm_sizes().sum())
    if not wl["reseed"]:
        f.restore_state()
    # Untimed, right before the timed region (the instrumented pass above waits on the host and lets the GPU idle): a fresh
    # process spends its first ~100 ms of GPU work at low clocks (tools/sync_update_bench.py: 260 us per call there, 131
    # afterwards), so steps for 0.25 s first -- `pre_warmup_steps` in the JSON -- and then the W warm-up steps the caller asked for.
    # (the count comes from eight timed steps, agreed over the ranks: every rank must run the same number of collectives)
    t_pw = time.perf_counter()
    for k in range(8):
        step(k)
    f.synchronize()
    t8 = time.perf_counter() - t_pw
    if multi:
        tt = torch.tensor([t8], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t8 = float(tt.item())
    pre_warm = 8 + 8 * int(min(1024, max(0.0, 0.25 / max(t8 / 8, 1e-6)) // 8))
    for k in range(8, pre_warm):
        step(k)
    f.synchronize()
    ring_refill()                         # the W + K steps from here on find their inputs resident (see `reseed` above)
    # (host-side bookkeeping BEFORE the warm-up steps: nothing but a synchronisation and three cheap calls may sit between the warm-up and
    #  the timed region -- a GPU left idle for a millisecond starts the timed region at lower clocks, which a 20-step run shows)
    bytes_sweep, bytes_step = survey_bytes(n_local, nM, nAfter - nM, nKept, N_Z, BG, DZ)
    dbytes = dict(zip(kernels, design_bytes(n_local, nM, nAfter - nM, nKept, N_Z, BG).values()))
    for k in range(args.warmup):
        step(pre_warm + k)

    f.synchronize()
    f.kernel_time_stats()            # discard the warm-up statistics
    # The kernel durations behind `roofline` come from HIP events on the engine's stream inside the timed region.  Three event
    # records per step cost a C2a step 8 us of its 144 (each is a marker packet the queue drains before the next kernel
    # starts), so they ride on a few steps only -- two of K < 64 steps, six of a longer run (every 8th until round 6: 1 us per step of
    # the default run, 1.6 of a 20-step one -- tools/fixed_cost_probe.py: a timed region's own fixed cost is ~10 us, the rest of the
    # difference between a 20-step and a 200-step figure was these records); the statistics average over those, and the kernel's
    # spread from launch to launch is ~1 %.
    timing_stride = max(1, -(-args.steps // (2 if args.steps < 64 else 6)))
    f.set_step_timing_stride(timing_stride)
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    per_step = []
    for k in range(args.steps):
        step(k)
        if not wl["reseed"]:         # (C2b's predict_map syncs anyway: per-step wall times for the median)
            per_step.append(time.perf_counter())
    if deferred:
        collective_flush()           # the last step's total: wait for its collective, divide (inside the timed region)
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    dt = time.perf_counter() - t0
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    f.synchronize()                  # raises if any step overflowed / hit an unsupported case
    ka, n_timed = f.kernel_time_stats()    # HIP-event pairs recorded on the engine's stream inside the timed region (every timing_stride-th step)
    kern_ms = np.array(ka) / 1e6
    post_ms = f.post_kernel_avg_ns() / 1e6     # the step's post kernel (Murty-200 partitions when queued, weight sums, division)
    ms_per_step = dt / args.steps * 1e3
    wsum = float(f.get_weights().sum())
    assert np.isfinite(wsum) and (multi or abs(wsum - 1.0) < 1e-6), "weights did not normalise"
    fused = kern_ms[1] == 0.0 and kern_ms[2] == 0.0

    # weak-scaling reference on the SAME workload: this rank's shard alone, without the collective (N > 1 only)
    solo = None
    if multi and wl["reseed"]:
        ring_refill()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(args.steps):
            reseed(); f.step_async(Z, True)
        torch.cuda.synchronize()
        st = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cuda")
        dist.all_reduce(st, op=dist.ReduceOp.MAX)
        solo = args.steps / float(st.item())
        f.synchronize(); f.kernel_time_stats()
    f.set_step_timing_stride(1)
    ring_slots = ring["slots"] if ring["on"] else 0
    ring_stop()

    # per-phase breakdown (and the likelihood-sweep rate the north star asks for): the three stand-alone kernels, HIP events,
    # untimed pass after the region
    phase_ms = kern_ms
    if fused and wl["reseed"]:
        acc = np.zeros(3)
        reps = 20
        f.set_phase_timing(True)      # (update() as three launches, each with its own event pair)
        for r in range(reps + 3):
            f.restore_state()
            f.update(Z)
            if r >= 3:
                acc += np.array(f.last_kernel_ns()[:3], dtype=np.float64)
        phase_ms = acc / reps / 1e6
        f.set_phase_timing(False)

    if rank == 0:
        per_kernel = {}
        for k, name in enumerate(kernels):
            if phase_ms[k] > 0:
                per_kernel[name] = dict(ms=round(float(phase_ms[k]), 5), design_bytes=int(dbytes[name]),
                                        design_GBps=round(dbytes[name] / (phase_ms[k] * 1e-3) / 1e9, 2))
        sweep_ms = float(phase_ms[0]) if phase_ms[0] > 0 and not (fused and not wl["reseed"]) else None
        sweep = None
        if sweep_ms:
            g = bytes_sweep / (sweep_ms * 1e-3) / 1e9
            sweep = dict(kernel=("vp_update_map_kernel" if vp else "phd_update_map_block_kernel") + " (stand-alone form, untimed pass)", ms=round(sweep_ms, 5), algorithmic_bytes=int(bytes_sweep),
                         achieved_GBps=round(g, 2), frac=round(g / HBM_PEAK_GBS, 6))
        # the dominant kernel: the fused step, except where the Murty-200 post kernel outweighs it (configs[4])
        murty_dominant = fused and post_ms > float(kern_ms[0])
        dom_name = "murty_jobs_kernel" if murty_dominant else (fused_name if fused else "update_map+weighting+merge_prune")
        dom_ms = post_ms if murty_dominant else (float(kern_ms[0]) if fused else float(kern_ms.sum()))
        achieved = bytes_step / (dom_ms * 1e-3) / 1e9
        design_total = int(sum(dbytes.values()))
        traffic, traffic_note = None, "skipped (--no-pmc)" if args.no_pmc else None
        if not args.no_pmc and not multi and fused:
            child = ["--workload", wname, "--no-pmc", "--no-cpu-baseline"] + (["--particles", str(n_local)] if args.particles else [])
            tr, err = live_traffic(dom_name, child)
            if tr is None:
                traffic_note = "PMC collection failed: " + str(err)
            else:
                traffic = tr["bytes"]
                traffic_note = (f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, two child runs of this workload in this invocation: "
                                f"avg over {tr['launches']} launches, FETCH_SIZE {tr['fetch_size_kib']} KiB, WRITE_SIZE {tr['write_size_kib']} KiB, "
                                f"bytes = (2*FETCH + WRITE)*1024 (gfx950 correction, MI355X_MICROARCH.md)")
        elif not args.no_pmc:
            traffic_note = "collected at --gpus 1 only"
        issue, issue_note = None, "skipped (--no-pmc)" if args.no_pmc else None
        if not args.no_pmc and not multi and fused:
            issue, err = live_issue(dom_name, child, also=("vp_update_map_kernel" if vp else "phd_update_map_block_kernel"))
            if issue is None:
                issue_note = "SQ counter collection failed: " + str(err)
            else:
                issue_note = ("rocprofv3 --kernel-trace --pmc " + " ".join(SQ_COUNTERS) + ", one child run of this workload in this invocation; "
                              "valu_issue_frac = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * SQ_BUSY_CYCLES / 32 SEs)")
        elif not args.no_pmc:
            issue_note = "collected at --gpus 1 only"
        sweep_issue = None
        if issue and issue.get("also"):
            sweep_issue = list(issue.pop("also").values())[0]
        if sweep is not None:
            sweep["valu_issue"] = sweep_issue     # the stand-alone sweep kernel's own issue-slot share, from the SAME SQ pass (its launches are the untimed phase pass of the child run)
            sweep["valu_issue_frac"] = sweep_issue["valu_issue_frac"] if sweep_issue else None
        nE_cfg = int(f.get_filter_config().importanceWeightingEvalPointCount)
        flops = survey_flops_totals(n_local, nM, nAfter - nM, N_Z, nE_cfg if nE_cfg >= 0 else 1 << 30)
        tflops = flops["total"] / (dom_ms * 1e-3) / 1e12
        out = {
            "metric": "PHD filter-update steps/sec",
            # whole-job aggregate: every rank completes `steps` updates of its own shard in `dt` (weak scaling: the filter grows
            # with the GPUs); at N=1 this is the plain filter-update rate
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
                "workload": wl["label"].format(n=n_local),
                "workload_key": wname,
                "particles_total": n_local * world,
                "gm_capacity": CAP,
                "pre_warmup_steps": pre_warm,      # untimed clock-ramp steps (0.25 s) ahead of the W warm-up steps
                "state_reseed": (("ring: the input state of each of the W + K steps is a pre-seeded copy resident in HBM before the timed region "
                                  "(%d slots, rfsgpu_state_ring_*); a step takes it by a pointer swap" % ring_slots) if ring_slots
                                 else ("in the loop: rfsgpu_restore_state copies the saved state back inside every step" if wl["reseed"] else "none (steady state)")),
                "unit_definition": "one step = one update(Z) of one shard of %d particles; value sums the shard-steps of all ranks "
                                   "(global filter of %d particles: %.3f updates/s)" % (n_local, n_local * world, args.steps / dt),
                "parallelism": f"particle-sharded: {world} GPU(s), one process per GPU, RCCL all-reduce of 2 doubles/step" +
                               (" on a side stream beside the next step's kernel; the weights are divided by the previous step's total inside the post kernel "
                                "(%s), the last total is applied before the timed region ends" % ("rfsgpu_step_async_deferred, stream events" + (": fell back from the sequence-number form" if mode["fell_back"] else "")
                                                                                                           if mode["events"] else "rfsgpu_step_async_trailing + rfsgpu_collective_gate / _publish, device sequence numbers")
                                if deferred else " on the engine's stream"),
                "gm_before": nM // n_local, "gm_after_update": nAfter // n_local, "gm_after_prune": nKept // n_local,
                "kernels": {(fused_name if fused else "three_kernels"): dict(ms=round(float(kern_ms[0]) if fused else float(kern_ms.sum()), 5),
                                                                               survey_bytes_step=int(bytes_step), design_bytes=design_total),
                            "post_kernel(murty_jobs_kernel: Murty-200 partitions if queued, weight sums, division)": dict(ms=round(post_ms, 5)),
                            "standalone_phases_untimed_pass": per_kernel},
                "likelihood_sweep": sweep,
            },
            "roofline": {"bound": ("latency" if murty_dominant else "valu_issue"), "achieved_peak_frac_are": "hbm", "kernel": dom_name,
                         # what actually binds: the HBM fraction below is what SURVEY 8(d) asks for and stays the schema's
                         # achieved / peak / frac; the dominant kernels of this path are bound by fp64 vector issue slots and dependent-chain
                         # latency (2 x 2 / 3 x 3 algebra per Gaussian, no MFMA shape), which valu_issue_frac measures
                         "bound_definition": ("latency: a serial best-first search per partition, one assignment solve per pop" if murty_dominant else
                                              "valu_issue: fp64 vector ALU issue slots + dependent-chain latency; `bound` names what binds the kernel (VERDICT r4 item 6), "
                                              "achieved / peak / frac / traffic stay the HBM figures SURVEY 8(d) defines"),
                         "valu_issue_frac": issue["valu_issue_frac"] if issue else None,
                         "valu_issue": issue, "valu_issue_source": issue_note,
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         # the same launch in flops -- SURVEY 8(d)'s algorithmic fp64 flops / the kernel's duration / the fp64 vector peak
                         "flop_frac": round(tflops / FP64_VECTOR_PEAK_TFLOPS, 5), "achieved_tflops_fp64": round(tflops, 3), "peak_tflops_fp64_vector": FP64_VECTOR_PEAK_TFLOPS,
                         "algorithmic_flops": {k: int(v) for k, v in flops.items()},
                         "flop_definition": "SURVEY 8(d): 70 flop per (landmark, measurement) pair + 120 per landmark (sweep), 15 N^2 / 2 (merge, N = Gaussians after the "
                                            "map update), 25 per (evaluation point, Gaussian) evaluation (weighting); 2-D constants",
                         "algorithmic_bytes": int(bytes_step),
                         "achieved_definition": "SURVEY 8(d) bytes_step of one launch (sum over its particles) / the kernel's average HIP-event "
                                                "duration on the engine's stream inside the timed region",
                         "kernel_timing": "HIP events on every %s step of the timed region (%d launches averaged); event records are marker packets "
                                          "that cost a step ~8 us when every step carries them" % ("%dth" % timing_stride if timing_stride > 1 else "", int(n_timed)),
                         "traffic_source": traffic_note,
                         "traffic_over_algorithmic": round(traffic / bytes_step, 3) if traffic else None,
                         "design_bytes": design_total,
                         "design_bytes_note": "what this design's phases move by construction (w_prev plane + the weighting and merge phases' own "
                                              "reads of the mixture); NOT used for achieved/frac",
                         "likelihood_sweep_frac": sweep["frac"] if sweep else None,
                         "likelihood_sweep_GBps": sweep["achieved_GBps"] if sweep else None,
                         "bound_note": ("the dominant kernel of this workload is the Murty-200 assignment search (serial per partition, fp64 dependency "
                                        "chains: latency-bound, not a streaming kernel); the HBM figure is reported for the record, the quantity to "
                                        "watch is its duration.  Round 5: the ranked enumeration ends once no later term can change the sum (the same sum "
                                        "bit for bit).  Round 6: a child of an expansion is solved by ONE augmentation from its parent's dual variables instead of "
                                        "a solve from scratch (hungarian_warm_wave; the k best scores are what the path sums), five waves per job, two peeked heap "
                                        "positions; the search wave's own bookkeeping is what a pop costs now.  Its PMC traffic is the node pool (with a node's row "
                                        "duals since round 6) and the results (DESIGN.md section 8)") if murty_dominant else None},
        }
        # what the run was, as FIELDS (so that a SCALE record can show that RCCL saw N distinct devices)
        out["config"]["distributed"] = dict(
            world_size=world, torch_distributed_initialised=bool(multi), backend=(dist.get_backend() if multi else None),
            device_pci_bus_ids=pci_ids, distinct_devices=len(set(pci_ids)),
            collective=(None if not multi else ("inline" if not deferred else ("events" if mode["events"] else "sequence_numbers"))),
            collective_fell_back=(bool(mode["fell_back"]) if deferred else False),
            collective_probe=(probe_note if deferred else None))
        if per_step:
            d = np.diff(np.array([t0] + per_step)) * 1e3
            out["config"]["ms_per_step_median"] = round(float(np.median(d)), 5)
            out["config"]["ms_per_step_p90"] = round(float(np.percentile(d, 90)), 5)
        if solo is not None:
            out["config"]["same_workload_single_shard_steps_per_s"] = round(solo, 3)
        if not args.no_cpu_baseline and not multi:      # (rank 0 at N = 1 only)
            out["cpu_baseline"] = cpu_baseline(wname, n_local, args.particles)
        if not args.no_boundary and not multi and not vp and wl["reseed"]:
            try:
                out["boundary"] = boundary_cost(pkg, sc, wl, n_local, CAP, scen, local_rank)
            except Exception as e:   # noqa: BLE001  (a side figure must not cost the line)
                out["boundary"] = dict(error=repr(e)[:300])
    else:
        out = None

    def finish(migr):
        if rank == 0:
            if migr is not None:
                out["config"]["resample_migration"] = migr
            print(json.dumps(out), flush=True)

    # One global resampling step with cross-shard migration (N > 1), timed on its own -- LAST, and under a watchdog: it is a side
    # figure, and neither an exception in it nor a collective that never completes may cost the line above.
    migr = None
    if multi and wl["reseed"]:
        import threading
        dist.barrier()
        finished = threading.Event()

        def bail():
            if finished.is_set():
                return
            finish(dict(error="no result within 120 s (a collective of this side measurement did not complete); the numbers of the timed region are unaffected"))
            os._exit(0)
        timer = threading.Timer(120.0, bail)
        timer.daemon = True
        timer.start()
        try:
            migr = pkg.sharded.bench_resample_migration(pkg, f, rank, world, dev, stream=stream, sums=sums, reps=3)
        except Exception as e:   # noqa: BLE001
            migr = dict(error=repr(e)[:300])
        finished.set()
        timer.cancel()
    finish(migr)
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
