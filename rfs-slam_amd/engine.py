"""Loader for librfsgpu.so + the host-side mirror of rfs::RBPHDFilter for the device path.

`RBPHDFilter` mirrors the reference class's public members for the hot path
(include/RBPHDFilter.hpp:72-251): predict-side map ops, update(), getGMSize(), getLandmark(),
getTimingInfo(), public `config`; resampling follows ParticleFilter::resample
(include/ParticleFilter.hpp:399-492) on the host exactly like the reference (host RNG), and hands
the copy plan to the device through rfsgpu_resample_apply.
"""
import ctypes as C
import os
import numpy as np

from . import capi
from .build import LIB

_lib = None


def load_library():
    """dlopen the in-tree HIP extension.  Fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise RuntimeError(
                f"{LIB} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
                "There is no CPU fallback for the device path.")
        _lib = C.CDLL(LIB)
        _lib.rfsgpu_abi_version.restype = C.c_int
    return _lib


def mat_perm(A, device_id=0):
    return capi.mat_perm(load_library(), "rfsgpu_", A, device_id)


def FilterGroup(n_particles, device_ids, gm_capacity=512, model=capi.MODEL_RNGBRG_2D):
    """One filter over several GPUs from this process (rfsgpu_group_*): contiguous particle blocks, one shard per device id."""
    return capi.Group(load_library(), n_particles, device_ids, gm_capacity=gm_capacity, model=model)


class RBPHDFilter(capi.CFilter):
    """Device-resident RB-PHD filter shard (one GPU).  Mirrors rfs::RBPHDFilter for the update path."""

    def __init__(self, n_particles, device_id=0, gm_capacity=512, model=capi.MODEL_RNGBRG_2D, max_particles=None):
        super().__init__(load_library(), "rfsgpu_", n_particles, model=model, device_id=device_id, gm_capacity=gm_capacity,
                         max_particles=max_particles)
        self.config = self.default_filter_config()
        self.n_init = n_particles
        self.effNParticles_t = n_particles / 4.0  # ParticleFilter.hpp:232
        self.effNParticles_t_percent = self.effNParticles_t / n_particles
        self.nUpdatesSinceResample = 0
        self.nMeasurementsSinceResample = 0
        self.resampleOccured = False
        self.last_resample_plan = None    # src slot per slot of the last resampling that fired (None when the last call did not)

    # ParticleFilter::setEffectiveParticleCountThreshold (ParticleFilter.hpp:386-391)
    def setEffectiveParticleCountThreshold(self, t):
        self.effNParticles_t = float(t)
        self.effNParticles_t_percent = float(t) / self.n   # the particle count at the time of the call (:390)

    def apply_config(self):
        self.set_filter_config(self.config)

    def last_kernel_ns(self):
        ns = (C.c_longlong * 4)()
        self._call("last_kernel_ns", ns)
        return list(ns)

    def last_step_variant(self):
        """{waves per particle, phase priorities, merge grid log2, fused} of the last stream-ordered step."""
        out = (C.c_int * 4)()
        self._call("last_step_variant", out)
        return tuple(out)

    def update_async(self, Z):
        """Stream-ordered update: no host sync; errors surface at synchronize()."""
        Z, n = self._z(Z)
        self._call("update_async", self._ptr(Z), n)

    def step_async(self, Z, normalize=True):
        """update_async + the weight reduction (and, with normalize, the division) in the step's post kernel."""
        Z, n = self._z(Z)
        self._call("step_async", self._ptr(Z), n, C.c_int(1 if normalize else 0))

    def step_async_deferred(self, Z, prev_total_ptr=None, wait_event=None):
        """rfsgpu_step_async_deferred: the step of a sharded host whose normalisation trails by one step (the post kernel divides by
        the previous step's all-reduced total, device pointer, and waits for `wait_event` -- a hipEvent_t handle -- just before it)."""
        Z, n = self._z(Z)
        self._call("step_async_deferred", self._ptr(Z), n, C.c_void_p(prev_total_ptr), C.c_void_p(wait_event))

    def step_async_trailing(self, Z, total_ptr, have_prev):
        """rfsgpu_step_async_trailing: the deferred step without stream events (pair with collective_gate / collective_publish on the side stream)."""
        Z, n = self._z(Z)
        self._call("step_async_trailing", self._ptr(Z), n, C.c_void_p(total_ptr), C.c_int(1 if have_prev else 0))

    def collective_gate(self, hip_stream):
        self._call("collective_gate", C.c_void_p(hip_stream))

    def collective_publish(self, hip_stream):
        self._call("collective_publish", C.c_void_p(hip_stream))

    def collective_probe(self, hip_stream):
        """rfsgpu_collective_probe: do the engine's stream and `hip_stream` make progress side by side (what the sequence-number
        hand-over needs)?  Synchronises both streams."""
        ok = C.c_int(0)
        self._call("collective_probe", C.c_void_p(hip_stream), C.byref(ok))
        return bool(ok.value)

    def _opt(self, a, shape=None):
        if a is None:
            return None, C.c_void_p(None)
        a = np.ascontiguousarray(a, dtype=np.float64)
        if shape is not None:
            assert a.shape == shape, (a.shape, shape)
        return a, self._ptr(a)

    def cycle_async(self, predict, Z, poses=None, pose_cov=None, weights=None, normalize=True):
        """One submission per predict + update cycle (rfsgpu_cycle_async): predict = None (no predict part), False (static step
        only) or True (births + static step); poses / pose_cov / weights = the host's new inputs or None (unchanged)."""
        Z, n = self._z(Z)
        x, px = self._opt(poses, (self.n, 3))
        stride = 0
        if pose_cov is not None:
            pose_cov = np.ascontiguousarray(pose_cov, dtype=np.float64)
            assert pose_cov.size in (9, 9 * self.n)
            stride = 0 if pose_cov.size == 9 else 9
        cv, pc = self._opt(pose_cov)
        w, pw = self._opt(weights, (self.n,))
        self._call("cycle_async", C.c_int(-1 if predict is None else (1 if predict else 0)), px, pc, C.c_int(stride), pw, self._ptr(Z), n,
                   C.c_int(1 if normalize else 0))

    def update_io(self, Z, predict=None, poses=None, pose_cov=None, weights=None, want_weights=True):
        """RBPHDFilter::update with its inputs and outputs in one synchronous call (rfsgpu_update_io); returns the updated
        (un-normalised) particle weights."""
        Z, n = self._z(Z)
        x, px = self._opt(poses, (self.n, 3))
        stride = 0
        if pose_cov is not None:
            pose_cov = np.ascontiguousarray(pose_cov, dtype=np.float64)
            assert pose_cov.size in (9, 9 * self.n)
            stride = 0 if pose_cov.size == 9 else 9
        cv, pc = self._opt(pose_cov)
        w, pw = self._opt(weights, (self.n,))
        out = np.empty(self.n, dtype=np.float64) if want_weights else None
        self._call("update_io", C.c_int(-1 if predict is None else (1 if predict else 0)), px, pc, C.c_int(stride), pw, self._ptr(Z), n,
                   self._ptr(out) if want_weights else C.c_void_p(None))
        return out

    def kernel_time_stats(self):
        avg = (C.c_double * 3)()
        n = C.c_int()
        self._call("kernel_time_stats", avg, C.byref(n))
        return list(avg), n.value

    def post_kernel_avg_ns(self):
        """Average duration of the step's post kernel over the fused steps the last kernel_time_stats() call covered."""
        fn = self._fn("post_kernel_avg_ns")
        fn.restype = C.c_double
        return float(fn(self._h))

    def step_launch_order(self, order=None, mode=None, want_costs=True):
        """rfsgpu_step_launch_order: the last Victoria Park step's per-particle durations (ticks) out; launch order in: `order` (slot ->
        particle, frozen: mode 1), or mode 0 (slot == particle) / mode 2 (re-sorted by every step's post kernel, the default)."""
        cost = np.zeros(self.n, dtype=np.float32) if want_costs else None
        o = None if order is None else np.ascontiguousarray(order, dtype=np.int32)
        m = 1 if order is not None else (2 if mode is None else int(mode))
        self._call("step_launch_order", C.c_int(m), None if o is None else o.ctypes.data_as(C.c_void_p),
                   None if cost is None else cost.ctypes.data_as(C.c_void_p))
        return cost

    def set_step_timing_stride(self, every):
        """HIP events (kernel_time_stats / post_kernel_avg_ns) on every `every`-th fused stream-ordered step only."""
        self._call("set_step_timing_stride", C.c_int(int(every)))

    def weight_sums_async(self):
        self._call("weight_sums_async")

    def weight_sums_device_ptr(self):
        fn = self._fn("weight_sums_device_ptr")
        fn.restype = C.c_void_p
        return fn(self._h)

    def set_stream(self, hip_stream):
        self._call("set_stream", C.c_void_p(hip_stream))

    def bind_weight_sums_buffer(self, dev_ptr):
        self._call("bind_weight_sums_buffer", C.c_void_p(dev_ptr))

    def save_state(self):
        self._call("save_state")

    def restore_state(self):
        self._call("restore_state")

    def state_ring_create(self, n_slots):
        """n_slots pre-seeded copies of the saved state (bench.py: the inputs of the timed steps are resident before the timed region)."""
        self._call("state_ring_create", C.c_int(int(n_slots)))

    def state_ring_seed(self):
        self._call("state_ring_seed")

    def state_ring_next(self):
        self._call("state_ring_next")

    def stream(self):
        fn = self._fn("stream")
        fn.restype = C.c_void_p
        return fn(self._h)

    # RBPHDFilter::update (RBPHDFilter.hpp:444-541) including the resample-or-normalise tail.
    def update_and_resample(self, Z, u01_fn=np.random.random):
        self.apply_config()
        self.nUpdatesSinceResample += 1
        Z = np.asarray(Z, dtype=np.float64).reshape(-1, self.dz)
        if Z.shape[0] == 0:
            return False
        self.nMeasurementsSinceResample += Z.shape[0]
        self.update(Z)
        self.resampleOccured = False
        if (self.nUpdatesSinceResample >= self.config.minUpdatesBeforeResample and
                self.nMeasurementsSinceResample >= self.config.minMeasurementsBeforeResample):
            self.resampleOccured = self.resample(u01_fn)
        if self.resampleOccured:
            self.nUpdatesSinceResample = 0
            self.nMeasurementsSinceResample = 0
        else:
            s = self.weight_sums()
            self.normalize_weights(s[0])
        return self.resampleOccured

    # ParticleFilter::resample (ParticleFilter.hpp:399-492): host logic on the N weights.
    # resample(n, forceResample): n == 0 or n > nParticles_ keeps the count (:417-418); a smaller n shrinks the particle set
    # (FastSLAM::resampleWithMapCopy after multi-hypothesis growth).  The plan of the last resampling stays in
    # `last_resample_plan` so that a caller holding per-particle host data (poses) can apply the same copies.
    def resample(self, u01_fn=np.random.random, n_out=0, force=False):
        self.last_resample_plan = None
        s = self.weight_sums()
        self.normalize_weights(s[0])
        w = self.get_weights()
        n = self.n
        if not force:
            neff = 1.0 / float(np.sum(w * w))
            if neff > self.effNParticles_t and neff / n > self.effNParticles_t_percent:
                return False
        if n_out == 0 or n_out > n:
            n_out = n
        src = systematic_resample_plan(w, float(u01_fn()), n_out)
        self.resample_apply(src, n_out)
        self.last_resample_plan = src
        return True


class FastSLAM(RBPHDFilter):
    """rfs::FastSLAM (include/FastSLAM.hpp) for the 2-D range-bearing model on the same engine: the handle's mixtures are
    the landmark maps (weights = log-odds of existence).  With fs_config.maxNDataAssocHypotheses > 1 (MH-FastSLAM) an update
    multiplies particles (one per kept association hypothesis, :462-476): build the filter with `max_hypotheses` so that the
    handle has room for nParticlesMax * max_hypotheses particles; `parents` (slot -> the particle it was copied from in the
    last update) and `last_resample_plan` let a caller holding poses follow the copies."""

    def __init__(self, n_particles, device_id=0, gm_capacity=512, max_hypotheses=1, n_particles_max=None):
        n_max = 3 * n_particles if n_particles_max is None else int(n_particles_max)    # FastSLAM.hpp:250
        cap = max(n_particles, n_max) * max(1, int(max_hypotheses)) if max_hypotheses > 1 else None
        super().__init__(n_particles, device_id=device_id, gm_capacity=gm_capacity, max_particles=cap)
        self.fs_config = self.default_fastslam_config()
        self.fs_config.nParticlesMax = n_max
        self.fs_config.maxNDataAssocHypotheses = max(1, int(max_hypotheses))
        self.parents = np.arange(n_particles, dtype=np.int32)
        self.last_resample_plan = None

    # FastSLAM::predict (:362-385), map part: staticStep on every landmark, no births
    def predict_map(self, add_birth=False):
        super().predict_map(False)

    # FastSLAM::update (:387-421) + resampleWithMapCopy (:708-735)
    def update_and_resample(self, Z, u01_fn=np.random.random):
        self.apply_config()
        self.set_fastslam_config(self.fs_config)
        self.nUpdatesSinceResample += 1
        Z = np.asarray(Z, dtype=np.float64).reshape(-1, self.dz)
        if Z.shape[0] == 0:
            return False
        self.nMeasurementsSinceResample += Z.shape[0]
        self.fastslam_update(Z)                               # the particle count may have grown (self.n)
        self.parents = self.particle_parents()
        self.resampleOccured = False
        self.last_resample_plan = None
        if self.n > self.fs_config.nParticlesMax:             # :711-712 forced, back to the initial count
            self.resampleOccured = self.resample(u01_fn, self.n_init, True)
        elif (self.nUpdatesSinceResample >= self.fs_config.minUpdatesBeforeResample and
                self.nMeasurementsSinceResample >= self.fs_config.minMeasurementsBeforeResample):
            self.resampleOccured = self.resample(u01_fn, self.n_init)   # landmark candidates travel with their particle
        self.fastslam_set_resample_occured(self.resampleOccured)
        if self.resampleOccured:
            self.nUpdatesSinceResample = 0
            self.nMeasurementsSinceResample = 0
        else:
            s = self.weight_sums()
            self.normalize_weights(s[0])
        return self.resampleOccured


def systematic_resample_plan(w, u01, n_out=None):
    """Systematic sampling + slot assignment of ParticleFilter::resample (ParticleFilter.hpp:419-479).
    Returns src_slot[k]: which (kept-in-place) particle slot k copies; k itself when it is kept.
    n_out < len(w): resample(n) as FastSLAM::resampleWithMapCopy calls it -- n_out samples from all particles, kept or
    copied into the first n_out slots (cases 1-4 of :459-478); the returned plan has n_out entries."""
    if n_out is not None and n_out < w.size:
        return _systematic_resample_plan_shrink(w, u01, int(n_out))
    w = np.asarray(w, dtype=np.float64)
    n = w.size
    if n >= 256 and not (w < 0).any() and np.isfinite(w).all():
        return _systematic_resample_plan_vectorised(w, u01)
    return _systematic_resample_plan_loop(w, u01)


def _systematic_resample_plan_vectorised(w, u01):
    """The same plan without Python loops (20 000 particles at configs[2]).  np.cumsum adds sequentially in index order, i.e.
    the reference's `cumulative_weight += w[idx]` and `sample_point += sample_interval` roundings exactly; with non-negative
    weights the running sum is non-decreasing, so `while (sample_point > cumulative && idx < n-1) idx++` is a search."""
    n = w.size
    interval = 1.0 / float(n)
    cum = np.cumsum(w)
    steps = np.full(n, interval)
    steps[0] = interval * u01
    sp = np.cumsum(steps)
    sampled_idx = np.minimum(np.searchsorted(cum, sp, side="left"), n - 1)
    sampled_idx = np.maximum.accumulate(sampled_idx)      # (idx never moves back; a no-op for monotone sums)
    sampled = np.zeros(n, dtype=bool)
    sampled[sampled_idx] = True
    first = np.ones(n, dtype=bool)
    first[1:] = sampled_idx[1:] != sampled_idx[:-1]
    dups = sampled_idx[~first]                             # sources of the copies, in sampling order
    free = np.nonzero(~sampled)[0]                         # un-sampled slots, ascending: they take the copies in order
    src = np.arange(n, dtype=np.int32)
    src[free[:dups.size]] = dups
    return src


def _systematic_resample_plan_loop(w, u01):
    n = w.size
    interval = 1.0 / float(n)
    sample_point = interval * u01
    idx = 0
    cumulative = w[0]
    sampled = np.zeros(n, dtype=bool)
    sampled_idx = np.zeros(n, dtype=np.int64)
    for i in range(n):
        while sample_point > cumulative and idx < n - 1:
            idx += 1
            cumulative += w[idx]
        sampled_idx[i] = idx
        sampled[idx] = True
        sample_point += interval
    src = np.arange(n, dtype=np.int32)
    nxt = 0
    prev = -1
    for i in range(n):
        idx = int(sampled_idx[i])
        first = not (i > 0 and idx == prev)
        prev = idx
        if first:
            continue
        while nxt < n and sampled[nxt]:
            nxt += 1
        src[nxt] = idx
        nxt += 1
    return src


def _systematic_resample_plan_shrink(w, u01, n):
    N = w.size
    interval = 1.0 / float(n)
    sample_point = interval * u01
    idx = 0
    cumulative = w[0]
    sampled = np.zeros(N, dtype=bool)
    sampled_idx = np.zeros(n, dtype=np.int64)
    for i in range(n):
        while sample_point > cumulative and idx < N - 1:
            idx += 1
            cumulative += w[idx]
        sampled_idx[i] = idx
        sampled[idx] = True
        sample_point += interval
    src = np.arange(n, dtype=np.int32)
    nxt = 0
    prev = -1
    for i in range(n):
        idx = int(sampled_idx[i])
        first = not (i > 0 and idx == prev)
        prev = idx
        if idx < n and first:          # case 1: stays where it is
            continue
        while nxt < N and sampled[nxt]:
            nxt += 1
        assert nxt < n, "copies always land below the new count"
        src[nxt] = idx
        nxt += 1
    return src
