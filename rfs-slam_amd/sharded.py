"""Multi-GPU host for the RB-PHD update path: one process per GPU, particles sharded in contiguous blocks.

Every phase of RBPHDFilter::update() before resampling is independent per particle (reference
include/RBPHDFilter.hpp:469-520), so the only collective on the per-step path is the all-reduce (sum) of
{sum w, sum w^2} -- 2 doubles over RCCL/xGMI, written by the engine into a device tensor and consumed on the device -- for
ParticleFilter::normalizeWeights / the N_eff test (include/ParticleFilter.hpp:352-363, 405-415).  When resampling fires, the
reference's GLOBAL systematic resampling (ParticleFilter.hpp:419-479) is kept: the N weights are all-gathered (device
tensors), every rank computes the same plan from the same uniform draw, local children are a device gather
(rfsgpu_resample_apply), cross-shard children migrate as packed rows from device memory to device memory
(rfsgpu_export_slab_rows -> one RCCL all-to-all with per-peer byte counts -> rfsgpu_import_slab_rows): the host only ever
handles slot indices.
Shard-local resampling would change results and is not offered.

Backend-agnostic: `local` is any object with the filter interface of capi.CFilter (the device engine in production; the
tests substitute the CPU oracle, whose rows live in host memory, to run this host logic under gloo).  With the gloo backend
and device rows (two ranks sharing one GPU in the tests) the rows are staged through the host, because gloo has no
device-to-device transport (point-to-point send/recv there); over RCCL ("nccl") nothing is staged.
"""
import numpy as np
import os

import torch
import torch.distributed as dist

from . import capi
from .engine import systematic_resample_plan


class _DevArray:
    """Zero-copy view of engine-owned device memory for torch (CUDA array interface v2)."""

    def __init__(self, ptr, n, typestr="<f8"):
        self.__cuda_array_interface__ = dict(shape=(n,), typestr=typestr, data=(int(ptr), False), version=2, strides=None)


class ShardedRBPHDFilter:
    """inheritance: "reference" (default) = the reference's birth-state inheritance after a resampling (include/RBPHDFilter.hpp:
    1005-1011) over GLOBAL slots -- ids are global, so this host owns the rule and the local engine runs in
    RFSGPU_INHERIT_EXTERNAL; implemented for immediate-birth configurations (birthGaussianMeasurementCountThreshold == 1, the 2-D
    simulator) through the closed form of the slot-ordered walk over the 8-byte unused masks of all shards, and for configurations
    that keep candidate lists by the walk itself, level by level (predict_map / _predict_levels below).  "eager" = a child takes
    its parent's lists and FOV count at resampling time (rounds 1-2; not the reference's results)."""

    def __init__(self, local, group=None, device=None, stream=None, sums=None, inheritance="reference", defer_normalisation=False):
        assert inheritance in ("reference", "eager")
        self.inheritance = inheritance
        self.f = local
        # the engine handle is switched to the mode this host needs and switched BACK by close(): a handle that outlives the
        # wrapper must not stay in EXTERNAL mode (in which the engine refuses a birth predict after a resampling unless the host
        # has dealt with the inheritance rule: every predict must go through self.predict_map, never through self.f.predict_map)
        self._mode_before = local.get_birth_inheritance() if hasattr(local, "get_birth_inheritance") else None
        local.set_birth_inheritance(capi.INHERIT_EXTERNAL if inheritance == "reference" else capi.INHERIT_EAGER)
        # update(): steps whose resample test is not due may let the normalisation trail by one step (update_deferred: the collective leaves
        # the step's critical path).  OPT-IN: the trailing form multiplies before it divides -- (w L) / T against the
        # reference's (w / T) L, include/RBPHDFilter.hpp:537-539 -- an ulp apart, so the default keeps the sharded path bit-identical
        # to the one-handle path; hosts that want the collective off the critical path (bench.py --gpus N) pass defer_normalisation=True.
        self.defer_normalisation = bool(defer_normalisation)
        self.deferred_check_every = 16     # deferred steps between two looks at the device error word (nothing else waits for the GPU on that path)
        self._deferred_since_check = 0
        self._handover = None              # "sequence_numbers" | "events": decided by a probe at the first deferred step (all ranks agree)
        self.cand_lists_seen = False       # a candidate list has been moved / kept on SOME shard (all-reduced in predict_map)
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        self.n_local = local.n
        self.n_total = self.n_local * self.world
        if device is None:   # the device engine's rows / sums / weights are device memory; any other backend (the CPU oracle) lives on the host
            device = torch.device("cuda", local.device_id) if getattr(local, "_p", "") == "rfsgpu_" else torch.device("cpu")
        self.device = device
        self.on_gpu = self.device.type == "cuda"
        self.effNParticles_t = self.n_total / 4.0          # ParticleFilter.hpp:232
        self.nUpdatesSinceResample = 0
        self.nMeasurementsSinceResample = 0
        self.resampleOccured = False
        self.last_resample_plan = None                     # global slot -> global source slot of the last resampling
        self.last_migration = dict(rows_sent=0, rows_received=0, bytes_sent=0)
        self.pid = np.arange(self.n_total, dtype=np.int64)     # Particle::id_ / idParent_ by GLOBAL slot (the same on every rank)
        self.ppid = np.arange(self.n_total, dtype=np.int64)
        self.stream = None
        self.sums = None
        self._w_view = None
        if self.on_gpu:
            # engine kernels, the RCCL collectives and the row transport all order on ONE stream; the weight sums live in a
            # device tensor that the all-reduce updates in place and normalize_kernel reads: no host round trip per step
            self.stream = stream if stream is not None else torch.cuda.Stream(device=self.device)
            self.f.set_stream(self.stream.cuda_stream)
            # the trailing normalisation (update_deferred): the collective of step k on a side stream beside step k + 1's kernel
            self._side = torch.cuda.Stream(device=self.device)
            self._tot = torch.ones(2, dtype=torch.float64, device=self.device)
            self._ev_post, self._ev_tot = torch.cuda.Event(), torch.cuda.Event()
            self._ev_post.record(self.stream); self._ev_tot.record(self._side)
            self._pending = False
            self.sums = sums if sums is not None else torch.zeros(2, dtype=torch.float64, device=self.device)
            self.f.bind_weight_sums_buffer(self.sums.data_ptr())
            try:
                self._w_view = torch.as_tensor(_DevArray(self.f.weights_device_ptr(), self.n_local), device=self.device)
            except Exception:
                self._w_view = None                        # (fallback: weights through the host, see gather_weights)

    # -- collectives ---------------------------------------------------------------------------------------
    def _stream_ctx(self):
        return torch.cuda.stream(self.stream) if self.on_gpu else _NullCtx()

    def normalize(self, need_totals=False):
        """normalizeWeights over ALL shards.  Device engine: weight sums -> all-reduce in place -> on-device divide, all on
        the engine's stream; the two totals come back to the host only when the caller needs N_eff (16 bytes)."""
        if self.on_gpu:
            self.flush_deferred()
            self.f.weight_sums_async()
            if self.world > 1:
                with self._stream_ctx():
                    dist.all_reduce(self.sums, group=self.group)
            self.f.normalize_weights(0.0, self.sums.data_ptr())   # divisor read on the device
            if not need_totals:
                return None
            with self._stream_ctx():
                tot = self.sums.cpu()
            return tot.numpy()
        s = torch.from_numpy(np.asarray(self.f.weight_sums(), dtype=np.float64))
        if self.world > 1:
            dist.all_reduce(s, group=self.group)
        tot = s.numpy()
        self.f.normalize_weights(float(tot[0]))
        return tot

    # -- trailing normalisation (round 5) -----------------------------------------------------------------
    def update_deferred(self, Z):
        """RBPHDFilter::update's device part for a step after which the host does NOT need the normalised weights or N_eff (the
        resample test is not due: minUpdatesBeforeResample / minMeasurementsBeforeResample, include/RBPHDFilter.hpp:527-531): the
        all-reduce of this step's {sum w, sum w^2} goes on a side stream and the division by its result is done by the NEXT step's
        post kernel (rfsgpu_step_async_deferred) -- the collective leaves the step's critical path.  Any call that needs the
        weights (normalize, resample, gather_weights) first applies the pending total (flush_deferred).  Device engine only."""
        assert self.on_gpu, "the trailing normalisation is a device-path feature (the CPU stand-in normalises in place)"
        Z = np.asarray(Z, dtype=np.float64).reshape(-1, self.f.dz)
        if self._handover is None:
            self._choose_handover()
        if self._handover == "sequence_numbers":
            # (no event on the step's stream: this step's post kernel and the side stream's gate kernel meet through two device words,
            #  rfsgpu_step_async_trailing; the event behind the collective is only waited for when the pending total is applied)
            self.f.step_async_trailing(Z, self._tot.data_ptr(), self._pending)
        else:
            # the fall-back where the two streams do not run side by side: an event record behind the post kernel, an event wait in
            # front of the next one (two packets on the step's stream, ~+9 us per step with one rank)
            self.f.step_async_deferred(Z, self._tot.data_ptr() if self._pending else None, self._ev_tot.cuda_event if self._pending else None)
            self._ev_post.record(self.stream)
        with torch.cuda.stream(self._side):
            if self._handover == "sequence_numbers":
                self.f.collective_gate(self._side.cuda_stream)
            else:
                self._side.wait_event(self._ev_post)
            self._tot.copy_(self.sums)
            if self.world > 1:
                dist.all_reduce(self._tot, group=self.group)
            if self._handover == "sequence_numbers":
                self.f.collective_publish(self._side.cuda_stream)
            self._ev_tot.record(self._side)
        self._pending = True
        # nothing on this path waits for the GPU: look at the device error word (capacity, Murty, the hand-over's time-out) every so often
        self._deferred_since_check += 1
        if self._deferred_since_check >= self.deferred_check_every:
            self._deferred_since_check = 0
            self.f.synchronize()

    def _choose_handover(self):
        """Sequence numbers need the step's stream and the collective's stream to make progress side by side (the post kernel waits, on
        the device, for a word a LATER submission on the side stream publishes): rfsgpu_collective_probe plays that hand-over once with
        nothing at stake.  Every rank probes its own stream pair; the ranks must issue the same collectives, so the verdict is the
        minimum over the ranks.  RFSGPU_SHARDED_HANDOVER=events | sequence_numbers overrides the probe."""
        forced = os.environ.get("RFSGPU_SHARDED_HANDOVER")
        if forced in ("events", "sequence_numbers"):
            self._handover, self.handover_probe = forced, "forced by RFSGPU_SHARDED_HANDOVER"
            return
        self.stream.synchronize()
        ok = 1 if self.f.collective_probe(self._side.cuda_stream) else 0
        if self.world > 1:
            t = torch.tensor([ok], dtype=torch.int64)
            if self.backend == "nccl":
                with self._stream_ctx():
                    td = t.to(self.device)
                    dist.all_reduce(td, op=dist.ReduceOp.MIN, group=self.group)
                    t = td.cpu()
            else:
                dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
            ok = int(t.item())
        self._handover = "sequence_numbers" if ok else "events"
        self.handover_probe = "side by side" if ok else "serialised (or a rank's probe timed out): stream events"

    def get_weights(self):
        """This shard's particle weights, NORMALISED: a total left pending by a deferred step is applied first (reading the engine
        handle directly, self.f.get_weights(), would return them undivided in the meantime)."""
        self.flush_deferred()
        return self.f.get_weights()

    def flush_deferred(self):
        """Apply the total a deferred step left pending: wait for its collective (stream-ordered), divide on the device."""
        if self.on_gpu and self._pending:
            self.stream.wait_event(self._ev_tot)
            self.f.normalize_weights(0.0, self._tot.data_ptr())
            self._pending = False

    def gather_weights(self):
        """All N weights on every rank (host array, for the systematic-resampling plan): all-gather of device tensors."""
        self.flush_deferred()
        if self.on_gpu and self._w_view is not None:
            with self._stream_ctx():
                if self.world == 1:
                    return self._w_view.cpu().numpy()
                out = torch.empty(self.n_total, dtype=torch.float64, device=self.device)
                dist.all_gather_into_tensor(out, self._w_view, group=self.group)
                return out.cpu().numpy()
        w = torch.from_numpy(self.f.get_weights())
        if self.on_gpu:
            w = w.to(self.device)
        if self.world == 1:
            return w.cpu().numpy()
        out = [torch.empty_like(w) for _ in range(self.world)]
        dist.all_gather(out, w, group=self.group)
        return torch.cat(out).cpu().numpy()

    # -- RBPHDFilter::predict, map part (:415-442) ----------------------------------------------------------
    def predict_map(self, add_birth=True):
        """In the predicts that follow a resampling the reference first copies, slot by slot in ascending order,
        unused_measurements_ / birthGaussians_ from SLOT idParent_ in its current state (:1005-1011).  With immediate births no
        candidate list exists and the walk has a closed form over the lists as they are before the predict: own list if
        idParent_ == slot, the parent slot's list if that is a HIGHER slot (not yet visited), nothing if it is a LOWER one
        (already consumed).  One all-gather of N 8-byte masks, only in those predicts."""
        if add_birth and self.resampleOccured and self.inheritance == "reference":
            if np.any(self.ppid != np.arange(self.n_total)):
                cfg = self.f.get_filter_config()
                # ONE predicate for every rank (and the same one rfsgpu_group_predict_map uses): the closed form over the masks is
                # only valid while no shard holds a candidate list -- imported lists on an immediate-birth configuration included
                if cfg.birthGaussianMeasurementCountThreshold == 1 and self.f.dz == 2 and not self._any_shard_has_candidates():
                    m_all = self._gather_masks()
                    lo = self.rank * self.n_local
                    g = np.arange(lo, lo + self.n_local)
                    p = self.ppid[g]
                    new = np.where(p == g, m_all[g], np.where(p > g, m_all[np.clip(p, 0, self.n_total - 1)], np.uint64(0))).astype(np.uint64)
                    if np.any(new != m_all[g]):
                        self.f.set_unused_masks(new)
                else:
                    self._predict_levels(add_birth)
                    return
            # acknowledge to the engine that the rule has been applied (or that no slot has a foreign parent) for this predict
            self.f.set_birth_inheritance(capi.INHERIT_EXTERNAL)
        self.f.predict_map(add_birth)

    def _any_shard_has_candidates(self):
        """Does any shard hold birth candidates (count over its slots > 0)?  All-reduced so that every rank takes the same branch."""
        has = int(bool(self.f.has_birth_candidates()))   # (rfsgpu_has_birth_candidates: a missing symbol fails loudly)
        if self.world > 1:
            t = torch.tensor([has], dtype=torch.int64)
            if self.on_gpu and self.backend == "nccl":
                with self._stream_ctx():
                    td = t.to(self.device)
                    dist.all_reduce(td, op=dist.ReduceOp.MAX, group=self.group)
                    t = td.cpu()
            else:
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            has = int(t.item())
        return bool(has)

    def close(self):
        """Give the engine handle back in the inheritance mode it had before this wrapper took it over."""
        if self.f is not None:
            try:
                self.flush_deferred()
            except Exception:   # noqa: BLE001  (handle already closed)
                pass
        if self._mode_before is not None and self.f is not None:
            try:
                self.f.set_birth_inheritance(self._mode_before)
            except Exception:   # noqa: BLE001  (handle already closed)
                pass
        self._mode_before = None

    def _predict_levels(self, add_birth):
        """Configurations that keep birth-candidate lists (Victoria Park; CountThreshold > 1): the walk in full, level by level over
        GLOBAL slots (csrc/birth.h has the single-GPU form).  Level 0 = slots that keep their lists or copy from a HIGHER slot (its
        lists as they are before this predict); level L = slots whose parent id names a LOWER slot of level L - 1 (its lists after
        its own birth step).  Per level: the owners of the needed source slots publish (unused mask, candidate list), every rank
        installs what its slots need, then the shard runs the birth step of that level (rfsgpu_predict_map_level).  Host-staged
        (an all-gather of small Python objects): it happens in the predicts that follow a resampling only."""
        N, n, lo = self.n_total, self.n_local, self.rank * self.n_local
        p = self.ppid
        level = np.zeros(N, dtype=np.int32)
        for i in range(N):
            if p[i] < i:
                level[i] = level[p[i]] + 1
        local_levels = np.ascontiguousarray(level[lo:lo + n])
        for L in range(int(level.max()) + 1):
            dst = np.nonzero((level == L) & (p != np.arange(N)))[0]
            srcs = np.unique(p[dst])
            masks = self.f.get_unused_masks()
            mine = {int(q): (int(masks[q - lo]), self.f.export_birth_candidates(int(q - lo))) for q in srcs if lo <= q < lo + n}
            if self.world > 1:
                parts = [None] * self.world
                dist.all_gather_object(parts, mine, group=self.group)
                table = {k: v for part in parts for k, v in part.items()}
            else:
                table = mine
            changed = False
            for i in dst:
                if lo <= i < lo + n:
                    m, (mean, cov, sup, chk) = table[int(p[i])]
                    masks[i - lo] = np.uint64(m)
                    self.f.import_birth_candidates(int(i - lo), mean, cov, sup, chk)
                    changed = True
            if changed:
                self.f.set_unused_masks(masks)
            self.f.predict_map_level(add_birth, local_levels, L, L == 0)

    def _gather_masks(self):
        local = np.ascontiguousarray(self.f.get_unused_masks(), dtype=np.uint64)
        if self.world == 1:
            return local
        t = torch.from_numpy(local.view(np.int64))
        if self.on_gpu and self.backend == "nccl":
            with self._stream_ctx():
                t = t.to(self.device)
                out = torch.empty(self.n_total, dtype=torch.int64, device=self.device)
                dist.all_gather_into_tensor(out, t, group=self.group)
                return out.cpu().numpy().view(np.uint64)
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t, group=self.group)
        return torch.cat(out).numpy().view(np.uint64)

    # -- RBPHDFilter::update incl. the resample-or-normalise tail (:444-541) -------------------------------
    def update(self, Z, u01=None):
        Z = np.asarray(Z, dtype=np.float64).reshape(-1, self.f.dz)
        self.nUpdatesSinceResample += 1
        if Z.shape[0] == 0:
            return False
        self.nMeasurementsSinceResample += Z.shape[0]
        cfg = self.f.get_filter_config()
        due = (self.nUpdatesSinceResample >= cfg.minUpdatesBeforeResample and
               self.nMeasurementsSinceResample >= cfg.minMeasurementsBeforeResample)
        if self.on_gpu and not due and self.defer_normalisation:
            # the resample test is not due: nobody reads N_eff after this step, the normalisation may trail (update_deferred)
            self.update_deferred(Z)
            self.resampleOccured = False
            return False
        if self.on_gpu:
            self.flush_deferred()         # (a trailing total of the previous step divides the weights before this step multiplies them)
            self.f.update_async(Z)        # stream-ordered; device errors surface at the next synchronize()
        else:
            self.f.update(Z)
        self.resampleOccured = False
        if due:
            self.resampleOccured = self.resample(u01)
        if self.resampleOccured:
            self.nUpdatesSinceResample = 0
            self.nMeasurementsSinceResample = 0
        else:
            self.normalize()          # (after a resample() that did not fire this is a second division, as in the reference :537-539)
        if self.on_gpu:
            self.f.synchronize()
        return self.resampleOccured

    # -- ParticleFilter::resample with global semantics (:399-492) -----------------------------------------
    def resample(self, u01=None):
        self.last_resample_plan = None
        tot = self.normalize(need_totals=True)        # weights now sum to 1 across all shards
        # N_eff = 1 / sum (w_i / S)^2 = S^2 / sum w_i^2  (tot holds the pre-normalisation sums)
        neff = float(tot[0] * tot[0] / tot[1])
        if neff > self.effNParticles_t and neff / self.n_total > self.effNParticles_t / self.n_total:
            return False
        # one uniform draw for the whole filter (the reference's single drand48()): rank 0 decides
        u = torch.tensor([np.random.random() if u01 is None else float(u01)], dtype=torch.float64)
        if self.world > 1:
            if self.on_gpu and self.backend == "nccl":
                with self._stream_ctx():
                    ud = u.to(self.device)
                    dist.broadcast(ud, 0, group=self.group)
                    u = ud.cpu()
            else:
                dist.broadcast(u, 0, group=self.group)
        w_all = self.gather_weights()
        plan = systematic_resample_plan(w_all, float(u.item()))   # global slot -> global source slot
        self.apply_plan(plan)
        self.last_resample_plan = plan
        return True

    def apply_plan(self, plan):
        """Carry out a global resampling plan: cross-shard children first leave as packed rows (device memory), local
        children are a device gather, then the received rows are unpacked into this shard's dead slots."""
        n, r, W = self.n_local, self.rank, self.world
        lo = r * n
        plan = np.asarray(plan, dtype=np.int64)
        g_all = np.nonzero(plan != np.arange(self.n_total))[0]             # children (ascending global slot)
        src_rank, dst_rank = plan[g_all] // n, g_all // n
        cross = src_rank != dst_rank
        # rows I send, grouped by destination rank, children in ascending order (the receiver derives the same order)
        send_g = [g_all[cross & (src_rank == r) & (dst_rank == d)] for d in range(W)]
        recv_g = [g_all[cross & (dst_rank == r) & (src_rank == s)] for s in range(W)]
        n_send, n_recv = sum(len(x) for x in send_g), sum(len(x) for x in recv_g)
        R = self.f.slab_row_bytes() if (n_send or n_recv) else 0
        rows_dev = self.device if self.on_gpu else torch.device("cpu")
        with self._stream_ctx():
            send_buf = torch.empty(max(n_send, 1) * max(R, 1), dtype=torch.uint8, device=rows_dev)
            recv_buf = torch.empty(max(n_recv, 1) * max(R, 1), dtype=torch.uint8, device=rows_dev)
            if n_send:
                slots = np.concatenate([plan[g] - lo for g in send_g]).astype(np.int32)
                self.f.export_slab_rows(slots, send_buf.data_ptr())
            if W > 1 and self.on_gpu and self.backend == "nccl":
                # RCCL: ONE all-to-all with per-peer byte counts (every rank derives both lists from the same plan); a collective
                # every rank joins, rows move device to device over xGMI
                in_splits = [len(send_g[d]) * R for d in range(W)]
                out_splits = [len(recv_g[q]) * R for q in range(W)]
                dist.all_to_all_single(recv_buf[:n_recv * R], send_buf[:n_send * R], out_splits, in_splits, group=self.group)
            elif W > 1 and (n_send or n_recv):
                staged = self.on_gpu and self.backend != "nccl"          # gloo: no device-to-device send/recv
                sb = send_buf.cpu() if staged else send_buf
                rb = torch.empty_like(recv_buf, device="cpu") if staged else recv_buf
                ops, off = [], 0
                for d in range(W):
                    k = len(send_g[d])
                    if k:
                        ops.append(dist.P2POp(dist.isend, sb[off * R:(off + k) * R], d, group=self.group))
                        off += k
                off = 0
                for s in range(W):
                    k = len(recv_g[s])
                    if k:
                        ops.append(dist.P2POp(dist.irecv, rb[off * R:(off + k) * R], s, group=self.group))
                        off += k
                for req in dist.batch_isend_irecv(ops):
                    req.wait()                                           # nccl: the stream waits; gloo: the host does
                if staged:
                    recv_buf.copy_(rb)
            # local children: device gather (sources keep themselves, so in place is hazard-free); resets every weight to 1
            local_src = np.arange(n, dtype=np.int32)
            mine = g_all[(dst_rank == r) & ~cross]
            local_src[mine - lo] = (plan[mine] - lo).astype(np.int32)
            self.f.resample_apply(local_src)
            if n_recv:
                slots = (np.concatenate(recv_g) - lo).astype(np.int32)
                self.f.import_slab_rows(slots, recv_buf.data_ptr())
            if self.on_gpu:
                self.f.synchronize()                                     # the buffers may be released after this
        self.last_migration = dict(rows_sent=int(n_send), rows_received=int(n_recv), bytes_sent=int(n_send * R),
                                   rows_to_rank=[int(len(x)) for x in send_g], rows_from_rank=[int(len(x)) for x in recv_g])
        # ids as ParticleFilter::resample leaves them (:446-479; a copy keeps its source's id, Particle::copy), over GLOBAL slots
        child = plan != np.arange(self.n_total)
        src_id = self.pid[plan]
        self.pid = np.where(child, src_id, self.pid)
        self.ppid = np.where(child, src_id, self.pid)


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def bench_resample_migration(pkg, f, rank, world, dev, stream=None, sums=None, reps=3):
    """bench.py's separate figure: ONE global resampling step with cross-shard migration on the benchmark's shards.  The
    weights are skewed by rank (rank r's particles weigh r + 1), so the systematic plan sends about a third of the
    higher ranks' particles' children to lower ranks.  Returns max-over-ranks time, rows and bytes moved."""
    import time
    sh = ShardedRBPHDFilter(f, device=dev, stream=stream, sums=sums)
    sh.effNParticles_t = sh.n_total + 1.0                       # force
    times = []
    for rep in range(reps + 1):
        f.restore_state()
        f.set_weights(np.full(f.n, float(rank + 1)))
        f.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        fired = sh.resample(u01=0.37)
        f.synchronize()
        dt = time.perf_counter() - t0
        assert fired
        if rep:
            times.append(dt)
    t = torch.tensor([float(np.median(times)), float(sh.last_migration["rows_sent"]), float(sh.last_migration["bytes_sent"])],
                     dtype=torch.float64, device=dev)
    sh.close()                                                  # the handle goes back in the mode it came in
    tmax = t.clone()
    # who sent how many rows to whom (row r = rank r's per-destination counts): the all-to-all's split lists, zeros included
    pair = torch.tensor(sh.last_migration.get("rows_to_rank", [0] * world), dtype=torch.float64, device=dev)
    pairs = pair.clone().unsqueeze(0)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        allp = torch.zeros(world * world, dtype=torch.float64, device=dev)
        allp[rank * world:(rank + 1) * world] = pair
        dist.all_reduce(allp, op=dist.ReduceOp.SUM)
        pairs = allp.reshape(world, world)
    return dict(ms=round(float(tmax[0]) * 1e3, 4), rows_migrated_total=int(t[1].item()), bytes_migrated_total=int(t[2].item()),
                row_bytes=int(f.slab_row_bytes()), rows_from_rank_to_rank=[[int(v) for v in row] for row in pairs.cpu().tolist()],
                note="one forced global systematic resampling: all-reduce + all-gather of the weights, plan on the host, local gather, "
                     "cross-shard children as packed rows device->device over RCCL send/recv; weights skewed by rank so that rows move")
