"""Multi-GPU host for the RB-PHD update path: one process per GPU, particles sharded in contiguous blocks.

Every phase of RBPHDFilter::update() before resampling is independent per particle (reference
include/RBPHDFilter.hpp:469-520), so the only collective on the per-step path is the all-reduce (sum) of
{sum w, sum w^2} -- 2 doubles over RCCL/xGMI -- for ParticleFilter::normalizeWeights / the N_eff test
(include/ParticleFilter.hpp:352-363, 405-415).  When resampling fires, the reference's GLOBAL systematic
resampling (ParticleFilter.hpp:419-479) is kept: the N weights are all-gathered, every rank computes the same plan
from the same uniform draw, local children are a device gather (rfsgpu_resample_apply), cross-shard children migrate
as packed mixtures.  Shard-local resampling would change results and is not offered.

Backend-agnostic: `local` is any object with the filter interface of capi.CFilter (the device engine in production;
tests substitute a CPU stand-in to exercise this host logic under gloo).
"""
import numpy as np
import torch
import torch.distributed as dist

from .engine import systematic_resample_plan


class ShardedRBPHDFilter:
    def __init__(self, local, group=None, device=None):
        self.f = local
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.n_local = local.n
        self.n_total = self.n_local * self.world
        self.device = device if device is not None else torch.device("cpu")
        self.effNParticles_t = self.n_total / 4.0          # ParticleFilter.hpp:232
        self.nUpdatesSinceResample = 0
        self.nMeasurementsSinceResample = 0
        self.resampleOccured = False

    # -- collectives ---------------------------------------------------------------------------------------
    def _allreduce_sums(self, sums_tensor=None):
        """{sum w, sum w^2} over all shards.  With a device tensor bound to the engine the reduction never
        leaves the GPU; otherwise the 2 doubles go through a host tensor (tests / CPU stand-in)."""
        if sums_tensor is not None:
            self.f.weight_sums_async()
            if self.world > 1:
                dist.all_reduce(sums_tensor, group=self.group)
            return sums_tensor
        s = torch.from_numpy(np.asarray(self.f.weight_sums(), dtype=np.float64)).to(self.device)
        if self.world > 1:
            dist.all_reduce(s, group=self.group)
        return s

    def normalize(self, sums_tensor=None):
        s = self._allreduce_sums(sums_tensor)
        if sums_tensor is not None:
            self.f.normalize_weights(0.0, sums_tensor.data_ptr())   # divisor read on the device
            return None
        tot = s.cpu().numpy()
        self.f.normalize_weights(float(tot[0]))
        return tot

    def gather_weights(self):
        w = torch.from_numpy(self.f.get_weights()).to(self.device)
        if self.world == 1:
            return w.cpu().numpy()
        out = [torch.empty_like(w) for _ in range(self.world)]
        dist.all_gather(out, w, group=self.group)
        return torch.cat(out).cpu().numpy()

    # -- RBPHDFilter::update incl. the resample-or-normalise tail (:444-541) -------------------------------
    def update(self, Z, u01=None):
        Z = np.asarray(Z, dtype=np.float64).reshape(-1, self.f.dz)
        self.nUpdatesSinceResample += 1
        if Z.shape[0] == 0:
            return False
        self.nMeasurementsSinceResample += Z.shape[0]
        self.f.update(Z)
        cfg = self.f.get_filter_config()
        self.resampleOccured = False
        if (self.nUpdatesSinceResample >= cfg.minUpdatesBeforeResample and
                self.nMeasurementsSinceResample >= cfg.minMeasurementsBeforeResample):
            self.resampleOccured = self.resample(u01)
        if self.resampleOccured:
            self.nUpdatesSinceResample = 0
            self.nMeasurementsSinceResample = 0
        else:
            self.normalize()
        return self.resampleOccured

    # -- ParticleFilter::resample with global semantics (:399-492) -----------------------------------------
    def resample(self, u01=None):
        tot = self.normalize()                        # weights now sum to 1 across all shards
        # N_eff = 1 / sum (w_i / S)^2 = S^2 / sum w_i^2  (tot holds the pre-normalisation sums)
        neff = float(tot[0] * tot[0] / tot[1])
        if neff > self.effNParticles_t and neff / self.n_total > self.effNParticles_t / self.n_total:
            return False
        # one uniform draw for the whole filter (the reference's single drand48()): rank 0 decides
        u = torch.tensor([np.random.random() if u01 is None else float(u01)], dtype=torch.float64, device=self.device)
        if self.world > 1:
            dist.broadcast(u, 0, group=self.group)
        w_all = self.gather_weights()
        plan = systematic_resample_plan(w_all, float(u.item()))   # global slot -> global source slot
        self.apply_plan(plan)
        return True

    def apply_plan(self, plan):
        n, r = self.n_local, self.rank
        lo = r * n
        # 1. what I must send: children on other ranks whose source is one of my slots
        outgoing = {}
        for g in np.nonzero(plan != np.arange(self.n_total))[0]:
            s = int(plan[g])
            if lo <= s < lo + n and not (lo <= g < lo + n):
                w, _, mean, cov = self.f.export_gm(s - lo)
                outgoing[int(g)] = (w, mean, cov, self.f.get_unused(s - lo), self.f.landmarks_in_fov(s - lo),
                                    self.f.export_birth_candidates(s - lo))
        if self.world > 1:
            gathered = [None] * self.world
            dist.all_gather_object(gathered, outgoing, group=self.group)
        else:
            gathered = [outgoing]
        # 2. local children: device gather (sources keep themselves, so in place is hazard-free)
        local_src = np.arange(n, dtype=np.int32)
        for k in range(n):
            s = int(plan[lo + k])
            if lo <= s < lo + n:
                local_src[k] = s - lo
        self.f.resample_apply(local_src)              # also resets every weight to 1 (ParticleFilter.hpp:486-489)
        # 3. migrated children: import the packed mixtures into the dead slots
        for d in gathered:
            for g, (w, mean, cov, unused, nfov, cands) in d.items():
                if lo <= g < lo + n:
                    self.f.import_gm(g - lo, w, mean, cov)
                    self.f.import_aux(g - lo, unused, nfov)
                    self.f.import_birth_candidates(g - lo, *cands)   # birthGaussians_ travel with the particle (:1005-1011)
