// step_fused.h -- one launch for the whole map update of a step (RBPHDFilter::update body, include/RBPHDFilter.hpp:469-520):
// a workgroup of STEP_WPP waves owns one particle and takes it through updateMap -> importanceWeighting -> merge -> prune.
//
// Why: as three kernels, every phase ends with a tail in which the device waits for its slowest particles (a phase lasts
// as long as its slowest particle, about twice the median at C2), and the slabs make a round trip through L2/HBM between
// launches.  Fused, workgroups drift apart -- some are in the latency-bound map update while others are in the
// ALU-heavy weighting -- so the SIMDs see a mix of work, a particle that is slow in one phase is usually not slow in
// the next, and each particle's mixture is still warm in L2 when its next phase reads it.
//
// The phases are the per-particle device functions of the stand-alone kernels, unchanged (update_map.h, weighting.h,
// merge_prune.h) -- the map update in its workgroup form -- sharing one LDS allocation, separated by workgroup barriers.
#pragma once
#include "common.h"
#include "update_map.h"
#include "weighting.h"
#include "merge_prune.h"

#ifndef STEP_WPP
#define STEP_WPP 2
#endif
// What the head of a step does before the map update (rfsgpu_cycle_async / rfsgpu_update_io, round 5).
// mode: the predict folded in -- 0 = none, 1 = static step only, 2 = births + static step.
// inPacked: the host's new inputs in a PINNED host buffer the kernel reads over PCIe itself, 13 doubles per particle side by side
// (pose 3 | pose covariance 9 | weight 1: ONE contiguous 104-byte read per workgroup; three separate arrays were three small PCIe
// reads per workgroup and made the kernel 25 us longer), which the workgroup leaves in the device arrays for every later kernel --
// no copy commands in front of the step: three SDMA / blit copies and their queue hand-overs were 40 us of a 180 us update through
// the boundary at configs[1] (tools/boundary_trace.py).  inMask: bit 0 poses, bit 1 covariances, bit 2 weights present.
struct StepPredict {
  int mode;
  int nZprev;                // measurements of the previous update (B.Z holds them until this step's post kernel)
  const double *birthPose;   // [N][3] the poses the previous update used
  const double *inPacked;
  int inMask;
};
// Cost-ordered launch (round 6; murty.h step_cost_order_class): only the instantiations WITHOUT phase priorities use it -- those are the
// launches whose workgroups are not all resident at once, where the last round's length is that of whatever started last.  The
// all-resident instantiations (configs[1]'s headline among them) ignore the argument: same code as without it.
struct StepLaunchOrder {
  float *cost;        // [N] this step's duration per particle (100 MHz ticks), or nullptr
  const int *order;   // [N] launch slot -> particle, or nullptr: slot == particle
};
// (Measured and dropped, round 5 -- profiles/r05b_*: the step's POST work by the last workgroup of this kernel to finish (a ticket,
//  weights re-stored write-through, results to the pinned landing area) instead of the post kernel's launch, for filters without
//  Murty work: the launch, the queue hand-over and the post kernel it saves cost what the ticket and the delivery cost here --
//  136.7 + 4.2 us against 134.5 + 7.9 us per update through the boundary.  On the way: a device-scope fence per finishing workgroup
//  writes back the XCD's whole L2 each time, 112 -> 215 us; one agent-scope load per weight in the last workgroup, +30 us.)
#ifndef STEP_WAVES_PER_EU
#define STEP_WAVES_PER_EU 4  // <= 128 VGPRs: 8 workgroups of 2 waves per CU, i.e. all 2000 particles of C2 resident at once
#endif

__host__ __device__ inline size_t step_fused_lds_bytes(int cap, int evalCap, int nZ, int wpp, int gridLog = 5) {
  size_t a = (size_t)RFS_Z_LDS_BYTES + update_map_block_lds_bytes(cap);
  const size_t b = (size_t)RFS_Z_LDS_BYTES + weight_lds_bytes_per_wave(cap, evalCap, nZ) + WEIGHT_SCRATCH_BYTES;
  const size_t c = merge_lds_bytes_per_block(cap, wpp, gridLog);
  if (b > a) a = b;
  if (c > a) a = c;
  return (a + 15) & ~(size_t)15;
}
// + the sorting permutation ([cap] u16) handed from the weighting phase to the merge phase, behind every phase's own layout
__host__ __device__ inline size_t step_fused_lds_total(int cap, int evalCap, int nZ, int wpp, int gridLog = 5) {
  return step_fused_lds_bytes(cap, evalCap, nZ, wpp, gridLog) + (((size_t)cap * 2 + 15) & ~(size_t)15);
}

// useWeighting == 0: SC-PHD (useClusterProcess_): the particle weight comes out of the map update, the mixture is not
// sorted, merge works on the slab the update wrote.
// PRED: the instantiation with the predict at its head (a template parameter, not a run-time branch: the births' sin / cos bring
// 60 B of scratch per lane into the kernel, and the plain step -- configs[1]'s headline -- lost 10 us to it when both shared one body;
// the head as a real call with its own register allocation: C2b 0.112 -> 0.208 ms per cycle, profiles/r05a_*).
template <int WPP, bool PHASE_PRIO, int GL = 5, bool PRED = false>
__global__ __launch_bounds__(WPP * 64) __attribute__((amdgpu_waves_per_eu(STEP_WAVES_PER_EU)))
void phd_step_fused_kernel(Buffers B, Params P, int cur, int nZ, int evalCap, int useWeighting, MurtyQueue Q, ZArg zarg, StepPredict SP,
                           StepLaunchOrder SLO) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  int i = blockIdx.x;
  unsigned long long tStart = 0ull;
  if constexpr (!PHASE_PRIO) {
    if (SLO.order) i = SLO.order[blockIdx.x];      // (workgroup-uniform: a scalar load)
    if (SLO.cost) tStart = wall_clock64();
  }
  // Round 5: the map part of the PREDICT that precedes this update (RBPHDFilter::predict, include/RBPHDFilter.hpp:415-442 -- birth
  // Gaussians from the previous update's unused measurements at the poses that update used, then Sigma += Q on every Gaussian,
  // include/ProcessModel.hpp:195-208) runs at the head of the step, by the workgroup that owns the particle: one launch chain per
  // predict + update cycle, and the covariances the static step rewrites are still in L2 when the map update reads them.  The
  // function is the stand-alone predict kernel's (merge_prune.h, predict_map_particle): same bits.  B.Z still holds the PREVIOUS
  // measurement set here (this step's post kernel writes the new one).
  if constexpr (PRED) {
    if (SP.mode) predict_map_particle<WPP * 64>(B, P, cur, i, tid, SP.mode > 1, SP.nZprev, SP.birthPose, true);
    // the host's inputs for THIS update, after the births have read the old pose (same wave, program order)
    if (SP.inMask && tid < 13) {
      const double v = SP.inPacked[13 * i + tid];
      if (tid < 3) { if (SP.inMask & 1) B.pose[3 * i + tid] = v; }
      else if (tid < 12) { if (SP.inMask & 2) B.poseCov[9 * i + (tid - 3)] = v; }
      else if (SP.inMask & 4) B.weight[i] = v;
    }
    __threadfence_block();
    __syncthreads();
    __builtin_amdgcn_s_dcache_inv();   // count / pose / covariance are read through the scalar cache below: drop what the head's own loads left there
  }
  // Issue priority falls from phase to phase (s_setprio 2/3 -> 1 -> 0): the SIMD arbiter otherwise always prefers its oldest waves, so the
  // last workgroups to arrive on a CU crawl through the map update while the first ones race ahead, and the launch lasts as
  // long as those stragglers.  With a workgroup that is a phase behind outranking the ones ahead, the eight workgroups of
  // a CU finish together.  Only when the whole grid is resident at once (PHASE_PRIO, picked by the host): with several
  // rounds of workgroups per CU, newcomers outranking workgroups that are about to free their slots costs more than it gives
  // (measured: +2.3 % at 2000 x 200, -5 % at 2500 x 500).  Levels: map update 2 (3 for a SIMD's last arrivals; set inside
  // phd_update_map_block), weighting 2, the parallel half of the merge (stage, grid, candidate scan, pair tests) 1, its replay and
  // the prune 0 (set inside gm_merge_particle) -- r02: one more level inside the merge bought another 1 % (126.7 -> 125.3 us).
  // The measurement set arrives in the kernel-argument block (no staging launch); the step's post kernel leaves it in the
  // device buffer that later kernels read (the next predict's births).  (Writing it from here cost 112 B/lane of scratch.)
  stage_measurements_lds(smem_raw, [&](int t) { return zarg.v[t]; }, nZ, tid, WPP * 64);
  __syncthreads();
#ifdef RFS_PROFILE
  long long *fd = B.dbg ? B.dbg + 64 + 4 * (size_t)B.N + 4 * (size_t)i : nullptr;
  if (fd && tid == 0) {  // start tick | (XCC_ID << 60) | (HW_ID[15:0] << 44)
    const unsigned hw = __builtin_amdgcn_s_getreg((16 - 1) << 11 | 4), xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20);
    fd[0] = (long long)((wall_clock64() & 0xfffffffffffull) | ((unsigned long long)(hw & 0xffffu) << 44) | ((unsigned long long)(xcc & 0xfu) << 60));
  }
#endif
  if (WPP == 1) phd_update_map_particle(B, P, cur, nZ, i, lane, smem_raw, smem_raw + RFS_Z_LDS_BYTES);
  else phd_update_map_block<WPP>(B, P, cur, nZ, i, tid, smem_raw, smem_raw + RFS_Z_LDS_BYTES, PHASE_PRIO);
  __threadfence_block();  // the slab / count written by wave 0 -> the whole workgroup
  __syncthreads();
  RFS_CUT(8);
#ifdef RFS_PROFILE
  if (fd && tid == 0) fd[1] = (long long)wall_clock64();
#endif
  int mergeSrc = cur;
  // Each phase gets its own copy of the thread index behind a compiler barrier: otherwise address arithmetic common to the
  // phases (tid * 8, ...) is hoisted to the top of the kernel and held -- or spilled -- across all of them.
  int tidW = threadIdx.x;
  asm volatile("" : "+v"(tidW));
#ifndef STEP_W_PRIO
#define STEP_W_PRIO 2
#endif
#ifndef STEP_MERGE_PRIO
#define STEP_MERGE_PRIO 1
#endif
  if (PHASE_PRIO) __builtin_amdgcn_s_setprio(STEP_W_PRIO);
  // The weighting phase sorts the mixture by weight (sortByWeight, include/RBPHDFilter.hpp:733) -- as a permutation kept in
  // LDS; the merge phase walks the slab through it, so the sorted mixture is never written out and read back.
  unsigned short *sPerm = reinterpret_cast<unsigned short *>(smem_raw + step_fused_lds_bytes(B.cap, evalCap, nZ, WPP, GL));
  const unsigned short *mergePerm = nullptr;
  if (useWeighting) {
    phd_weight_particle<WPP>(B, P, cur, cur ^ 1, nZ, evalCap, Q, i, tidW, smem_raw, sPerm);
    __threadfence_block();
    __syncthreads();
    mergePerm = sPerm;
  }
  RFS_CUT(16);
  int tidM = threadIdx.x;
  asm volatile("" : "+v"(tidM));
  if (PHASE_PRIO) __builtin_amdgcn_s_setprio(STEP_MERGE_PRIO);
#ifdef RFS_PROFILE
  if (fd && tid == 0) fd[2] = (long long)wall_clock64();
#endif
  gm_merge_particle<WPP, true, GL>(B, P, mergeSrc, mergeSrc ^ 1, i, tidM, smem_raw, mergePerm);
#ifdef RFS_PROFILE
  if (fd && tid == 0) fd[3] = (long long)wall_clock64();
#endif
  if constexpr (!PHASE_PRIO) {
    if (SLO.cost && threadIdx.x == 0) SLO.cost[i] = (float)(wall_clock64() - tStart);
  }
}
