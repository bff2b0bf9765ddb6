// rfsgpu_engine.hip -- host side of the C ABI in include/rfsgpu.h: device memory, kernel launches, timing.
// One handle == one GPU == one shard of particles.  No CPU fallback: every entry point either runs the HIP
// kernels or returns an error status.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define RFSGPU_ENABLE_BENCH_API 1   // the library defines (and exports) the [bench] / [test] entry points too
#include "rfsgpu.h"
#include "common.h"
#include "update_map.h"
#include "weighting.h"
#include "merge_prune.h"
#include "step_fused.h"
#include "murty.h"
#include "mat_perm.h"
#include "vp.h"
#include "birth.h"
#include "fastslam.h"
#include "fastslam_mh.h"
#include "motion.h"

namespace {

inline long long now_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

enum { EV_UM0 = 0, EV_UM1, EV_W1, EV_MG1, EV_PR1, EV_P0, EV_P1, EV_R0, EV_R1, EV_COUNT };

}  // namespace

struct rfsgpu_filter {
  int device = 0;
  int N = 0, cap = 0;       // particles in use, Gaussians per particle
  int snapN = 0;            // particle count of the saved state
  int Ncap = 0;             // particle slots allocated (rfsgpu_create_ex >= N): the multi-hypothesis FastSLAM update grows N up to it
  int model = RFSGPU_MODEL_RNGBRG_2D;
  int D = 2;          // d_m == d_z
  rfsgpu_vp_config vp{};
  double vpClutter = 0.0;
  hipStream_t stream = nullptr;
  hipStream_t ownStream = nullptr;
  double *ownSums = nullptr;
  // snapshot slot (rfsgpu_save_state)
  double *snapSlab = nullptr, *snapWeight = nullptr;
  int *snapCount = nullptr, *snapFov = nullptr;
  unsigned long long *snapUnused = nullptr;
  int snapNZ = 0;
  // ring of pre-seeded copies of the saved state (rfsgpu_state_ring_*): a timed step takes its input from the next slot by a pointer swap
  struct RingSlot { double *slab = nullptr, *weight = nullptr; int *count = nullptr, *fov = nullptr; unsigned long long *unused = nullptr; };
  std::vector<RingSlot> stateRing;
  size_t stateRingPos = 0;
  hipEvent_t ev[EV_COUNT] = {};
  Buffers B{};
  int cur = 0;
  Params P{};
  rfsgpu_filter_config cfg{};
  rfsgpu_rngbrg_config rb{};
  rfsgpu_kf_config kf{};
  double Qlm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  int nZ = 0;  // measurements of the last update (birth uses them)
  double *dSums = nullptr;  // [2]
  int *dSrcSlot = nullptr;  // [N]
  int *dRowSlots = nullptr; // slots of the rows being exported / imported (allocated on first use, grown on demand)
  int rowSlotsCap = 0;
  double *hStage[4] = {nullptr, nullptr, nullptr, nullptr};  // pinned staging ring of rfsgpu_set_step_inputs_async
  double *hWeights = nullptr;         // pinned landing buffer of rfsgpu_get_weights
  double *hOutW = nullptr;            // pinned: where the post kernel leaves the weights of a synchronous rfsgpu_update_io ...
  int *hOutFlag = nullptr;            // ... and {error word, sequence number} behind them (same allocation)
  int outSeq = 0;                     // sequence number of the last delivery asked for
  bool outArmed = false;              // the step in flight delivers through hOutW / hOutFlag (update_io_end spins instead of synchronising)

  int stagePendingSlot = -1;          // a staging slot whose event must be recorded behind the step that reads it
  bool denseIntensity = false;        // RFSGPU_DENSE_INTENSITY=1 at create: deviation 9 off (the dense loop for every mixture size), for runs against a future pinned fixture
  bool ioPull = true;                 // RFSGPU_IO_PULL=0: inputs by copy commands, outputs by copies + a stream synchronisation (A/B, rounds 3-4 form)
  int vpParity = 0;                   // which pair of duration extrema (vpCost[Ncap ..]) the next post kernel reads
  float *vpCost = nullptr;            // [Ncap + 4] Victoria Park fused step: last measured duration per particle (100 MHz ticks)
  int *vpOrder = nullptr;             // [Ncap] launch slot -> particle (identity until the first post kernel has sorted the costs)
  int vpOrderN = -1;                  // particle count the order array is valid for
  int vpOrderMode = 2;                // 0 slot == particle | 1 a host-given order, frozen | 2 re-sorted by every step's post kernel (default; RFSGPU_VP_COST_ORDER=0: 0)
  int collProbeSeq = 0;               // probes made so far (rfsgpu_collective_probe)
  int *dCollSeq = nullptr;            // [2 + 2] device: {number of the last step whose sums are out, number of the last step whose collective is done} (rfsgpu_step_async_trailing)
  int collStep = 0;                   // steps issued in the event-free trailing form
  double *poseAlt = nullptr;          // [Ncap][3] second pose buffer: a fused predict + update cycle births at the old poses and updates at the new ones (rfsgpu_cycle_async)
  hipEvent_t evStage[4] = {};
  int stageNext = 0;
  bool predPending = false;  // rfsgpu_predict_map_async's event pair has not been accumulated yet
  bool candUsed = false;     // candidate lists were imported or the FastSLAM path ran with landmark candidates: migration rows carry them
  bool poseCovZero = false;  // B.poseCov[0..9) holds zeros (the "no covariance" form), so a further cov == NULL push need not rewrite it
  MurtyQueue Q{};
  MurtyScratch MS{};
  int *hErr = nullptr;      // pinned
  int *hJobCount = nullptr; // pinned
  double *hSums = nullptr;  // pinned [2]
  rfsgpu_timing timing{};
  long long lastKernelNs[4] = {0, 0, 0, 0};
  int lastStepVariant[4] = {0, 0, 0, 0};   // the last stream-ordered step: {waves per particle, phase priorities, merge grid log2, fused}
  bool phaseOpen = false;   // update_map ran, weighting/merge/prune may follow
  bool normPending = false; // a normalize_kernel event pair has not been accumulated yet
  // async steps: ring of per-phase event sets, harvested at the next sync
  hipEvent_t ring[RFSGPU_ASYNC_RING][5] = {};  // step phases: 0 start, 1 map update done, 2 weighting (+Murty) done, 3 merge done, 4 weighting kernel done
  bool ringHasMid[RFSGPU_ASYNC_RING] = {};
  bool ringFused[RFSGPU_ASYNC_RING] = {};   // the step ran as ONE kernel: only events 0 and 3 were recorded
  rfsgpu_fastslam_config fs;          // FastSLAM::Config (rfsgpu_fastslam_update)
  unsigned char *fsArena = nullptr;   // per-particle Hungarian scratch (allocated on first use)
  unsigned char *mhArena = nullptr;   // multi-hypothesis FastSLAM: per-particle table / Murty arena / assignments
  int *mhInts = nullptr;              // [5][Ncap] slotSrc, slotHyp, slotNH, copyDst, copySrc
  bool fsResampleOccured = false;     // FastSLAM::resampleOccured_ of the previous update (rfsgpu_fastslam_set_resample_occured)
  std::vector<int> parents;           // source slot of every particle after the last FastSLAM update (identity when none multiplied)
  // birth-state inheritance after a resampling (rfsgpu_set_birth_inheritance; RBPHDFilter.hpp:1005-1011)
  int inheritMode = RFSGPU_INHERIT_REFERENCE;
  std::vector<int> pid, ppid;         // Particle::id_ / idParent_ of the particle in each slot (ParticleFilter.hpp:446-479)
  bool resampleOccured = false;       // RBPHDFilter::resampleOccured_
  bool externalAck = true;            // RFSGPU_INHERIT_EXTERNAL: the host has taken care of the inheritance rule since the last resampling (rfsgpu.h)
  bool fastSlamHandle = false;        // rfsgpu_fastslam_update has run: the filter class is rfs::FastSLAM, whose resampleWithMapCopy
                                      // copies the candidate lists right at resampling time (FastSLAM.hpp:747-753) -> eager copies
  int *dInhParent = nullptr, *dInhLevel = nullptr;   // [Ncap] each (allocated on first use)
  BirthLists inhTmp{};                // staging area of birth_inherit_kernel<0/1> (allocated on first use)
  std::vector<int> hInhParent, hInhLevel;
  bool holes = false;       // between rfsgpu_merge and rfsgpu_prune merged-away entries sit in the slab with w = -1; at any other
                            // time a negative weight is a value (FastSLAM's log-odds) and every stored entry counts
  int nCU = 256;            // multiProcessorCount of the device
  bool fuseSteps = true;    // rfsgpu_update / _update_async / _step_async use phd_step_fused_kernel (2-D model); RFSGPU_FUSED_STEP=0 turns it off
  bool phaseTiming = false; // rfsgpu_set_phase_timing: rfsgpu_update runs its phases as separate launches (TimingInfo per phase)
  int stepWppOverride = 0;  // RFSGPU_STEP_WPP: waves per particle of the fused step kernel (2 or 3); 0 = chosen per launch
  unsigned stepSeq = 0;      // fused steps launched so far
  int timingStride = 8;      // every timingStride-th of them carries the timing events (rfsgpu_set_step_timing_stride; round 5: 8 by default --
                             // three marker packets per step cost a configs[1] step 8 us; the FIRST step of a filter is always timed)
  int untimedSince = 0;      // fused steps launched since the last one that carried events (and not yet booked into TimingInfo)
  long long lastFusedNs = 0; // duration of the last sampled fused step (what an un-sampled step is booked at)
  int ringStands[RFSGPU_ASYNC_RING] = {};   // how many steps a timed ring entry stands for in TimingInfo (itself + the untimed ones before it)
  int mergeGridOverride = 0; // RFSGPU_MERGE_GRID: log2 of the merge grid's cells per side in the three-wave fused kernel (5 or 6); 0 = chosen per launch
  hipEvent_t evAfterWeightKernel = nullptr;   // where launch_weighting drops its mid-phase event (async steps only)
  int ringCount = 0;        // async steps recorded since the last harvest
  double statNs[3] = {0, 0, 0};
  double lastPostNs = 0;
  double statPostNs = 0;    // fused steps: the post kernel's share (rfsgpu_post_kernel_time_stats)
  int statSteps = 0;
  std::string err;
  int maxLds = 0;
  int wpbUpdate = 4, wpbWeight = 4, wpbMerge = 4, wpbPrune = 4;
};

#define HIPCHK(call)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) {                                                                \
      f->err = std::string(#call) + ": " + hipGetErrorString(e_);                          \
      return RFSGPU_ERR_HIP;                                                               \
    }                                                                                      \
  } while (0)

#define CHECK_HANDLE(f) \
  if (!(f)) return RFSGPU_ERR_INVALID;

static int fail(rfsgpu_filter *f, int code, const char *msg) {
  f->err = msg;
  return code;
}

static void rebuild_params(rfsgpu_filter *f) {
  Params &P = f->P;
  P.denseIntensity = f->denseIntensity ? 1 : 0;
  for (int k = 0; k < 4; k++) P.R[k] = f->rb.R[k];
  P.Pd = f->rb.probabilityOfDetection;
  P.clutter = f->rb.uniformClutterIntensity;
  P.rmax = f->rb.rangeLimMax;
  P.rmin = f->rb.rangeLimMin;
  P.rbuf = f->rb.rangeLimBuffer;
  P.rmaxIn = P.rmax - P.rbuf; P.rmaxOut = P.rmax + P.rbuf; P.rminIn = P.rmin + P.rbuf; P.rminOut = P.rmin - P.rbuf;
  P.kfRange = f->kf.rangeInnovationThreshold;
  P.kfBearing = f->kf.bearingInnovationThreshold;
  P.birthW = f->cfg.birthGaussianWeight;
  P.newGaussMd2 = f->cfg.newGaussianCreateInnovMDThreshold * f->cfg.newGaussianCreateInnovMDThreshold;
  P.evalMinW = f->cfg.importanceWeightingEvalPointGuassianWeight;
  P.weightingMd2 = f->cfg.importanceWeightingMeasurementLikelihoodMDThreshold * f->cfg.importanceWeightingMeasurementLikelihoodMDThreshold;
  P.mergeT2 = f->cfg.gaussianMergingThreshold * f->cfg.gaussianMergingThreshold;
  P.mergeInfl = f->cfg.gaussianMergingCovarianceInflationFactor;
  P.pruneT = f->cfg.gaussianPruningThreshold;
  if (f->D == 2) {
    P.Qlm[0] = f->Qlm[0];
    P.Qlm[1] = f->Qlm[1];
    P.Qlm[2] = f->Qlm[3];
    P.twoPiPowD = pow(2 * acos(-1), 2);
  } else {
    P.Qlm6[0] = f->Qlm[0]; P.Qlm6[1] = f->Qlm[1]; P.Qlm6[2] = f->Qlm[2];
    P.Qlm6[3] = f->Qlm[4]; P.Qlm6[4] = f->Qlm[5]; P.Qlm6[5] = f->Qlm[8];
    P.twoPiPowD = pow(2 * acos(-1), 3);
    for (int k = 0; k < 9; k++) P.R9[k] = f->vp.R[k];
    P.R[0] = f->vp.R[0]; P.R[1] = f->vp.R[1]; P.R[2] = f->vp.R[3]; P.R[3] = f->vp.R[4];  // 2x2 range-bearing block
    P.Slb = f->vp.Slb;
    for (int k = 0; k < RFSGPU_VP_MAX_PD; k++) P.PdTable[k] = f->vp.PdTable[k];
    P.nPd = f->vp.nPd;
    P.vpClutter = f->vpClutter;
    P.vpExpClutter = f->vp.expectedClutterNumber;
    P.rmax = f->vp.rangeLimMax; P.rmin = f->vp.rangeLimMin;
    P.bmax = f->vp.bearingLimitMax; P.bmin = f->vp.bearingLimitMin;
    P.bufferPd = f->vp.bufferZonePd;
  }
  P.birthCheckThr = f->cfg.birthGaussianMeasurementCheckThreshold;
  P.birthSupportD2 = f->cfg.birthGaussianMeasurementSupportDist * f->cfg.birthGaussianMeasurementSupportDist;
  P.evalCount = f->cfg.importanceWeightingEvalPointCount;
  P.useCluster = f->cfg.useClusterProcess ? 1 : 0;
  P.birthCountThr = f->cfg.birthGaussianMeasurementCountThreshold;
  P.birthCurThr = f->cfg.birthGaussianCurrentMeasurementCountThreshold;
}

static int check_device_errors(rfsgpu_filter *f) {
  // one 4-byte D2H + stream sync per phase group
  HIPCHK(hipMemcpyAsync(f->hErr, f->B.err, sizeof(int), hipMemcpyDeviceToHost, f->stream));
  HIPCHK(hipStreamSynchronize(f->stream));
  const int e = *f->hErr;
  if (e == 0) return RFSGPU_OK;
  HIPCHK(hipMemsetAsync(f->B.err, 0, sizeof(int), f->stream));
  // every flag that was raised goes into the message; the status code is the first one's
  std::string msg;
  int code = RFSGPU_OK;
  auto add = [&](int bit, int c, const char *m) {
    if (!(e & bit)) return;
    if (!msg.empty()) msg += "; ";
    msg += m;
    if (code == RFSGPU_OK) code = c;
  };
  add(ERRBIT_CAPACITY, RFSGPU_ERR_CAPACITY, "a particle's Gaussian mixture outgrew gm_capacity (raise it in rfsgpu_create)");
  add(ERRBIT_MURTY, RFSGPU_ERR_UNSUPPORTED, "Murty job queue overflow, a partition larger than MURTY_MAXN, or a FastSLAM association component beyond the in-kernel solver");
  add(ERRBIT_EVALPTS, RFSGPU_ERR_UNSUPPORTED, "more than RFSGPU_MAX_EVAL evaluation points requested");
  add(ERRBIT_BIRTHLIST, RFSGPU_ERR_UNSUPPORTED, "a particle's birth-candidate / landmark-candidate list outgrew RFSGPU_MAX_CANDIDATES");
  add(ERRBIT_COLLECTIVE, RFSGPU_ERR_UNSUPPORTED, "collective hand-over timed out: the sequence number of the weight all-reduce was not published within 0.5 s (rfsgpu_step_async_trailing / rfsgpu_collective_gate / _publish)");
  if (code != RFSGPU_OK) { f->err = msg; return code; }
  return fail(f, RFSGPU_ERR_HIP, "unknown device error flag");
}

static void accumulate(hipEvent_t a, hipEvent_t b, long long &acc, long long *last) {
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, a, b) == hipSuccess) {
    long long ns = (long long)(ms * 1e6);
    acc += ns;
    if (last) *last = ns;
  }
}

extern "C" {

int rfsgpu_abi_version(void) { return RFSGPU_ABI_VERSION; }

void rfsgpu_default_filter_config(rfsgpu_filter_config *c) {  // RBPHDFilter.hpp:370-382
  memset(c, 0, sizeof(*c));
  c->birthGaussianWeight = 0.25;
  c->birthGaussianMeasurementCountThreshold = 1;
  c->birthGaussianMeasurementCheckThreshold = 1;
  c->birthGaussianMeasurementSupportDist = 1;
  c->birthGaussianCurrentMeasurementCountThreshold = 1;
  c->gaussianMergingThreshold = 0.5;
  c->gaussianMergingCovarianceInflationFactor = 1.5;
  c->gaussianPruningThreshold = 0.2;
  c->importanceWeightingEvalPointCount = 8;
  c->importanceWeightingEvalPointGuassianWeight = 0;
  c->importanceWeightingMeasurementLikelihoodMDThreshold = 3.0;
  c->newGaussianCreateInnovMDThreshold = 0.2;
  c->minUpdatesBeforeResample = 1;
  c->minMeasurementsBeforeResample = 1;
  c->useClusterProcess = 0;
}

int rfsgpu_create(rfsgpu_filter **out, int model, int n_particles, int device_id, int gm_capacity) {
  return rfsgpu_create_ex(out, model, n_particles, device_id, gm_capacity, n_particles);
}
int rfsgpu_create_ex(rfsgpu_filter **out, int model, int n_particles, int device_id, int gm_capacity, int max_particles) {
  if (max_particles < n_particles) return RFSGPU_ERR_INVALID;
  if (!out || n_particles <= 0 || (model != RFSGPU_MODEL_RNGBRG_2D && model != RFSGPU_MODEL_VICTORIAPARK_3D) || gm_capacity <= 0) return RFSGPU_ERR_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) return RFSGPU_ERR_NO_DEVICE;
  rfsgpu_filter *f = new rfsgpu_filter();
  f->device = device_id;
  f->N = n_particles;
  f->Ncap = max_particles;
  f->model = model;
  f->D = (model == RFSGPU_MODEL_VICTORIAPARK_3D) ? 3 : 2;
  f->cap = ((gm_capacity + 63) / 64) * 64;
  rfsgpu_default_fastslam_config(&f->fs);
  f->fs.nParticlesMax = 3 * n_particles;
  { const char *e = getenv("RFSGPU_FUSED_STEP"); if (e && e[0] == '0') f->fuseSteps = false; }
  { const char *e = getenv("RFSGPU_IO_PULL"); if (e && e[0] == '0') f->ioPull = false; }
  { const char *e = getenv("RFSGPU_DENSE_INTENSITY"); if (e && e[0] == '1') f->denseIntensity = true; }
  { const char *e = getenv("RFSGPU_STEP_WPP"); if (e) f->stepWppOverride = atoi(e); }
  { const char *e = getenv("RFSGPU_BIRTH_INHERITANCE"); if (e && !strcmp(e, "eager")) f->inheritMode = RFSGPU_INHERIT_EAGER; }   // (initial mode; rfsgpu_set_birth_inheritance)
  { const char *e = getenv("RFSGPU_MERGE_GRID"); if (e) f->mergeGridOverride = atoi(e); }
  if (f->cap > 2048) { delete f; return RFSGPU_ERR_INVALID; }
  auto bail = [&](int code) { rfsgpu_destroy(f); return code; };
  if (hipSetDevice(device_id) != hipSuccess) return bail(RFSGPU_ERR_NO_DEVICE);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return bail(RFSGPU_ERR_NO_DEVICE);
  f->maxLds = (int)prop.sharedMemPerBlock;
  f->nCU = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (hipStreamCreateWithFlags(&f->ownStream, hipStreamNonBlocking) != hipSuccess) return bail(RFSGPU_ERR_HIP);
  f->stream = f->ownStream;
  for (int k = 0; k < EV_COUNT; k++)
    if (hipEventCreate(&f->ev[k]) != hipSuccess) return bail(RFSGPU_ERR_HIP);
  Buffers &B = f->B;
  B.N = f->N;
  B.cap = f->cap;
  B.npl = (f->D == 3) ? (int)P3_COUNT : (int)PL_COUNT;
  const size_t slabBytes = (size_t)f->Ncap * B.npl * f->cap * sizeof(double);
  bool ok = true;
  ok &= hipMalloc(&B.slab[0], slabBytes) == hipSuccess;
  ok &= hipMalloc(&B.slab[1], slabBytes) == hipSuccess;
  ok &= hipMalloc(&B.count, f->Ncap * sizeof(int)) == hipSuccess;
  ok &= hipMalloc(&B.pose, (size_t)f->Ncap * 3 * sizeof(double)) == hipSuccess;
  ok &= hipMalloc(&B.poseCov, (size_t)f->Ncap * 9 * sizeof(double)) == hipSuccess;
  ok &= hipMalloc(&f->poseAlt, (size_t)f->Ncap * 3 * sizeof(double)) == hipSuccess;
  ok &= hipMalloc(&f->dCollSeq, 4 * sizeof(int)) == hipSuccess;      // + [2] probe word, [3] probe verdict (rfsgpu_collective_probe)
  if (ok) ok &= hipMemset(f->dCollSeq, 0, 4 * sizeof(int)) == hipSuccess;
  ok &= hipMalloc(&B.weight, f->Ncap * sizeof(double)) == hipSuccess;
  ok &= hipMalloc(&B.unusedMask, f->Ncap * sizeof(unsigned long long)) == hipSuccess;
  ok &= hipMalloc(&B.nInFov, f->Ncap * sizeof(int)) == hipSuccess;
  ok &= hipMalloc(&B.err, sizeof(int)) == hipSuccess;
  ok &= hipMalloc(&B.Z, RFSGPU_MAX_Z * 3 * sizeof(double)) == hipSuccess;
  ok &= hipMalloc(&B.scan, RFSGPU_VP_MAX_SCAN * sizeof(double)) == hipSuccess;
  ok &= hipMalloc(&B.candMean, (size_t)f->Ncap * RFSGPU_MAX_CANDIDATES * 3 * sizeof(double)) == hipSuccess;
  ok &= hipMalloc(&B.candCov, (size_t)f->Ncap * RFSGPU_MAX_CANDIDATES * 6 * sizeof(double)) == hipSuccess;
  ok &= hipMalloc(&B.candSup, (size_t)f->Ncap * RFSGPU_MAX_CANDIDATES * sizeof(int)) == hipSuccess;
  ok &= hipMalloc(&B.candChk, (size_t)f->Ncap * RFSGPU_MAX_CANDIDATES * sizeof(int)) == hipSuccess;
  ok &= hipMalloc(&B.candCount, (size_t)f->Ncap * sizeof(int)) == hipSuccess;
  ok &= hipMalloc(&f->ownSums, 2 * sizeof(double)) == hipSuccess;
  f->dSums = f->ownSums;
  ok &= hipMalloc(&f->dSrcSlot, f->Ncap * sizeof(int)) == hipSuccess;
  ok &= hipHostMalloc(&f->hErr, sizeof(int)) == hipSuccess;
  ok &= hipHostMalloc(&f->hJobCount, sizeof(int)) == hipSuccess;
  if (ok) *f->hJobCount = 0;   // set by the post kernel once a step has queued Murty jobs (murty_launch)
  ok &= hipHostMalloc(&f->hSums, 2 * sizeof(double)) == hipSuccess;
  for (int k = 0; k < RFSGPU_ASYNC_RING; k++)
    for (int e = 0; e < 5; e++) ok &= hipEventCreate(&f->ring[k][e]) == hipSuccess;
  if (!ok) return bail(RFSGPU_ERR_HIP);
  if (murty_alloc(f->Q, f->MS, f->Ncap) != 0) return bail(RFSGPU_ERR_HIP);
  hipMemsetAsync(B.slab[0], 0, slabBytes, f->stream);
  hipMemsetAsync(B.slab[1], 0, slabBytes, f->stream);
  hipMemsetAsync(B.count, 0, f->Ncap * sizeof(int), f->stream);
  hipMemsetAsync(B.pose, 0, (size_t)f->Ncap * 3 * sizeof(double), f->stream);
  hipMemsetAsync(B.poseCov, 0, (size_t)f->Ncap * 9 * sizeof(double), f->stream);
  hipMemsetAsync(B.unusedMask, 0, f->Ncap * sizeof(unsigned long long), f->stream);
  hipMemsetAsync(B.nInFov, 0, f->Ncap * sizeof(int), f->stream);
  hipMemsetAsync(B.err, 0, sizeof(int), f->stream);
  hipMemsetAsync(B.Z, 0, RFSGPU_MAX_Z * 3 * sizeof(double), f->stream);
  hipMemsetAsync(B.scan, 0, RFSGPU_VP_MAX_SCAN * sizeof(double), f->stream);
  hipMemsetAsync(B.candCount, 0, (size_t)f->Ncap * sizeof(int), f->stream);
  B.nScan = 0;
  set_weights_kernel<<<(f->N + 255) / 256, 256, 0, f->stream>>>(B.weight, f->N, 1.0);
  if (hipStreamSynchronize(f->stream) != hipSuccess) return bail(RFSGPU_ERR_HIP);
  rfsgpu_default_filter_config(&f->cfg);
  memset(&f->rb, 0, sizeof(f->rb));  // MeasurementModel_RngBrg defaults, src/MeasurementModel_RngBrg.cpp:35-43
  f->rb.probabilityOfDetection = 0.95;
  f->rb.uniformClutterIntensity = 0.1;
  f->rb.rangeLimMax = 5;
  f->rb.rangeLimMin = 0.3;
  f->rb.rangeLimBuffer = 0.25;
  memset(&f->vp, 0, sizeof(f->vp));
  f->kf.rangeInnovationThreshold = -1;
  f->kf.bearingInnovationThreshold = -1;
  f->P.poseCovStride = 0;
  rebuild_params(f);
  *out = f;
  return RFSGPU_OK;
}

void rfsgpu_destroy(rfsgpu_filter *f) {
  if (!f) return;
  hipSetDevice(f->device);
  if (f->stream) hipStreamSynchronize(f->stream);
  if (f->ownStream) hipStreamSynchronize(f->ownStream);
  Buffers &B = f->B;
  hipFree(f->snapSlab); hipFree(f->snapWeight); hipFree(f->snapCount); hipFree(f->snapFov); hipFree(f->snapUnused);
  for (auto &r : f->stateRing) { hipFree(r.slab); hipFree(r.weight); hipFree(r.count); hipFree(r.fov); hipFree(r.unused); }
  hipFree(B.slab[0]); hipFree(B.slab[1]); hipFree(B.count); hipFree(B.pose); hipFree(f->poseAlt); hipFree(f->dCollSeq); if (f->vpCost) hipFree(f->vpCost); if (f->vpOrder) hipFree(f->vpOrder); hipFree(B.poseCov); hipFree(B.weight);
  hipFree(B.unusedMask); hipFree(B.nInFov); hipFree(B.err); hipFree(B.Z); hipFree(f->ownSums); hipFree(f->dSrcSlot); if (f->dRowSlots) hipFree(f->dRowSlots); if (f->fsArena) hipFree(f->fsArena); if (f->mhArena) hipFree(f->mhArena); if (f->mhInts) hipFree(f->mhInts);
  if (f->dInhParent) hipFree(f->dInhParent);
  if (f->dInhLevel) hipFree(f->dInhLevel);
  hipFree(f->inhTmp.unused); hipFree(f->inhTmp.count); hipFree(f->inhTmp.sup); hipFree(f->inhTmp.chk); hipFree(f->inhTmp.mean); hipFree(f->inhTmp.cov);  // (hipFree(nullptr) is a no-op)
  hipFree(B.scan); hipFree(B.candMean); hipFree(B.candCov); hipFree(B.candSup); hipFree(B.candChk); hipFree(B.candCount);
  murty_free(f->Q, f->MS);
  if (f->hErr) hipHostFree(f->hErr);
  if (f->hJobCount) hipHostFree(f->hJobCount);
  if (f->hSums) hipHostFree(f->hSums);
  if (f->hWeights) hipHostFree(f->hWeights);
  if (f->hOutW) hipHostFree(f->hOutW);
  for (int k = 0; k < 4; k++) { if (f->hStage[k]) hipHostFree(f->hStage[k]); if (f->evStage[k]) hipEventDestroy(f->evStage[k]); }
  for (int k = 0; k < RFSGPU_ASYNC_RING; k++)
    for (int e = 0; e < 5; e++) if (f->ring[k][e]) hipEventDestroy(f->ring[k][e]);
  for (int k = 0; k < EV_COUNT; k++) if (f->ev[k]) hipEventDestroy(f->ev[k]);
  if (f->ownStream) hipStreamDestroy(f->ownStream);
  delete f;
}

const char *rfsgpu_last_error(const rfsgpu_filter *f) { return f ? f->err.c_str() : "null handle"; }

int rfsgpu_set_filter_config(rfsgpu_filter *f, const rfsgpu_filter_config *c) {
  CHECK_HANDLE(f);
  if (!c) return RFSGPU_ERR_INVALID;
  f->cfg = *c;
  rebuild_params(f);
  return RFSGPU_OK;
}
int rfsgpu_set_partition_mode(rfsgpu_filter *f, int mode) {
  CHECK_HANDLE(f);
  if (mode != RFSGPU_PARTITION_MURTY200 && mode != RFSGPU_PARTITION_EXACT) return fail(f, RFSGPU_ERR_INVALID, "unknown partition mode");
  f->P.exactPartitions = mode == RFSGPU_PARTITION_EXACT ? 1 : 0;
  return RFSGPU_OK;
}
// [test] The Murty-200 partition sums of given extended tables by the step's own post kernel (murty_jobs_kernel): what
// rfsMeasurementLikelihood's loop (include/RBPHDFilter.hpp:942-959) leaves in partition_likelihood for each table.  The tables
// go into the handle's job queue exactly as the weighting phase writes them (n x n row-major, n = nR + nC, slot k), the kernel
// runs once, the results come back; the particle weights are put back afterwards (the kernel multiplies the factors into them).
int rfsgpu_murty_partition_sums(rfsgpu_filter *f, const double *mats, const int *nR, const int *nC, int n_jobs, double *sums_out) {
  CHECK_HANDLE(f);
  if (!mats || !nR || !nC || !sums_out || n_jobs < 1 || n_jobs > f->Q.maxJobs) return fail(f, RFSGPU_ERR_INVALID, "murty_partition_sums: bad arguments");
  hipSetDevice(f->device);
  HIPCHK(hipStreamSynchronize(f->stream));
  std::vector<double> w((size_t)f->N);
  HIPCHK(hipMemcpy(w.data(), f->B.weight, (size_t)f->N * sizeof(double), hipMemcpyDeviceToHost));
  std::vector<MurtyJob> jobs((size_t)n_jobs);
  size_t off = 0;
  for (int k = 0; k < n_jobs; k++) {
    const int n = nR[k] + nC[k];
    if (nR[k] < 0 || nC[k] < 0 || n < 1 || n > MURTY_N) return fail(f, RFSGPU_ERR_INVALID, "murty_partition_sums: extended dimension out of range");
    jobs[k].particle = k % f->N; jobs[k].nR = nR[k]; jobs[k].nC = nC[k]; jobs[k].slot = k / f->N;
    HIPCHK(hipMemcpy(f->Q.mats + (size_t)k * MURTY_MAXN * MURTY_MAXN, mats + off, (size_t)n * n * sizeof(double), hipMemcpyHostToDevice));
    off += (size_t)n * n;
  }
  HIPCHK(hipMemcpy(f->Q.jobs, jobs.data(), (size_t)n_jobs * sizeof(MurtyJob), hipMemcpyHostToDevice));
  const int cnt[2] = {n_jobs, 0};
  HIPCHK(hipMemcpy(f->Q.count, cnt, sizeof(cnt), hipMemcpyHostToDevice));
  if (murty_launch(f->Q, f->MS, f->B, f->stream, nullptr, 0, nullptr, 0, f->hJobCount) != 0) return fail(f, RFSGPU_ERR_HIP, "murty launch failed");
  HIPCHK(hipMemcpyAsync(sums_out, f->Q.results, (size_t)n_jobs * sizeof(double), hipMemcpyDeviceToHost, f->stream));
  HIPCHK(hipMemcpyAsync(f->B.weight, w.data(), (size_t)f->N * sizeof(double), hipMemcpyHostToDevice, f->stream));
  return check_device_errors(f);
}
int rfsgpu_get_partition_mode(const rfsgpu_filter *f) { return f ? (f->P.exactPartitions ? RFSGPU_PARTITION_EXACT : RFSGPU_PARTITION_MURTY200) : -1; }
int rfsgpu_get_filter_config(const rfsgpu_filter *f, rfsgpu_filter_config *c) {
  if (!f || !c) return RFSGPU_ERR_INVALID;
  *c = f->cfg;
  return RFSGPU_OK;
}
int rfsgpu_set_model_rngbrg(rfsgpu_filter *f, const rfsgpu_rngbrg_config *c) {
  CHECK_HANDLE(f);
  if (!c) return RFSGPU_ERR_INVALID;
  if (f->model != RFSGPU_MODEL_RNGBRG_2D) return fail(f, RFSGPU_ERR_INVALID, "set_model_rngbrg on a handle created for another model");
  f->rb = *c;
  rebuild_params(f);
  return RFSGPU_OK;
}
int rfsgpu_set_model_victoriapark(rfsgpu_filter *f, const rfsgpu_vp_config *c) {
  CHECK_HANDLE(f);
  if (!c) return RFSGPU_ERR_INVALID;
  if (f->model != RFSGPU_MODEL_VICTORIAPARK_3D) return fail(f, RFSGPU_ERR_INVALID, "set_model_victoriapark on a handle created for another model");
  if (c->nPd < 1 || c->nPd > RFSGPU_VP_MAX_PD) return fail(f, RFSGPU_ERR_INVALID, "Pd table size out of range");
  f->vp = *c;
  rebuild_params(f);
  return RFSGPU_OK;
}
// MeasurementModel_VictoriaPark::setLaserScan (src/MeasurementModel_VictoriaPark.cpp:267-281): the FoV area / clutter
// intensity are a 361-term host sum; the raw scan goes to the device for the occlusion-based Pd.
int rfsgpu_set_laser_scan(rfsgpu_filter *f, const double *scan, int n) {
  CHECK_HANDLE(f);
  if (f->model != RFSGPU_MODEL_VICTORIAPARK_3D) return fail(f, RFSGPU_ERR_INVALID, "set_laser_scan needs the Victoria Park model");
  if (!scan || n < 2 || n > RFSGPU_VP_MAX_SCAN) return fail(f, RFSGPU_ERR_INVALID, "laser scan size out of range");
  double area = 0;
  for (int k = 1; k < n; k++) area += scan[k] * scan[k - 1];
  area += scan[0] * scan[n - 1];
  area *= sin(acos(-1) / 360) / 2;
  f->vpClutter = f->vp.expectedClutterNumber / area;
  hipSetDevice(f->device);
  HIPCHK(hipMemcpyAsync(f->B.scan, scan, (size_t)n * sizeof(double), hipMemcpyHostToDevice, f->stream));
  HIPCHK(hipStreamSynchronize(f->stream));
  f->B.nScan = n;
  rebuild_params(f);
  return RFSGPU_OK;
}
int rfsgpu_vp_probe_pd(rfsgpu_filter *f, int slot, double *pd, int *close_to_limit, int max_n) {
  CHECK_HANDLE(f);
  if (f->model != RFSGPU_MODEL_VICTORIAPARK_3D || slot < 0 || slot >= f->N || !pd || !close_to_limit) return fail(f, RFSGPU_ERR_INVALID, "vp_probe_pd: bad arguments");
  if (f->B.nScan < 2) return fail(f, RFSGPU_ERR_INVALID, "vp_probe_pd: rfsgpu_set_laser_scan first");
  hipSetDevice(f->device);
  const int n = std::min(max_n, f->cap);
  double *dPd = nullptr;
  int *dCl = nullptr;
  HIPCHK(hipMalloc(&dPd, (size_t)f->cap * sizeof(double)));
  if (hipMalloc(&dCl, (size_t)f->cap * sizeof(int)) != hipSuccess) { hipFree(dPd); return fail(f, RFSGPU_ERR_HIP, "vp_probe_pd: out of device memory"); }
  vp_probe_pd_kernel<<<1, 64, 0, f->stream>>>(f->B, f->P, f->cur, slot, dPd, dCl);
  hipError_t e1 = hipMemcpyAsync(pd, dPd, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, f->stream);
  hipError_t e2 = hipMemcpyAsync(close_to_limit, dCl, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, f->stream);
  hipError_t e3 = hipStreamSynchronize(f->stream);
  hipFree(dPd);
  hipFree(dCl);
  HIPCHK(e1); HIPCHK(e2); HIPCHK(e3);
  return RFSGPU_OK;
}
int rfsgpu_set_kf_config(rfsgpu_filter *f, const rfsgpu_kf_config *c) {
  CHECK_HANDLE(f);
  if (!c) return RFSGPU_ERR_INVALID;
  f->kf = *c;
  rebuild_params(f);
  return RFSGPU_OK;
}
int rfsgpu_set_lmk_process_noise(rfsgpu_filter *f, const double *Q) {
  CHECK_HANDLE(f);
  if (!Q) return RFSGPU_ERR_INVALID;
  memcpy(f->Qlm, Q, (size_t)f->D * f->D * sizeof(double));
  rebuild_params(f);
  return RFSGPU_OK;
}

// One slot of the pinned staging ring (four slots of Ncap * 12 + RFSGPU_VP_MAX_SCAN doubles, created on first use): waits until
// the copies issued from this slot four calls ago are done; the caller copies into *h, enqueues its host-to-device copies on the
// stream and records f->evStage[*k].
static int stage_slot(rfsgpu_filter *f, double **h, int *k_out) {
  const int k = f->stageNext;
  f->stageNext = (k + 1) & 3;
  const size_t slotDoubles = (size_t)f->Ncap * 22 + RFSGPU_VP_MAX_SCAN;   // poses | pose covariances | weights (or the 13-per-particle packed form) | covariances of the packed form's copy path | laser scan
  const bool fresh = !f->hStage[k] || !f->evStage[k];   // (either creation may have failed on an earlier call: each is retried on its own)
  if (!f->hStage[k]) HIPCHK(hipHostMalloc(&f->hStage[k], slotDoubles * sizeof(double)));
  if (!f->evStage[k]) HIPCHK(hipEventCreateWithFlags(&f->evStage[k], hipEventDisableTiming));
  if (fresh) HIPCHK(hipStreamSynchronize(f->stream));    // nothing recorded on this slot's event yet
  else HIPCHK(hipEventSynchronize(f->evStage[k]));       // the copies issued from this slot four calls ago
  *h = f->hStage[k];
  *k_out = k;
  return RFSGPU_OK;
}

// (r04: poses and weights go through the pinned staging ring of rfsgpu_set_step_inputs_async -- the caller's buffers are copied
//  before the call returns, the host-to-device copies are stream-ordered and nothing waits for the GPU.  The synchronous forms
//  these replaced -- a copy from pageable memory + a stream synchronisation each -- were 46 + 23 us of the 261 us that one
//  RBPHDFilter::update costs through the binding at configs[1], bench.py `boundary`.)
int rfsgpu_set_poses(rfsgpu_filter *f, const double *x, const double *cov, int cov_stride) {
  CHECK_HANDLE(f);
  if (!x || (cov_stride != 0 && cov_stride != 9)) return fail(f, RFSGPU_ERR_INVALID, "set_poses: bad arguments");
  return rfsgpu_set_step_inputs_async(f, x, cov, cov_stride, nullptr, 0);
}
int rfsgpu_get_poses(rfsgpu_filter *f, double *x) {
  CHECK_HANDLE(f);
  hipSetDevice(f->device);
  HIPCHK(hipMemcpyAsync(x, f->B.pose, (size_t)f->N * 3 * sizeof(double), hipMemcpyDeviceToHost, f->stream));
  HIPCHK(hipStreamSynchronize(f->stream));
  return RFSGPU_OK;
}
int rfsgpu_set_weights(rfsgpu_filter *f, const double *w) {
  CHECK_HANDLE(f);
  if (!w) return fail(f, RFSGPU_ERR_INVALID, "set_weights: null buffer");
  hipSetDevice(f->device);
  double *h = nullptr;
  int k = 0;
  { const int rc = stage_slot(f, &h, &k); if (rc != RFSGPU_OK) return rc; }
  memcpy(h, w, (size_t)f->N * sizeof(double));
  HIPCHK(hipMemcpyAsync(f->B.weight, h, (size_t)f->N * sizeof(double), hipMemcpyHostToDevice, f->stream));
  HIPCHK(hipEventRecord(f->evStage[k], f->stream));
  return RFSGPU_OK;
}
int rfsgpu_get_weights(rfsgpu_filter *f, double *w) {
  CHECK_HANDLE(f);
  if (!w) return fail(f, RFSGPU_ERR_INVALID, "get_weights: null buffer");
  hipSetDevice(f->device);
  if (!f->hWeights) HIPCHK(hipHostMalloc(&f->hWeights, (size_t)f->Ncap * sizeof(double)));   // pinned: a real DMA, no staging inside the runtime
  HIPCHK(hipMemcpyAsync(f->hWeights, f->B.weight, (size_t)f->N * sizeof(double), hipMemcpyDeviceToHost, f->stream));
  HIPCHK(hipStreamSynchronize(f->stream));
  memcpy(w, f->hWeights, (size_t)f->N * sizeof(double));
  return RFSGPU_OK;
}

int rfsgpu_gm_sizes(rfsgpu_filter *f, int *sizes) {
  CHECK_HANDLE(f);
  hipSetDevice(f->device);
  if (f->holes) {
    valid_count_kernel<<<(f->N + 3) / 4, 256, 0, f->stream>>>(f->B, f->cur, f->dSrcSlot);  // dSrcSlot doubles as int scratch
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(sizes, f->dSrcSlot, (size_t)f->N * sizeof(int), hipMemcpyDeviceToHost, f->stream));
  } else {
    HIPCHK(hipMemcpyAsync(sizes, f->B.count, (size_t)f->N * sizeof(int), hipMemcpyDeviceToHost, f->stream));
  }
  HIPCHK(hipStreamSynchronize(f->stream));
  return RFSGPU_OK;
}

// Host staging of one particle's planes (valid entries only: holes carry w < 0 between merge and prune).
static int fetch_particle(rfsgpu_filter *f, int slot, std::vector<double> &planes, int &n) {
  hipSetDevice(f->device);
  HIPCHK(hipMemcpyAsync(&n, f->B.count + slot, sizeof(int), hipMemcpyDeviceToHost, f->stream));
  HIPCHK(hipStreamSynchronize(f->stream));
  planes.resize((size_t)f->B.npl * f->cap);
  HIPCHK(hipMemcpyAsync(planes.data(), f->B.slab[f->cur] + (size_t)slot * f->B.npl * f->cap, planes.size() * sizeof(double),
                        hipMemcpyDeviceToHost, f->stream));
  HIPCHK(hipStreamSynchronize(f->stream));
  return RFSGPU_OK;
}

// plane index of mean component d / of covariance entry (r,c) in the slab layout of the handle's model
static inline int mean_plane(const rfsgpu_filter *f, int d) { return 2 + d; }
static inline int cov_plane(const rfsgpu_filter *f, int r, int c) {
  if (r > c) { int t = r; r = c; c = t; }
  if (f->D == 2) return PL_SXX + (r == 0 ? c : 2);          // xx, xy, yy
  static const int idx[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};  // xx, xy, xd, yy, yd, dd
  return P3_SXX + idx[r][c];
}

int rfsgpu_gm_size(rfsgpu_filter *f, int slot) {
  if (!f || slot < 0 || slot >= f->N) return -1;
  std::vector<double> pl;
  int n = 0;
  if (fetch_particle(f, slot, pl, n) != RFSGPU_OK) return -1;
  int k = 0;
  for (int m = 0; m < n; m++) if (!f->holes || pl[m] >= 0) k++;  // plane 0 = weight
  return k;
}

int rfsgpu_export_gm(rfsgpu_filter *f, int slot, int max_n, int *n_out, double *w, double *w_prev, double *mean, double *cov) {
  CHECK_HANDLE(f);
  if (slot < 0 || slot >= f->N) return fail(f, RFSGPU_ERR_INVALID, "export_gm: bad slot");
  std::vector<double> pl;
  int n = 0;
  int rc = fetch_particle(f, slot, pl, n);
  if (rc != RFSGPU_OK) return rc;
  const size_t c = f->cap;
  const int D = f->D;
  int k = 0;
  for (int m = 0; m < n; m++) {
    if (f->holes && pl[m] < 0) continue;  // hole left by merge
    if (k < max_n) {
      if (w) w[k] = pl[m];
      if (w_prev) w_prev[k] = pl[1 * c + m];
      if (mean) for (int d = 0; d < D; d++) mean[D * k + d] = pl[(size_t)mean_plane(f, d) * c + m];
      if (cov)
        for (int r = 0; r < D; r++)
          for (int q = 0; q < D; q++) cov[(size_t)D * D * k + D * r + q] = pl[(size_t)cov_plane(f, r, q) * c + m];
    }
    k++;
  }
  if (n_out) *n_out = k;
  return RFSGPU_OK;
}

int rfsgpu_get_landmark(rfsgpu_filter *f, int slot, int m, double *mean, double *cov, double *w) {
  CHECK_HANDLE(f);
  if (slot < 0 || slot >= f->N || m < 0) return RFSGPU_ERR_INVALID;
  hipSetDevice(f->device);
  int n = 0;
  HIPCHK(hipMemcpyAsync(&n, f->B.count + slot, sizeof(int), hipMemcpyDeviceToHost, f->stream));
  HIPCHK(hipStreamSynchronize(f->stream));
  if (m >= n) return RFSGPU_ERR_INVALID;
  double v[P3_COUNT];
  for (int pl = 0; pl < f->B.npl; pl++)
    HIPCHK(hipMemcpyAsync(&v[pl], f->B.slab[f->cur] + ((size_t)slot * f->B.npl + pl) * f->cap + m, sizeof(double), hipMemcpyDeviceToHost,
                          f->stream));
  HIPCHK(hipStreamSynchronize(f->stream));
  const int D = f->D;
  for (int d = 0; d < D; d++) mean[d] = v[mean_plane(f, d)];
  for (int r = 0; r < D; r++)
    for (int q = 0; q < D; q++) cov[D * r + q] = v[cov_plane(f, r, q)];
  *w = v[0];
  return RFSGPU_OK;
}

int rfsgpu_import_gm(rfsgpu_filter *f, int slot, int n, const double *w, const double *mean, const double *cov) {
  CHECK_HANDLE(f);
  if (slot < 0 || slot >= f->N || n < 0) return fail(f, RFSGPU_ERR_INVALID, "import_gm: bad arguments");
  if (n > f->cap) return fail(f, RFSGPU_ERR_CAPACITY, "import_gm: more Gaussians than gm_capacity");
  hipSetDevice(f->device);
  const size_t c = f->cap;
  const int D = f->D;
  std::vector<double> pl((size_t)f->B.npl * c, 0.0);
  for (int m = 0; m < n; m++) {
    pl[m] = w[m];
    pl[1 * c + m] = 0.0;
    for (int d = 0; d < D; d++) pl[(size_t)mean_plane(f, d) * c + m] = mean[D * m + d];
    for (int r = 0; r < D; r++)
      for (int q = r; q < D; q++) pl[(size_t)cov_plane(f, r, q) * c + m] = cov[(size_t)D * D * m + D * r + q];  // upper triangle
  }
  HIPCHK(hipMemcpyAsync(f->B.slab[f->cur] + (size_t)slot * f->B.npl * c, pl.data(), pl.size() * sizeof(double), hipMemcpyHostToDevice, f->stream));
  HIPCHK(hipMemcpyAsync(f->B.count + slot, &n, sizeof(int), hipMemcpyHostToDevice, f->stream));
  HIPCHK(hipStreamSynchronize(f->stream));
  return RFSGPU_OK;
}

int rfsgpu_import_aux(rfsgpu_filter *f, int slot, const int *unused_idx, int n_unused, int n_in_fov) {
  CHECK_HANDLE(f);
  if (slot < 0 || slot >= f->N || n_unused < 0 || (n_unused > 0 && !unused_idx)) return fail(f, RFSGPU_ERR_INVALID, "import_aux: bad arguments");
  unsigned long long m = 0;
  for (int k = 0; k < n_unused; k++) {
    if (unused_idx[k] < 0 || unused_idx[k] >= RFSGPU_MAX_Z) return fail(f, RFSGPU_ERR_INVALID, "import_aux: measurement index out of range");
    m |= 1ull << unused_idx[k];
  }
  hipSetDevice(f->device);
  HIPCHK(hipMemcpyAsync(f->B.unusedMask + slot, &m, sizeof(m), hipMemcpyHostToDevice, f->stream));
  HIPCHK(hipMemcpyAsync(f->B.nInFov + slot, &n_in_fov, sizeof(int), hipMemcpyHostToDevice, f->stream));
  HIPCHK(hipStreamSynchronize(f->stream));
  return RFSGPU_OK;
}

int rfsgpu_export_birth_candidates(rfsgpu_filter *f, int slot, int max_n, int *n_out, double *mean, double *cov, int *n_support, int *n_checks) {
  CHECK_HANDLE(f);
  if (slot < 0 || slot >= f->N) return fail(f, RFSGPU_ERR_INVALID, "export_birth_candidates: bad slot");
  hipSetDevice(f->device);
  int nc = 0;
  std::vector<double> m((size_t)RFSGPU_MAX_CANDIDATES * 3), c((size_t)RFSGPU_MAX_CANDIDATES * 6);
  std::vector<int> su(RFSGPU_MAX_CANDIDATES), ch(RFSGPU_MAX_CANDIDATES);
  HIPCHK(hipMemcpyAsync(&nc, f->B.candCount + slot, sizeof(int), hipMemcpyDeviceToHost, f->stream));
  HIPCHK(hipMemcpyAsync(m.data(), f->B.candMean + (size_t)slot * RFSGPU_MAX_CANDIDATES * 3, m.size() * sizeof(double), hipMemcpyDeviceToHost, f->stream));
  HIPCHK(hipMemcpyAsync(c.data(), f->B.candCov + (size_t)slot * RFSGPU_MAX_CANDIDATES * 6, c.size() * sizeof(double), hipMemcpyDeviceToHost, f->stream));
  HIPCHK(hipMemcpyAsync(su.data(), f->B.candSup + (size_t)slot * RFSGPU_MAX_CANDIDATES, su.size() * sizeof(int), hipMemcpyDeviceToHost, f->stream));
  HIPCHK(hipMemcpyAsync(ch.data(), f->B.candChk + (size_t)slot * RFSGPU_MAX_CANDIDATES, ch.size() * sizeof(int), hipMemcpyDeviceToHost, f->stream));
  HIPCHK(hipStreamSynchronize(f->stream));
  const int D = f->D;
  static const int idx3[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
  for (int k = 0; k < nc && k < max_n; k++) {
    for (int d = 0; d < D; d++) mean[D * k + d] = m[3 * k + d];
    for (int r = 0; r < D; r++)
      for (int q = 0; q < D; q++) cov[(size_t)D * D * k + D * r + q] = c[6 * k + idx3[r][q]];
    n_support[k] = su[k];
    n_checks[k] = ch[k];
  }
  if (n_out) *n_out = nc;
  return RFSGPU_OK;
}
int rfsgpu_import_birth_candidates(rfsgpu_filter *f, int slot, int n, const double *mean, const double *cov, const int *n_support, const int *n_checks) {
  CHECK_HANDLE(f);
  if (slot < 0 || slot >= f->N || n < 0 || n > RFSGPU_MAX_CANDIDATES) return fail(f, RFSGPU_ERR_INVALID, "import_birth_candidates: bad arguments");
  hipSetDevice(f->device);
  if (n > 0) f->candUsed = true;
  const int D = f->D;
  static const int idx3[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
  std::vector<double> m((size_t)RFSGPU_MAX_CANDIDATES * 3, 0.0), c((size_t)RFSGPU_MAX_CANDIDATES * 6, 0.0);
  std::vector<int> su(RFSGPU_MAX_CANDIDATES, 0), ch(RFSGPU_MAX_CANDIDATES, 0);
  for (int k = 0; k < n; k++) {
    for (int d = 0; d < D; d++) m[3 * k + d] = mean[D * k + d];
    for (int r = 0; r < D; r++)
      for (int q = r; q < D; q++) c[6 * k + idx3[r][q]] = cov[(size_t)D * D * k + D * r + q];
    su[k] = n_support[k];
    ch[k] = n_checks[k];
  }
  HIPCHK(hipMemcpyAsync(f->B.candMean + (size_t)slot * RFSGPU_MAX_CANDIDATES * 3, m.data(), m.size() * sizeof(double), hipMemcpyHostToDevice, f->stream));
  HIPCHK(hipMemcpyAsync(f->B.candCov + (size_t)slot * RFSGPU_MAX_CANDIDATES * 6, c.data(), c.size() * sizeof(double), hipMemcpyHostToDevice, f->stream));
  HIPCHK(hipMemcpyAsync(f->B.candSup + (size_t)slot * RFSGPU_MAX_CANDIDATES, su.data(), su.size() * sizeof(int), hipMemcpyHostToDevice, f->stream));
  HIPCHK(hipMemcpyAsync(f->B.candChk + (size_t)slot * RFSGPU_MAX_CANDIDATES, ch.data(), ch.size() * sizeof(int), hipMemcpyHostToDevice, f->stream));
  HIPCHK(hipMemcpyAsync(f->B.candCount + slot, &n, sizeof(int), hipMemcpyHostToDevice, f->stream));
  HIPCHK(hipStreamSynchronize(f->stream));
  return RFSGPU_OK;
}

// ---- the hot path ----------------------------------------------------------------------------------

static int set_lds_impl(rfsgpu_filter *f, const void *kernel, size_t bytes) {
  if ((long long)bytes > 160 * 1024) return fail(f, RFSGPU_ERR_CAPACITY, "kernel needs more than 160 KiB of LDS: lower gm_capacity");
  if (bytes > 48 * 1024) HIPCHK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return RFSGPU_OK;
}
#define set_lds(f, kernel, bytes) set_lds_impl(f, reinterpret_cast<const void *>(&kernel), bytes)

static int launch_update_map(rfsgpu_filter *f) {
  const int nZ = f->nZ;
  if (f->D == 3) {
    if (f->B.nScan < 2) return fail(f, RFSGPU_ERR_INVALID, "Victoria Park model: rfsgpu_set_laser_scan must precede the update");
    const size_t b = vp_shared_lds_bytes(nZ, f->B.nScan) + 2 * vp_update_lds_bytes_per_wave(f->cap);
    int rc3;
    if ((rc3 = set_lds(f, vp_update_map_kernel<2>, b)) != RFSGPU_OK) return rc3;
    vp_update_map_kernel<2><<<(f->N + 1) / 2, 128, b, f->stream>>>(f->B, f->P, f->cur, nZ);
    HIPCHK(hipGetLastError());
    return RFSGPU_OK;
  }
  auto bytes = [&](int wpb) { return (size_t)RFS_Z_LDS_BYTES + (size_t)wpb * update_map_lds_bytes_per_wave(f->cap); };
  int rc;
  // The workgroup form of the phase (the fused step's), several waves per particle: two up to capacity 512, four above (a pass
  // covers 64 landmarks per wave).  configs[1] (cap 384): 47.0 us with the one-wave kernel, 38.9 with two waves, 57.4 with three,
  // 43.7 with four; configs[2]'s shard (cap 640): 106.8 / 75.4 / 77.1 / 67.8.  RFSGPU_UPDMAP_WPP = 1 .. 4 overrides (1: the
  // one-wave kernels below).
  const size_t bb = (size_t)RFS_Z_LDS_BYTES + update_map_block_lds_bytes(f->cap);
  static const int wppEnv = [] { const char *e = getenv("RFSGPU_UPDMAP_WPP"); return e ? atoi(e) : 0; }();
  const int wppU = (wppEnv >= 1 && wppEnv <= 4) ? wppEnv : (f->cap > 512 ? 4 : 2);
  if (wppU != 1 && bb <= 64 * 1024) {
    if (wppU == 2) {
      if ((rc = set_lds(f, phd_update_map_block_kernel<2>, bb)) != RFSGPU_OK) return rc;
      phd_update_map_block_kernel<2><<<f->N, 128, bb, f->stream>>>(f->B, f->P, f->cur, nZ, f->B.Z);
    } else if (wppU == 3) {
      if ((rc = set_lds(f, phd_update_map_block_kernel<3>, bb)) != RFSGPU_OK) return rc;
      phd_update_map_block_kernel<3><<<f->N, 192, bb, f->stream>>>(f->B, f->P, f->cur, nZ, f->B.Z);
    } else {
      if ((rc = set_lds(f, phd_update_map_block_kernel<4>, bb)) != RFSGPU_OK) return rc;
      phd_update_map_block_kernel<4><<<f->N, 256, bb, f->stream>>>(f->B, f->P, f->cur, nZ, f->B.Z);
    }
    HIPCHK(hipGetLastError());
    return RFSGPU_OK;
  }
  if (bytes(4) <= 64 * 1024) {
    if ((rc = set_lds(f, phd_update_map_kernel<4>, bytes(4))) != RFSGPU_OK) return rc;
    phd_update_map_kernel<4><<<(f->N + 3) / 4, 256, bytes(4), f->stream>>>(f->B, f->P, f->cur, nZ, f->B.Z);
  } else {
    if ((rc = set_lds(f, phd_update_map_kernel<1>, bytes(1))) != RFSGPU_OK) return rc;
    phd_update_map_kernel<1><<<f->N, 64, bytes(1), f->stream>>>(f->B, f->P, f->cur, nZ, f->B.Z);
  }
  HIPCHK(hipGetLastError());
  return RFSGPU_OK;
}

static int eval_cap(const rfsgpu_filter *f) {
  int c = f->cfg.importanceWeightingEvalPointCount;
  if (c < 0 || c > RFSGPU_MAX_EVAL) c = RFSGPU_MAX_EVAL;
  if (c < 1) c = 1;
  return c;
}

#ifndef WEIGHT_WPP
#define WEIGHT_WPP 2
#endif
static int launch_weighting(rfsgpu_filter *f) {
  const int nZ = f->nZ, ec = eval_cap(f);
  const size_t per = weight_lds_bytes_per_wave(f->cap, ec, nZ);
  const int src = f->cur, dst = f->cur ^ 1;
  int rc;  // (the Murty job counter was cleared by stage_step_kernel)
  if (f->D == 3) {
    const size_t b = vp_shared_lds_bytes(nZ, f->B.nScan) + 2 * vp_weight_lds_bytes_per_wave(f->cap, ec, nZ);
    if ((rc = set_lds(f, vp_weighting_kernel<2>, b)) != RFSGPU_OK) return rc;
    vp_weighting_kernel<2><<<(f->N + 1) / 2, 128, b, f->stream>>>(f->B, f->P, src, dst, nZ, ec, f->Q);
    HIPCHK(hipGetLastError());
    f->cur = dst;
    if (murty_launch(f->Q, f->MS, f->B, f->stream, nullptr, 0, nullptr, 0, f->hJobCount) != 0) return fail(f, RFSGPU_ERR_HIP, "murty launch failed");
    return RFSGPU_OK;
  }
  {  // one workgroup of WEIGHT_WPP waves per particle
    const size_t b = (size_t)RFS_Z_LDS_BYTES + per + WEIGHT_SCRATCH_BYTES;
    if ((rc = set_lds(f, phd_weight_multifeature_kernel<WEIGHT_WPP>, b)) != RFSGPU_OK) return rc;
    phd_weight_multifeature_kernel<WEIGHT_WPP><<<f->N, WEIGHT_WPP * 64, b, f->stream>>>(f->B, f->P, src, dst, nZ, ec, f->Q);
  }
  HIPCHK(hipGetLastError());
  f->cur = dst;
  if (f->evAfterWeightKernel) HIPCHK(hipEventRecord(f->evAfterWeightKernel, f->stream));
  // Murty-200 for partitions with nR + nC > 8: runs only when the queue is non-empty (device-side early exit)
  rc = murty_launch(f->Q, f->MS, f->B, f->stream, nullptr, 0, nullptr, 0, f->hJobCount);
  if (rc != 0) return fail(f, RFSGPU_ERR_HIP, "murty launch failed");
  return RFSGPU_OK;
}

}  // extern "C" (templates need C++ linkage)
#ifndef MERGE_WPP
#define MERGE_WPP 2
#endif
template <bool FUSE>
static int launch_merge_t(rfsgpu_filter *f) {
  const int cur = f->cur, dst = f->cur ^ 1;
  int rc;
  if (f->D == 3) {
    const size_t b = vp_merge_lds_bytes_per_wave(f->cap);
    if ((rc = set_lds(f, (vp_merge_kernel<1, FUSE>), b)) != RFSGPU_OK) return rc;
    vp_merge_kernel<1, FUSE><<<f->N, 64, b, f->stream>>>(f->B, f->P, cur, dst);
    HIPCHK(hipGetLastError());
    if (FUSE) f->cur = dst;
    return RFSGPU_OK;
  }
  // one workgroup of MERGE_WPP waves per particle
  const size_t per = merge_lds_bytes_per_block(f->cap, MERGE_WPP);
  if ((rc = set_lds(f, (gm_merge_kernel<MERGE_WPP, FUSE>), per)) != RFSGPU_OK) return rc;
  gm_merge_kernel<MERGE_WPP, FUSE><<<f->N, MERGE_WPP * 64, per, f->stream>>>(f->B, f->P, cur, dst);
  HIPCHK(hipGetLastError());
  if (FUSE) f->cur = dst;
  return RFSGPU_OK;
}
extern "C" {
static int launch_merge(rfsgpu_filter *f) { return launch_merge_t<false>(f); }
static int launch_merge_prune(rfsgpu_filter *f) { return launch_merge_t<true>(f); }

static int launch_prune(rfsgpu_filter *f) {
  const size_t per = gm_prune_lds_bytes_per_wave(f->cap);
  const int src = f->cur, dst = f->cur ^ 1;
  int rc;
  if ((rc = set_lds(f, gm_prune_kernel<4>, 4 * per)) != RFSGPU_OK) return rc;
  gm_prune_kernel<4><<<(f->N + 3) / 4, 256, 4 * per, f->stream>>>(f->B, f->P, src, dst);
  HIPCHK(hipGetLastError());
  f->cur = dst;
  return RFSGPU_OK;
}

static int stage_measurements(rfsgpu_filter *f, const double *z, int n_z) {
  if (n_z < 0 || n_z > RFSGPU_MAX_Z) return fail(f, RFSGPU_ERR_INVALID, "at most RFSGPU_MAX_Z measurements per update");
  if (n_z > 0 && !z) return fail(f, RFSGPU_ERR_INVALID, "null measurement buffer");
  hipSetDevice(f->device);
  {
    // The measurement set (<= 1.5 KB) travels in the kernel-argument block of one tiny kernel that also clears the Murty
    // queue for this step: no staging buffer, no copy-engine hop between compute kernels, and the caller's buffer is
    // free again when this call returns.
    ZArg za;
    const int nd = n_z * f->D;
    if (nd > 0) memcpy(za.v, z, (size_t)nd * sizeof(double));
    stage_step_kernel<<<1, 256, 0, f->stream>>>(za, f->B.Z, nd, f->Q.count);
    HIPCHK(hipGetLastError());
  }
  f->nZ = n_z;
  if (n_z > 0) f->resampleOccured = false;   // RBPHDFilter.hpp:526 (an update with measurements; its resampling decision follows)
  return RFSGPU_OK;
}

int rfsgpu_update_map(rfsgpu_filter *f, const double *z, int n_z) {
  CHECK_HANDLE(f);
  long long t0 = now_ns();
  int rc = stage_measurements(f, z, n_z);
  if (rc != RFSGPU_OK) return rc;
  HIPCHK(hipEventRecord(f->ev[EV_UM0], f->stream));
  if ((rc = launch_update_map(f)) != RFSGPU_OK) return rc;
  HIPCHK(hipEventRecord(f->ev[EV_UM1], f->stream));
  rc = check_device_errors(f);
  accumulate(f->ev[EV_UM0], f->ev[EV_UM1], f->timing.mapUpdate_wall, &f->lastKernelNs[0]);
  f->timing.mapUpdate_kf_wall = f->timing.mapUpdate_wall;  // KF correct is fused into the same kernel
  f->timing.mapUpdate_cpu += now_ns() - t0;
  return rc;
}
int rfsgpu_importance_weighting(rfsgpu_filter *f) {
  CHECK_HANDLE(f);
  long long t0 = now_ns();
  hipSetDevice(f->device);
  HIPCHK(hipEventRecord(f->ev[EV_UM1], f->stream));
  int rc = launch_weighting(f);
  if (rc != RFSGPU_OK) return rc;
  HIPCHK(hipEventRecord(f->ev[EV_W1], f->stream));
  rc = check_device_errors(f);
  accumulate(f->ev[EV_UM1], f->ev[EV_W1], f->timing.particleWeighting_wall, &f->lastKernelNs[1]);
  f->timing.particleWeighting_cpu += now_ns() - t0;
  return rc;
}
int rfsgpu_merge(rfsgpu_filter *f) {
  CHECK_HANDLE(f);
  f->holes = true;
  long long t0 = now_ns();
  hipSetDevice(f->device);
  HIPCHK(hipEventRecord(f->ev[EV_W1], f->stream));
  int rc = launch_merge(f);
  if (rc != RFSGPU_OK) return rc;
  HIPCHK(hipEventRecord(f->ev[EV_MG1], f->stream));
  HIPCHK(hipStreamSynchronize(f->stream));
  accumulate(f->ev[EV_W1], f->ev[EV_MG1], f->timing.mapMerge_wall, &f->lastKernelNs[2]);
  f->timing.mapMerge_cpu += now_ns() - t0;
  return RFSGPU_OK;
}
int rfsgpu_prune(rfsgpu_filter *f) {
  CHECK_HANDLE(f);
  f->holes = false;
  long long t0 = now_ns();
  hipSetDevice(f->device);
  HIPCHK(hipEventRecord(f->ev[EV_MG1], f->stream));
  int rc = launch_prune(f);
  if (rc != RFSGPU_OK) return rc;
  HIPCHK(hipEventRecord(f->ev[EV_PR1], f->stream));
  HIPCHK(hipStreamSynchronize(f->stream));
  accumulate(f->ev[EV_MG1], f->ev[EV_PR1], f->timing.mapPrune_wall, &f->lastKernelNs[3]);
  f->timing.mapPrune_cpu += now_ns() - t0;
  return RFSGPU_OK;
}

// All four phases back to back on the stream, ONE host sync at the end (RBPHDFilter::update body :444-523).
static const StepOut NO_OUT{nullptr, nullptr, 0, nullptr, nullptr, 0, 0};
static const StepPredict NO_HEAD{0, 0, nullptr, nullptr, 0};
static int vp_order_buffers(rfsgpu_filter *f);
static int update_async_impl(rfsgpu_filter *f, const double *z, int n_z, bool with_sums, int normalize, const StepPredict &sp = NO_HEAD, const StepOut &so = NO_OUT,
                             hipEvent_t waitBeforePost = nullptr);
int rfsgpu_update(rfsgpu_filter *f, const double *z, int n_z) {
  CHECK_HANDLE(f);
  f->holes = false;
  if (n_z == 0) return RFSGPU_OK;  // :450-452
  // 2-D model: ONE fused launch (+ the post kernel) unless the caller asked for phase-resolved timing (rfsgpu_set_phase_timing):
  // the same bits either way, 26 % less device time at configs[1]; the whole step is then booked under TimingInfo::mapUpdate
  if (f->fuseSteps && !f->phaseTiming) {   // (r3: the Victoria Park model too -- vp_step_fused_kernel)
    const int rc = update_async_impl(f, z, n_z, false, 0);
    return rc != RFSGPU_OK ? rc : rfsgpu_synchronize(f);
  }
  long long t0 = now_ns();
  int rc = stage_measurements(f, z, n_z);
  if (rc != RFSGPU_OK) return rc;
  HIPCHK(hipEventRecord(f->ev[EV_UM0], f->stream));
  if ((rc = launch_update_map(f)) != RFSGPU_OK) return rc;
  HIPCHK(hipEventRecord(f->ev[EV_UM1], f->stream));
  if (!f->cfg.useClusterProcess) {
    if ((rc = launch_weighting(f)) != RFSGPU_OK) return rc;
  }
  HIPCHK(hipEventRecord(f->ev[EV_W1], f->stream));
  // merge + prune fused in one kernel (the mapPrune bucket stays 0 on this path; rfsgpu_merge / rfsgpu_prune keep
  // the two-kernel form for phase-by-phase use)
  if ((rc = launch_merge_prune(f)) != RFSGPU_OK) return rc;
  HIPCHK(hipEventRecord(f->ev[EV_MG1], f->stream));
  HIPCHK(hipEventRecord(f->ev[EV_PR1], f->stream));
  rc = check_device_errors(f);  // syncs
  accumulate(f->ev[EV_UM0], f->ev[EV_UM1], f->timing.mapUpdate_wall, &f->lastKernelNs[0]);
  f->timing.mapUpdate_kf_wall = f->timing.mapUpdate_wall;
  accumulate(f->ev[EV_UM1], f->ev[EV_W1], f->timing.particleWeighting_wall, &f->lastKernelNs[1]);
  accumulate(f->ev[EV_W1], f->ev[EV_MG1], f->timing.mapMerge_wall, &f->lastKernelNs[2]);
  accumulate(f->ev[EV_MG1], f->ev[EV_PR1], f->timing.mapPrune_wall, &f->lastKernelNs[3]);
  f->timing.mapUpdate_cpu += now_ns() - t0;
  return rc;
}

// Does the fused step being launched carry the timing events?  Every timingStride-th one does (the first of a filter always); a
// timed ring entry stands for itself + the untimed steps launched since the previous booking (2-D and Victoria Park steps alike).
static bool step_carries_events(rfsgpu_filter *f) {
  const bool timed = (f->stepSeq++ % (unsigned)f->timingStride) == 0;
  if (timed) { f->ringStands[f->ringCount] = f->untimedSince + 1; f->untimedSince = 0; }
  else f->untimedSince++;
  return timed;
}

// Fold the event pairs of the async steps recorded since the last harvest into TimingInfo / the kernel statistics.
// Caller has synchronised the stream.
static void harvest_async(rfsgpu_filter *f) {
  const int n = f->ringCount < RFSGPU_ASYNC_RING ? f->ringCount : RFSGPU_ASYNC_RING;
  for (int k = 0; k < n; k++) {
    hipEvent_t *e = f->ring[k];
    long long ns[3] = {0, 0, 0};
    if (f->ringFused[k]) {  // one kernel for the whole step: booked under mapUpdate, reported as kernel 0
      accumulate(e[0], e[3], f->timing.mapUpdate_wall, &ns[0]);
      // (only every timingStride-th fused step carries events: TimingInfo books the sampled step once for each step it stands for)
      if (f->ringStands[k] > 1) f->timing.mapUpdate_wall += ns[0] * (long long)(f->ringStands[k] - 1);
      for (int q = 0; q < 3; q++) { f->statNs[q] += (double)ns[q]; f->lastKernelNs[q] = ns[q]; }
      f->lastFusedNs = ns[0];
      {   // the post kernel (Murty jobs if any, queue reset, weight sums / division): event 1 is free on this path
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e[3], e[1]) == hipSuccess) f->statPostNs += (double)ms * 1.0e6;
      }
      f->statSteps++;
      continue;
    }
    accumulate(e[0], e[1], f->timing.mapUpdate_wall, &ns[0]);
    accumulate(e[1], e[2], f->timing.particleWeighting_wall, &ns[1]);
    if (f->ringHasMid[k]) {  // kernel statistics: the weighting kernel alone, without the (usually empty) Murty launch
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, e[1], e[4]) == hipSuccess) ns[1] = (long long)(ms * 1.0e6);
    }
    accumulate(e[2], e[3], f->timing.mapMerge_wall, &ns[2]);
    for (int q = 0; q < 3; q++) { f->statNs[q] += (double)ns[q]; f->lastKernelNs[q] = ns[q]; }
    f->statSteps++;
  }
  // un-sampled steps launched since the last sampled one: booked now at the last sample's duration (the stream is drained, they have
  // run), so that TimingInfo covers every step at every harvest instead of trailing by up to timingStride - 1 steps
  if (f->untimedSince > 0 && f->lastFusedNs > 0) {
    f->timing.mapUpdate_wall += f->lastFusedNs * (long long)f->untimedSince;
    f->untimedSince = 0;
  }
  f->timing.mapUpdate_kf_wall = f->timing.mapUpdate_wall;
  f->ringCount = 0;
}

// Shared body of the stream-ordered steps.  with_sums: the step's post kernel also leaves {sum w, sum w^2} in the bound sums
// buffer (and divides the weights by the sum when normalize != 0).
static int update_async_impl(rfsgpu_filter *f, const double *z, int n_z, bool with_sums, int normalize, const StepPredict &sp, const StepOut &so,
                             hipEvent_t waitBeforePost) {
  f->holes = false;
  if (n_z == 0) return RFSGPU_OK;  // :450-452
  long long t0 = now_ns();
  if (f->ringCount >= RFSGPU_ASYNC_RING) {  // ring full: drain (one sync per RFSGPU_ASYNC_RING steps)
    hipSetDevice(f->device);
    HIPCHK(hipStreamSynchronize(f->stream));
    harvest_async(f);
  }
  int rc;
  hipEvent_t *e = f->ring[f->ringCount];
  f->ringFused[f->ringCount] = false;
  if (f->D == 2 && f->fuseSteps) {
    // the whole step in one launch (step_fused.h), the measurement set riding in its kernel arguments; then the post kernel:
    // Murty partitions (if any), queue reset, weight sums / normalisation
    if (n_z < 0 || n_z > RFSGPU_MAX_Z) return fail(f, RFSGPU_ERR_INVALID, "at most RFSGPU_MAX_Z measurements per update");
    if (!z) return fail(f, RFSGPU_ERR_INVALID, "null measurement buffer");
    hipSetDevice(f->device);
    ZArg za;
    memcpy(za.v, z, (size_t)n_z * 2 * sizeof(double));
    f->nZ = n_z;
    if (n_z > 0) f->resampleOccured = false;
    const bool timed = step_carries_events(f);   // (rfsgpu_set_step_timing_stride)
    if (timed) HIPCHK(hipEventRecord(e[0], f->stream));
    const int ec = eval_cap(f), useW = f->cfg.useClusterProcess ? 0 : 1;
    // Waves per particle: two, unless the two-wave grid cannot be resident at once (large mixtures: the LDS block limits the
    // workgroups per CU) -- then three waves per particle finish each workgroup sooner and free its slot (measured on the
    // configs[2] shard, 2500 x 500 x 30 at cap 640: fused step 327 -> 263 us; at C2a, where all 2000 two-wave workgroups are
    // resident, three waves lose: 133 -> 180 us).  RFSGPU_STEP_WPP = 2 | 3 overrides.
    const size_t b2 = step_fused_lds_total(f->cap, ec, f->nZ, 2);
    const int perCU2 = (int)std::min<size_t>(8, b2 ? (size_t)(160 * 1024) / b2 : 8);
    // (three waves only where it is the LDS block that keeps two-wave workgroups below the 8 per CU the wave slots allow: at
    //  capacity 384 -- 9 blocks of LDS per CU -- a launch of more than 2048 particles is better off with two waves per particle
    //  in several rounds: 3000 particles 229 us against 389, 8000: 474 against 882)
    int wpp = ((long long)perCU2 * f->nCU >= f->N || perCU2 >= 8) ? 2 : 3;
    {  // (round 5) a launch small enough for every THREE-wave workgroup to be resident at once takes three waves per particle: a
       // particle's own chain is what such a launch lasts, and the third wave shortens it -- fused step at configs[1]'s shape with
       // 1000 / 500 / 250 particles 103.6 / 95.4 / 102.4 -> 99.3 / 86.1 / 94.6 us; at 2000 particles (1280 three-wave workgroups
       // resident) two waves stay: 112 against 180 us
      const size_t b3 = step_fused_lds_total(f->cap, ec, f->nZ, 3);
      const int perCU3 = (int)std::min<size_t>(16 / 3, b3 ? (size_t)(160 * 1024) / b3 : 16 / 3);
      if ((long long)perCU3 * f->nCU >= f->N) wpp = 3;
    }
    if (f->stepWppOverride == 2 || f->stepWppOverride == 3) wpp = f->stepWppOverride;
    const size_t b = step_fused_lds_total(f->cap, ec, f->nZ, wpp);
    // phase priorities only when every workgroup is resident at once: 16 waves per CU at 128 VGPRs, LDS permitting
    const int perCU = (int)std::min<size_t>(16 / wpp, b ? (size_t)(160 * 1024) / b : 16);
    // (RFSGPU_STEP_PHASE_PRIO = 0 | 1: tuning override of the choice below)
    static const int prioOverride = [] { const char *e = getenv("RFSGPU_STEP_PHASE_PRIO"); return e ? atoi(e) : -1; }();
    const int phasePrio = prioOverride >= 0 ? (prioOverride ? 1 : 0) : ((long long)perCU * f->nCU >= f->N ? 1 : 0);
    f->lastStepVariant[0] = wpp; f->lastStepVariant[1] = phasePrio; f->lastStepVariant[2] = 5; f->lastStepVariant[3] = sp.mode ? 2 : (sp.inMask ? 3 : 1);   // ([3]: 1 = fused step, 2 = with the predict at its head, 3 = with the input pull only)
    // (one instantiation per {waves per particle, phase priorities, merge grid} x {plain step, step with the predict at its head})
    // cost-ordered launch for the instantiations without phase priorities (several rounds of workgroups; step_fused.h StepLaunchOrder)
    if ((rc = vp_order_buffers(f)) != RFSGPU_OK) return rc;
    const StepLaunchOrder sloOn{f->vpOrderMode ? f->vpCost : nullptr, f->vpOrderMode ? f->vpOrder : nullptr}, sloOff{nullptr, nullptr};
    bool orderedLaunch = false;
#define STEP_LAUNCH(WPPV, PRIO, GLV, BYTES)                                                                                            \
    do {                                                                                                                                \
      if (sp.mode || sp.inMask) {                                                                                                        \
        if ((rc = set_lds(f, (phd_step_fused_kernel<WPPV, PRIO, GLV, true>), BYTES)) != RFSGPU_OK) return rc;                           \
        phd_step_fused_kernel<WPPV, PRIO, GLV, true><<<f->N, WPPV * 64, BYTES, f->stream>>>(f->B, f->P, f->cur, f->nZ, ec, useW, f->Q, za, sp, PRIO ? sloOff : sloOn);  \
      } else {                                                                                                                          \
        if ((rc = set_lds(f, (phd_step_fused_kernel<WPPV, PRIO, GLV, false>), BYTES)) != RFSGPU_OK) return rc;                          \
        phd_step_fused_kernel<WPPV, PRIO, GLV, false><<<f->N, WPPV * 64, BYTES, f->stream>>>(f->B, f->P, f->cur, f->nZ, ec, useW, f->Q, za, sp, PRIO ? sloOff : sloOn); \
      }                                                                                                                                 \
      orderedLaunch = !(PRIO);                                                                                                          \
    } while (0)
    if (wpp == 2) {
      if (phasePrio) STEP_LAUNCH(2, true, 5, b);
      else STEP_LAUNCH(2, false, 5, b);
    } else {
      // the 64 x 64 merge grid where its 6 KB of extra cursors leave as many three-wave workgroups resident (merge_prune.h);
      // RFSGPU_MERGE_GRID = 5 | 6 overrides
      const size_t b6 = step_fused_lds_total(f->cap, ec, f->nZ, 3, 6);
      const int perCU6 = (int)std::min<size_t>(16 / 3, b6 ? (size_t)(160 * 1024) / b6 : 16);
      const bool fine = f->mergeGridOverride ? f->mergeGridOverride == 6 : (perCU6 == perCU && perCU6 > 0);
      if (fine) {
        const int prio6 = (long long)perCU6 * f->nCU >= f->N ? 1 : 0;
        f->lastStepVariant[1] = prio6; f->lastStepVariant[2] = 6;
        if (prio6) STEP_LAUNCH(3, true, 6, b6);
        else STEP_LAUNCH(3, false, 6, b6);
      } else {
        if (phasePrio) STEP_LAUNCH(3, true, 5, b);
        else STEP_LAUNCH(3, false, 5, b);
      }
    }
#undef STEP_LAUNCH
    HIPCHK(hipGetLastError());
    if (timed) HIPCHK(hipEventRecord(e[3], f->stream));
    if (waitBeforePost) HIPCHK(hipStreamWaitEvent(f->stream, waitBeforePost, 0));   // (the collective that produced so.preDiv, on another stream)
    const bool sortCosts = orderedLaunch && f->vpOrderMode == 2;
    const StepOrderArg sord{sortCosts ? f->vpCost : nullptr, f->vpOrder, f->N, f->vpCost + f->Ncap, f->vpParity};
    if (sortCosts) f->vpParity ^= 1;
    if (murty_launch(f->Q, f->MS, f->B, f->stream, with_sums ? f->dSums : nullptr, normalize, &za, 2 * n_z, f->hJobCount, so, sord) != 0) return fail(f, RFSGPU_ERR_HIP, "post kernel launch failed");
    if (timed) HIPCHK(hipEventRecord(e[1], f->stream));
    f->cur ^= 1;  // the map update works in place, the weighting phase leaves only a permutation in LDS, merge + prune write the other slab
    if (timed) {
      f->ringFused[f->ringCount] = true;
      f->ringCount++;
    }
    f->timing.mapUpdate_cpu += now_ns() - t0;
    return RFSGPU_OK;
  }
  if (f->D == 3 && f->fuseSteps) {
    // Victoria Park: the whole step in one launch (vp.h, vp_step_fused_kernel: one wavefront per particle through updateMap ->
    // importanceWeighting -> merge + prune, the sorted order a permutation in LDS), then the post kernel as for the 2-D model
    if (n_z < 0 || n_z > RFSGPU_MAX_Z) return fail(f, RFSGPU_ERR_INVALID, "at most RFSGPU_MAX_Z measurements per update");
    if (!z) return fail(f, RFSGPU_ERR_INVALID, "null measurement buffer");
    if (f->B.nScan < 2) return fail(f, RFSGPU_ERR_INVALID, "Victoria Park model: rfsgpu_set_laser_scan must precede the update");
    hipSetDevice(f->device);
    ZArg za;
    memcpy(za.v, z, (size_t)n_z * 3 * sizeof(double));
    f->nZ = n_z;
    if (n_z > 0) f->resampleOccured = false;
    const bool timed = step_carries_events(f);
    if (timed) HIPCHK(hipEventRecord(e[0], f->stream));
    const int ec = eval_cap(f), useW = f->cfg.useClusterProcess ? 0 : 1;
    const size_t shared = vp_shared_lds_bytes(f->nZ, f->B.nScan), per = vp_step_lds_bytes_per_wave(f->cap, ec, f->nZ);
    f->lastStepVariant[0] = 1; f->lastStepVariant[1] = 0; f->lastStepVariant[2] = 0; f->lastStepVariant[3] = 1;
    // cost-ordered launch (round 6, murty.h step_cost_order_class): this step's particles in the order the previous step's post kernel
    // left -- longest first --, this step's durations captured for the next one
    if ((rc = vp_order_buffers(f)) != RFSGPU_OK) return rc;
    const int *vpOrd = f->vpOrderMode ? f->vpOrder : nullptr;
    const StepOrderArg sord{f->vpOrderMode == 2 ? f->vpCost : nullptr, f->vpOrder, f->N, f->vpCost + f->Ncap, f->vpParity};
    if (f->vpOrderMode == 2) f->vpParity ^= 1;
    if (shared + 2 * per <= (size_t)64 * 1024) {
      const size_t b = shared + 2 * per;
      if ((rc = set_lds(f, vp_step_fused_kernel<2>, b)) != RFSGPU_OK) return rc;
      vp_step_fused_kernel<2><<<(f->N + 1) / 2, 128, b, f->stream>>>(f->B, f->P, f->cur, f->nZ, ec, useW, f->Q, za, f->vpCost, vpOrd);
    } else {
      const size_t b = shared + per;
      if ((rc = set_lds(f, vp_step_fused_kernel<1>, b)) != RFSGPU_OK) return rc;
      vp_step_fused_kernel<1><<<f->N, 64, b, f->stream>>>(f->B, f->P, f->cur, f->nZ, ec, useW, f->Q, za, f->vpCost, vpOrd);
    }
    HIPCHK(hipGetLastError());
    if (timed) HIPCHK(hipEventRecord(e[3], f->stream));
    if (waitBeforePost) HIPCHK(hipStreamWaitEvent(f->stream, waitBeforePost, 0));
    if (murty_launch(f->Q, f->MS, f->B, f->stream, with_sums ? f->dSums : nullptr, normalize, &za, 3 * n_z, f->hJobCount, so, sord) != 0) return fail(f, RFSGPU_ERR_HIP, "post kernel launch failed");
    if (timed) HIPCHK(hipEventRecord(e[1], f->stream));
    f->cur ^= 1;
    if (timed) {
      f->ringFused[f->ringCount] = true;
      f->ringCount++;
    }
    f->timing.mapUpdate_cpu += now_ns() - t0;
    return RFSGPU_OK;
  }
  f->lastStepVariant[0] = f->lastStepVariant[1] = f->lastStepVariant[2] = f->lastStepVariant[3] = 0;   // three kernels
  rc = stage_measurements(f, z, n_z);
  if (rc != RFSGPU_OK) return rc;
  HIPCHK(hipEventRecord(e[0], f->stream));
  if ((rc = launch_update_map(f)) != RFSGPU_OK) return rc;
  HIPCHK(hipEventRecord(e[1], f->stream));
  f->ringHasMid[f->ringCount] = false;
  if (!f->cfg.useClusterProcess) {
    f->evAfterWeightKernel = (f->D == 2) ? e[4] : nullptr;
    rc = launch_weighting(f);
    f->ringHasMid[f->ringCount] = f->evAfterWeightKernel != nullptr;
    f->evAfterWeightKernel = nullptr;
    if (rc != RFSGPU_OK) return rc;
  }
  HIPCHK(hipEventRecord(e[2], f->stream));
  if ((rc = launch_merge_prune(f)) != RFSGPU_OK) return rc;
  HIPCHK(hipEventRecord(e[3], f->stream));
  f->ringCount++;
  f->timing.mapUpdate_cpu += now_ns() - t0;
  if (with_sums) {
    if ((rc = rfsgpu_weight_sums_async(f)) != RFSGPU_OK) return rc;
    if (normalize) return rfsgpu_normalize_weights(f, 0.0, f->dSums);
  }
  return RFSGPU_OK;
}
int rfsgpu_update_async(rfsgpu_filter *f, const double *z, int n_z) {
  CHECK_HANDLE(f);
  return update_async_impl(f, z, n_z, false, 0);
}
int rfsgpu_step_async(rfsgpu_filter *f, const double *z, int n_z, int normalize) {
  CHECK_HANDLE(f);
  if (n_z == 0) {  // no update (:450-452), but the weights are still summed / normalised as the caller asked
    int rc = rfsgpu_weight_sums_async(f);
    if (rc != RFSGPU_OK) return rc;
    return normalize ? rfsgpu_normalize_weights(f, 0.0, f->dSums) : RFSGPU_OK;
  }
  return update_async_impl(f, z, n_z, true, normalize);
}

// The step of a multi-GPU host whose weight normalisation trails by one step (round 5).  As rfsgpu_step_async
// with normalize = 0 -- this shard's {sum w, sum w^2} go to the bound sums buffer -- except that the post kernel first divides
// the weights by *prev_total_dev, the all-reduced total of the PREVIOUS step (NULL: no division), and waits for `wait_event` (a
// hipEvent_t recorded on ANOTHER stream behind the collective that produced that total; NULL: none) only between the step kernel
// and the post kernel: the collective of step k runs beside the step kernel of step k + 1 instead of in front of it.
int rfsgpu_step_async_deferred(rfsgpu_filter *f, const double *z, int n_z, const void *prev_total_dev, void *wait_event) {
  CHECK_HANDLE(f);
  hipSetDevice(f->device);
  if (n_z == 0 || !f->fuseSteps || f->phaseTiming) {        // no fused step: the same arithmetic with the stand-alone kernels, in stream order
    if (wait_event) HIPCHK(hipStreamWaitEvent(f->stream, (hipEvent_t)wait_event, 0));
    if (n_z > 0) { const int rc = update_async_impl(f, z, n_z, false, 0); if (rc != RFSGPU_OK) return rc; }
    if (prev_total_dev) { const int rc = rfsgpu_normalize_weights(f, 0.0, prev_total_dev); if (rc != RFSGPU_OK) return rc; }
    return rfsgpu_weight_sums_async(f);
  }
  StepOut so = NO_OUT;
  so.preDiv = reinterpret_cast<const double *>(prev_total_dev);
  return update_async_impl(f, z, n_z, true, 0, NO_HEAD, so, (hipEvent_t)wait_event);
}

// The same without stream events.  The caller pairs every call with rfsgpu_collective_gate(side stream) -> its collective over the
// shards' sums into total_dev -> rfsgpu_collective_publish(side stream): the gate kernel waits on the device for this step's post
// kernel to have written the sums, the publish kernel raises the number the NEXT step's post kernel waits for (on the device, just
// before it divides by total_dev).  have_prev == 0: the first step of a run (or the first after a flush): nothing to divide by.
int rfsgpu_step_async_trailing(rfsgpu_filter *f, const double *z, int n_z, const void *total_dev, int have_prev) {
  CHECK_HANDLE(f);
  hipSetDevice(f->device);
  if (!total_dev) return fail(f, RFSGPU_ERR_INVALID, "step_async_trailing: null total buffer");
  const int k = ++f->collStep;
  if (n_z == 0 || !f->fuseSteps || f->phaseTiming || f->D != 2) {   // no 2-D fused step: stand-alone kernels in stream order, the hand-over by the same two small kernels
    if (have_prev) coll_gate_kernel<<<1, 64, 0, f->stream>>>(f->dCollSeq + 1, k - 1, f->B.err);
    if (n_z > 0) { const int rc = update_async_impl(f, z, n_z, false, 0); if (rc != RFSGPU_OK) return rc; }
    if (have_prev) { const int rc = rfsgpu_normalize_weights(f, 0.0, total_dev); if (rc != RFSGPU_OK) return rc; }
    const int rc = rfsgpu_weight_sums_async(f);
    coll_publish_kernel<<<1, 1, 0, f->stream>>>(f->dCollSeq, k);         // word [0]: "the sums of step k are out"
    HIPCHK(hipGetLastError());
    return rc;
  }
  StepOut so = NO_OUT;
  so.preDiv = have_prev ? reinterpret_cast<const double *>(total_dev) : nullptr;
  so.collSeq = f->dCollSeq; so.collNeed = have_prev ? k - 1 : 0; so.collPost = k;
  return update_async_impl(f, z, n_z, true, 0, NO_HEAD, so, nullptr);
}
// [multi] on `hip_stream` (the side stream of the collective): wait, on the device, until the post kernel of the last
// rfsgpu_step_async_trailing call has written this shard's sums.
int rfsgpu_collective_gate(rfsgpu_filter *f, void *hip_stream) {
  CHECK_HANDLE(f);
  hipSetDevice(f->device);
  coll_gate_kernel<<<1, 64, 0, (hipStream_t)hip_stream>>>(f->dCollSeq, f->collStep, f->B.err);
  HIPCHK(hipGetLastError());
  return RFSGPU_OK;
}
// [multi] on the same stream, behind the collective: the total of the last step is in place.
int rfsgpu_collective_publish(rfsgpu_filter *f, void *hip_stream) {
  CHECK_HANDLE(f);
  hipSetDevice(f->device);
  coll_publish_kernel<<<1, 1, 0, (hipStream_t)hip_stream>>>(f->dCollSeq + 1, f->collStep);   // word [1]: "the total of step k is in place"
  HIPCHK(hipGetLastError());
  return RFSGPU_OK;
}
// [multi] Do the engine's stream and `hip_stream` make progress SIDE BY SIDE?  The sequence-number hand-over above needs it: a post
// kernel on the engine's stream waits, on the device, for a word that a kernel on the side stream publishes LATER in submission
// order.  Two streams that a runtime maps onto one hardware queue (GPU_MAX_HW_QUEUES streams share four by default) serialise instead,
// and the wait can only run out.  The probe plays the hand-over once with nothing at stake: a one-thread waiter on the engine's
// stream (bounded: 0.2 s), then the publish on the side stream; *side_by_side = 1 if the waiter saw the word.  Hosts probe once
// per (stream, side stream) pair and use the stream-event form (rfsgpu_step_async_deferred) when the answer is 0
// (ShardedRBPHDFilter, rfsgpu_group_update_deferred, bench.py).  RFSGPU_COLL_PROBE_DELAY_MS (test hook): the publish is held back by
// a spinning kernel for that long -- past the bound the probe must answer 0 and the hosts must take the event form.
int rfsgpu_collective_probe(rfsgpu_filter *f, void *hip_stream, int *side_by_side) {
  CHECK_HANDLE(f);
  if (!side_by_side) return fail(f, RFSGPU_ERR_INVALID, "collective_probe: null result pointer");
  hipSetDevice(f->device);
  const int k = ++f->collProbeSeq;
  hipStream_t side = (hipStream_t)hip_stream;
  coll_probe_kernel<<<1, 64, 0, f->stream>>>(f->dCollSeq + 2, k, f->dCollSeq + 3, COLL_PROBE_TICKS);
  HIPCHK(hipGetLastError());
  if (const char *d = getenv("RFSGPU_COLL_PROBE_DELAY_MS")) {
    const long long ms = atoll(d);
    if (ms > 0) coll_spin_kernel<<<1, 64, 0, side>>>(ms * 100000ll);      // (100 MHz constant clock)
  }
  coll_publish_kernel<<<1, 1, 0, side>>>(f->dCollSeq + 2, k);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(f->stream));
  HIPCHK(hipStreamSynchronize(side));
  int verdict = 0;
  HIPCHK(hipMemcpy(&verdict, f->dCollSeq + 3, sizeof(int), hipMemcpyDeviceToHost));
  *side_by_side = verdict == k ? 1 : 0;
  return RFSGPU_OK;
}

// Buffers of the fused step's cost-ordered launch (Victoria Park step; 2-D step when its workgroups are not all resident), created with the first fused step: durations (zero) and the identity order.
__global__ void iota_kernel(int *p, int n) { const int k = blockIdx.x * blockDim.x + threadIdx.x; if (k < n) p[k] = k; }
static int vp_order_buffers(rfsgpu_filter *f) {
  if (f->vpCost && f->vpOrder && f->vpOrderN == f->N) return RFSGPU_OK;
  static const bool off = [] { const char *e = getenv("RFSGPU_VP_COST_ORDER"); return e && atoi(e) == 0; }();
  if (off) f->vpOrderMode = 0;
  if (!f->vpCost) { HIPCHK(hipMalloc(&f->vpCost, (size_t)(f->Ncap + 4) * sizeof(float))); HIPCHK(hipMemsetAsync(f->vpCost, 0, (size_t)(f->Ncap + 4) * sizeof(float), f->stream)); }
  if (!f->vpOrder) HIPCHK(hipMalloc(&f->vpOrder, (size_t)f->Ncap * sizeof(int)));
  iota_kernel<<<(f->Ncap + 255) / 256, 256, 0, f->stream>>>(f->vpOrder, f->Ncap);      // (first use, or the particle count has changed: every slot its own particle)
  HIPCHK(hipGetLastError());
  f->vpOrderN = f->N;
  return RFSGPU_OK;
}
// [bench] / [test] The Victoria Park step's launch order.  mode 0: slot == particle; 1: order_in (slot -> particle, a permutation of 0..N-1),
// frozen; 2: re-sorted by every step's post kernel from that step's durations, longest first (the default).  cost_out (N floats, may be
// null): the last step's duration per particle, 100 MHz ticks.
int rfsgpu_step_launch_order(rfsgpu_filter *f, int mode, const int *order_in, float *cost_out) {
  CHECK_HANDLE(f);
  if (mode < 0 || mode > 2 || (mode == 1 && !order_in)) return fail(f, RFSGPU_ERR_INVALID, "step_launch_order: mode 0, 1 (with an order) or 2");
  hipSetDevice(f->device);
  int rc = vp_order_buffers(f);
  if (rc != RFSGPU_OK) return rc;
  HIPCHK(hipStreamSynchronize(f->stream));
  if (cost_out) HIPCHK(hipMemcpy(cost_out, f->vpCost, (size_t)f->N * sizeof(float), hipMemcpyDeviceToHost));
  if (mode == 1) {
    std::vector<char> seen((size_t)f->N, 0);
    for (int k = 0; k < f->N; k++) { const int p = order_in[k]; if (p < 0 || p >= f->N || seen[p]) return fail(f, RFSGPU_ERR_INVALID, "step_launch_order: not a permutation"); seen[p] = 1; }
    HIPCHK(hipMemcpy(f->vpOrder, order_in, (size_t)f->N * sizeof(int), hipMemcpyHostToDevice));
  }
  f->vpOrderMode = mode;
  return RFSGPU_OK;
}

// ---- one submission per predict + update cycle (round 5) --------------------------------------------------------------------
// RBPHDFilter::predict's map part (:415-442) + the host's new poses / weights + RBPHDFilter::update's body (:444-523) as ONE
// stream-ordered chain.  Where the configuration allows it the predict runs at the head of the fused step kernel
// (step_fused.h): 2-D model, immediate births (no candidate lists), no slot-ordered inheritance walk pending, fused steps on, a
// non-empty measurement set; the births then read the OLD poses from the buffer the previous update used while the update reads
// the new ones from the other buffer (the two swap roles).  Otherwise the same calls a host would make one by one are issued
// here, in the reference's order: predict kernels, input copies, step.  Either way the results are those of
// rfsgpu_predict_map + rfsgpu_set_poses + rfsgpu_set_weights + rfsgpu_step_async, bit for bit.
static void ensure_ids(rfsgpu_filter *f);
// RFSGPU_IO_PROFILE=1: host-side shares of rfsgpu_update_io, printed when the process ends (tuning aid: tools/boundary_trace.py)
struct IoProfile {
  bool on = getenv("RFSGPU_IO_PROFILE") != nullptr;
  long long n = 0, stage = 0, launch = 0, wait = 0, copyOut = 0;
  ~IoProfile() {
    if (on && n) fprintf(stderr, "rfsgpu_update_io x %lld: inputs into the pinned slot %.2f us, launches %.2f us, wait for the delivery %.2f us, weights out %.2f us\n", n,
                         stage / 1e3 / n, launch / 1e3 / n, wait / 1e3 / n, copyOut / 1e3 / n);
  }
};
static IoProfile g_ioProf;
static int cycle_impl(rfsgpu_filter *f, int predict, const double *x, const double *cov, int cov_stride, const double *w_in, const double *z, int n_z,
                      bool with_sums, int normalize, bool deliver = false) {
  if (predict < -1 || predict > 1) return fail(f, RFSGPU_ERR_INVALID, "cycle: predict is RFSGPU_CYCLE_NO_PREDICT (-1), 0 (static step only) or 1 (births + static step)");
  if (cov && cov_stride != 0 && cov_stride != 9) return fail(f, RFSGPU_ERR_INVALID, "cycle: cov_stride must be 0 or 9");
  if (n_z < 0 || n_z > RFSGPU_MAX_Z || (n_z > 0 && !z)) return fail(f, RFSGPU_ERR_INVALID, "cycle: bad measurement set");
  hipSetDevice(f->device);
  f->outArmed = false;
  // is this step the 2-D fused kernel (whose head can take the predict and pull the inputs)?
  const bool head = n_z > 0 && f->D == 2 && f->fuseSteps && !f->phaseTiming;
  bool fuse = head && predict >= 0 && f->cfg.birthGaussianMeasurementCountThreshold == 1u && !f->fastSlamHandle;
  if (fuse && predict == 1 && f->resampleOccured) {
    if (f->inheritMode == RFSGPU_INHERIT_REFERENCE) {      // a slot with a foreign parent: the level-ordered walk, stand-alone kernels
      ensure_ids(f);
      for (int i = 0; i < f->N && fuse; i++) fuse = f->ppid[i] == i || f->ppid[i] < 0 || f->ppid[i] >= f->N;
    } else if (f->inheritMode == RFSGPU_INHERIT_EXTERNAL && !f->externalAck) {
      fuse = false;                                        // (rfsgpu_predict_map_async below refuses with the full message)
    }
  }
  if (predict >= 0 && !fuse) { const int rc = rfsgpu_predict_map_async(f, predict); if (rc != RFSGPU_OK) return rc; }
  StepPredict sp = NO_HEAD;
  if (fuse) { sp.mode = predict ? 2 : 1; sp.nZprev = f->nZ; sp.birthPose = f->B.pose; }
  const long long tp0 = g_ioProf.on ? now_ns() : 0;
  const bool pull = head && f->ioPull;                    // the step kernel reads the pinned slot itself (step_fused.h, StepPredict)
  if (x || w_in) {
    double *h = nullptr;
    int k = 0;
    { const int rc = stage_slot(f, &h, &k); if (rc != RFSGPU_OK) return rc; }
    const bool pullCov = pull && x && cov && cov_stride == 9;
    if (pull) {
      // 13 doubles per particle side by side: the workgroup that owns a particle reads them in one go (step_fused.h, StepPredict)
      sp.inPacked = h;
      sp.inMask = (x ? 1 : 0) | (pullCov ? 2 : 0) | (w_in ? 4 : 0);
      for (int i = 0; i < f->N; i++) {
        double *d = h + (size_t)13 * i;
        if (x) { d[0] = x[3 * i]; d[1] = x[3 * i + 1]; d[2] = x[3 * i + 2]; }
        if (pullCov) memcpy(d + 3, cov + (size_t)9 * i, 9 * sizeof(double));
        if (w_in) d[12] = w_in[i];
      }
    }
    if (x) {
      if (!pull) {
        memcpy(h, x, (size_t)f->N * 3 * sizeof(double));
        double *dst = fuse ? f->poseAlt : f->B.pose;       // fused predict + copy commands: the births still need the old poses
        HIPCHK(hipMemcpyAsync(dst, h, (size_t)f->N * 3 * sizeof(double), hipMemcpyHostToDevice, f->stream));
        if (fuse) { f->poseAlt = f->B.pose; f->B.pose = dst; }
      }
      double *hc = h + (size_t)f->Ncap * 13;               // (behind the packed block: the small / copy-command forms of the covariance)
      if (cov) {                                           // (the births do not read the pose covariance)
        if (!pullCov) {
          const size_t n = cov_stride == 9 ? (size_t)f->N * 9 : 9;
          memcpy(hc, cov, n * sizeof(double));
          HIPCHK(hipMemcpyAsync(f->B.poseCov, hc, n * sizeof(double), hipMemcpyHostToDevice, f->stream));
        }
        f->P.poseCovStride = cov_stride;
        f->poseCovZero = false;
      } else {
        if (!f->poseCovZero) {
          memset(hc, 0, 9 * sizeof(double));
          HIPCHK(hipMemcpyAsync(f->B.poseCov, hc, 9 * sizeof(double), hipMemcpyHostToDevice, f->stream));
          f->poseCovZero = true;
        }
        f->P.poseCovStride = 0;
      }
    }
    if (w_in && !pull) {
      double *hw = h + (size_t)f->Ncap * 12;
      memcpy(hw, w_in, (size_t)f->N * sizeof(double));
      HIPCHK(hipMemcpyAsync(f->B.weight, hw, (size_t)f->N * sizeof(double), hipMemcpyHostToDevice, f->stream));
    }
    if (!pull) HIPCHK(hipEventRecord(f->evStage[k], f->stream));
    else f->stagePendingSlot = k;
  }
  if (n_z == 0) {       // no update (:450-452); the weights are still summed / normalised as the caller asked
    if (!with_sums) return RFSGPU_OK;
    const int rc = rfsgpu_weight_sums_async(f);
    if (rc != RFSGPU_OK) return rc;
    return normalize ? rfsgpu_normalize_weights(f, 0.0, f->dSums) : RFSGPU_OK;
  }
  StepOut so = NO_OUT;
  if (deliver && pull) {
    if (!f->hOutW) {
      HIPCHK(hipHostMalloc(&f->hOutW, (size_t)f->Ncap * sizeof(double) + 2 * sizeof(int)));
      f->hOutFlag = reinterpret_cast<int *>(f->hOutW + f->Ncap);
      f->hOutFlag[0] = 0; f->hOutFlag[1] = 0;
    }
    so.hostW = f->hOutW; so.hostFlag = f->hOutFlag; so.seq = ++f->outSeq;
    f->outArmed = true;
  }
  const long long tp1 = g_ioProf.on ? now_ns() : 0;
  const int rc = update_async_impl(f, z, n_z, with_sums, normalize, sp, so);
  if (g_ioProf.on) { const long long tp2 = now_ns(); g_ioProf.stage += tp1 - tp0; g_ioProf.launch += tp2 - tp1; }
  if (f->stagePendingSlot >= 0) {            // the pinned slot is read by the kernels just enqueued: its event goes behind them
    if (rc == RFSGPU_OK) HIPCHK(hipEventRecord(f->evStage[f->stagePendingSlot], f->stream));
    f->stagePendingSlot = -1;
  }
  return rc;
}
int rfsgpu_cycle_async(rfsgpu_filter *f, int predict, const double *x, const double *x_cov, int cov_stride, const double *w_in, const double *z, int n_z,
                       int normalize) {
  CHECK_HANDLE(f);
  return cycle_impl(f, predict, x, x_cov, cov_stride, w_in, z, n_z, true, normalize);
}
// The synchronous form for a host that stands where RBPHDFilter::update stands (:444-541): everything the update consumes goes in,
// the particle weights come back, ONE wait for the device and one error check -- in place of set_poses + set_weights + update +
// get_weights (four calls, two waits: 199 us per update at configs[1] against 125 us stream-ordered, bench.py `boundary`).
// (two halves, so that a group of shards can have every shard's chain in flight before it waits for the first: csrc/group.h)
static int update_io_begin(rfsgpu_filter *f, int predict, const double *x, const double *x_cov, int cov_stride, const double *w_in, const double *z, int n_z,
                           bool want_weights) {
  int rc;
  if (f->phaseTiming) {
    // rfsgpu_set_phase_timing (the binding's RFSGPU_PHASE_TIMING=1): the caller wants TimingInfo's per-phase buckets, i.e. the phases as
    // separate launches with their own event pairs -- the call-by-call sequence, synchronous rfsgpu_update included
    f->outArmed = false;
    if (predict < -1 || predict > 1) return fail(f, RFSGPU_ERR_INVALID, "update_io: predict is RFSGPU_CYCLE_NO_PREDICT (-1), 0 or 1");
    if (predict >= 0 && (rc = rfsgpu_predict_map(f, predict)) != RFSGPU_OK) return rc;
    if (x && (rc = rfsgpu_set_poses(f, x, x_cov, x_cov ? cov_stride : 0)) != RFSGPU_OK) return rc;
    if (w_in && (rc = rfsgpu_set_weights(f, w_in)) != RFSGPU_OK) return rc;
    rc = rfsgpu_update(f, z, n_z);
  } else {
    rc = cycle_impl(f, predict, x, x_cov, cov_stride, w_in, z, n_z, false, 0, true);
  }
  if (rc != RFSGPU_OK) return rc;
  if (want_weights && !f->outArmed) {      // (3-D model, phase timing, an empty measurement set, RFSGPU_IO_PULL=0: copy commands)
    if (!f->hWeights) HIPCHK(hipHostMalloc(&f->hWeights, (size_t)f->Ncap * sizeof(double)));
    HIPCHK(hipMemcpyAsync(f->hWeights, f->B.weight, (size_t)f->N * sizeof(double), hipMemcpyDeviceToHost, f->stream));
  }
  return RFSGPU_OK;
}
static int update_io_end(rfsgpu_filter *f, double *w_out) {
  hipSetDevice(f->device);
  if (f->outArmed) {
    // the post kernel's last workgroup writes weights, error word and -- behind a system-scope release -- the sequence number
    // into pinned memory: spin on it (bounded: after ~2 s the stream is asked instead, which also surfaces a launch failure)
    f->outArmed = false;
    const long long t0 = now_ns();
    bool seen = false;
    for (unsigned spin = 0;; spin++) {
      if (__atomic_load_n(&f->hOutFlag[1], __ATOMIC_ACQUIRE) == f->outSeq) { seen = true; break; }
      if ((spin & 1023u) == 1023u && now_ns() - t0 > 2000000000LL) break;
      __builtin_ia32_pause();
    }
    if (seen && f->hOutFlag[0] == 0) {
      const long long t1 = g_ioProf.on ? now_ns() : 0;
      if (w_out) memcpy(w_out, f->hOutW, (size_t)f->N * sizeof(double));
      if (g_ioProf.on) { g_ioProf.n++; g_ioProf.wait += t1 - t0; g_ioProf.copyOut += now_ns() - t1; }
      return RFSGPU_OK;
    }
    const int rc = check_device_errors(f);     // an error bit (or no answer): synchronise, read and clear the word, build the message
    harvest_async(f);
    // the stream is drained now: a step that merely outlasted the spin (large N, Murty-heavy, a profiler attached) HAS delivered
    if (!seen) seen = __atomic_load_n(&f->hOutFlag[1], __ATOMIC_ACQUIRE) == f->outSeq;
    if (w_out) memcpy(w_out, f->hOutW, (size_t)f->N * sizeof(double));
    if (rc == RFSGPU_OK && !seen) return fail(f, RFSGPU_ERR_HIP, "update_io: the post kernel never delivered its results");
    return rc;
  }
  const int rc = check_device_errors(f);      // the one wait (error word + weights land together)
  harvest_async(f);
  if (w_out && f->hWeights) memcpy(w_out, f->hWeights, (size_t)f->N * sizeof(double));
  return rc;
}
int rfsgpu_update_io(rfsgpu_filter *f, int predict, const double *x, const double *x_cov, int cov_stride, const double *w_in, const double *z, int n_z,
                     double *w_out) {
  CHECK_HANDLE(f);
  long long t0 = now_ns();
  int rc = update_io_begin(f, predict, x, x_cov, cov_stride, w_in, z, n_z, w_out != nullptr);
  if (rc != RFSGPU_OK) return rc;
  rc = update_io_end(f, w_out);
  f->timing.mapUpdate_cpu += now_ns() - t0;
  return rc;
}

int rfsgpu_kernel_time_stats(rfsgpu_filter *f, double *avg_ns3, int *n_steps) {
  CHECK_HANDLE(f);
  if (!avg_ns3 || !n_steps) return RFSGPU_ERR_INVALID;
  hipSetDevice(f->device);
  HIPCHK(hipStreamSynchronize(f->stream));
  harvest_async(f);
  for (int q = 0; q < 3; q++) avg_ns3[q] = f->statSteps ? f->statNs[q] / f->statSteps : 0.0;
  *n_steps = f->statSteps;
  f->lastPostNs = f->statSteps ? f->statPostNs / f->statSteps : 0.0;
  f->statSteps = 0;
  f->statNs[0] = f->statNs[1] = f->statNs[2] = 0.0;
  f->statPostNs = 0.0;
  return RFSGPU_OK;
}
// Average duration (ns) of the step's POST kernel (murty_jobs_kernel: the Murty-200 partitions when the queue is not empty, queue
// reset, weight sums / division) over the fused steps covered by the last rfsgpu_kernel_time_stats call.
double rfsgpu_post_kernel_avg_ns(const rfsgpu_filter *f) { return f ? f->lastPostNs : 0.0; }
int rfsgpu_set_step_timing_stride(rfsgpu_filter *f, int every) {
  CHECK_HANDLE(f);
  if (every < 1) return fail(f, RFSGPU_ERR_INVALID, "the timing stride is at least 1");
  f->timingStride = every;
  return RFSGPU_OK;
}

// The birth + static-step launches of one predict.  LV selects the particles whose birth step runs (birth.h).
static void launch_predict_kernels(rfsgpu_filter *f, int add_birth, const BirthLevel &LV) {
  if (f->D == 3)
  {
    predict_map_general_kernel<3, 4><<<(f->N + 3) / 4, 256, 0, f->stream>>>(f->B, f->P, f->cur, add_birth ? 1 : 0, f->nZ, LV);
    if (add_birth) predict_map_long_kernel<3, 2><<<(f->N + 1) / 2, 128, 0, f->stream>>>(f->B, f->P, f->cur, f->nZ, LV);   // (lists longer than a wavefront; the others exit at once)
  }
  else if (f->cfg.birthGaussianMeasurementCountThreshold != 1u)
  {
    predict_map_general_kernel<2, 4><<<(f->N + 3) / 4, 256, 0, f->stream>>>(f->B, f->P, f->cur, add_birth ? 1 : 0, f->nZ, LV);
    if (add_birth) predict_map_long_kernel<2, 2><<<(f->N + 1) / 2, 128, 0, f->stream>>>(f->B, f->P, f->cur, f->nZ, LV);
  }
  else  // CountThreshold == 1: every unused measurement is born at once, no candidate can exist -> lane-parallel kernel
    predict_map_kernel<4><<<(f->N + 3) / 4, 256, 0, f->stream>>>(f->B, f->P, f->cur, add_birth ? 1 : 0, f->nZ, LV);
}

// what a resampling copy / a migration row carries: pose + mixture only (RBPHDFilter: the per-slot birth state is copied
// lazily by the next predict, or by the host), or the birth bookkeeping as well (RFSGPU_INHERIT_EAGER, and FastSLAM handles)
static int map_only(const rfsgpu_filter *f) { return (f->inheritMode != RFSGPU_INHERIT_EAGER && !f->fastSlamHandle) ? 1 : 0; }
static void ensure_ids(rfsgpu_filter *f) {
  while ((int)f->pid.size() < f->Ncap) { f->pid.push_back((int)f->pid.size()); f->ppid.push_back((int)f->ppid.size()); }
}

// RBPHDFilter::predict's map part.  In the predicts that follow a resampling (resampleOccured_, RFSGPU_INHERIT_REFERENCE) the
// reference's slot-ordered lazy copy of unused_measurements_ / birthGaussians_ (RBPHDFilter.hpp:1005-1011) precedes each slot's
// birth step: see birth_inherit_kernel (birth.h) for how the walk is cut into levels.
static int predict_launch(rfsgpu_filter *f, int add_birth) {
  const BirthLevel all{nullptr, 0, 1};
  if (!(add_birth && f->resampleOccured && f->inheritMode == RFSGPU_INHERIT_REFERENCE && !f->fastSlamHandle)) {
    launch_predict_kernels(f, add_birth, all);
    return RFSGPU_OK;
  }
  ensure_ids(f);
  const int N = f->N;
  bool any = false;
  int maxLevel = 0;
  f->hInhParent.resize(N);
  f->hInhLevel.resize(N);
  for (int i = 0; i < N; i++) {
    int p = f->ppid[i];
    if (p < 0 || p >= N) p = i;     // (an id beyond the current particle count: a slot the reference would index out of range)
    f->hInhParent[i] = p;
    f->hInhLevel[i] = (p >= i) ? 0 : f->hInhLevel[p] + 1;
    any |= p != i;
    if (f->hInhLevel[i] > maxLevel) maxLevel = f->hInhLevel[i];
  }
  if (!any) { launch_predict_kernels(f, add_birth, all); return RFSGPU_OK; }
  if (!f->dInhLevel) HIPCHK(hipMalloc(&f->dInhLevel, (size_t)f->Ncap * sizeof(int)));
  {  // every buffer under its own guard: a failed allocation leaves the others usable and is retried by the next call
    const size_t nc = (size_t)f->Ncap * RFSGPU_MAX_CANDIDATES;
    if (!f->dInhParent) HIPCHK(hipMalloc(&f->dInhParent, (size_t)f->Ncap * sizeof(int)));
    if (!f->inhTmp.unused) HIPCHK(hipMalloc(&f->inhTmp.unused, (size_t)f->Ncap * sizeof(unsigned long long)));
    if (!f->inhTmp.count) HIPCHK(hipMalloc(&f->inhTmp.count, (size_t)f->Ncap * sizeof(int)));
    if (!f->inhTmp.sup) HIPCHK(hipMalloc(&f->inhTmp.sup, nc * sizeof(int)));
    if (!f->inhTmp.chk) HIPCHK(hipMalloc(&f->inhTmp.chk, nc * sizeof(int)));
    if (!f->inhTmp.mean) HIPCHK(hipMalloc(&f->inhTmp.mean, nc * 3 * sizeof(double)));
    if (!f->inhTmp.cov) HIPCHK(hipMalloc(&f->inhTmp.cov, nc * 6 * sizeof(double)));
  }
  // (pageable host vectors: hipMemcpyAsync from them returns after the copy has been staged, so they may be reused at once)
  HIPCHK(hipMemcpyAsync(f->dInhParent, f->hInhParent.data(), (size_t)N * sizeof(int), hipMemcpyHostToDevice, f->stream));
  HIPCHK(hipMemcpyAsync(f->dInhLevel, f->hInhLevel.data(), (size_t)N * sizeof(int), hipMemcpyHostToDevice, f->stream));
  birth_inherit_kernel<0><<<N, 128, 0, f->stream>>>(f->B, f->inhTmp, f->dInhParent, f->dInhLevel, 0);
  birth_inherit_kernel<1><<<N, 128, 0, f->stream>>>(f->B, f->inhTmp, f->dInhParent, f->dInhLevel, 0);
  launch_predict_kernels(f, add_birth, BirthLevel{f->dInhLevel, 0, 1});
  for (int L = 1; L <= maxLevel; L++) {
    birth_inherit_kernel<2><<<N, 128, 0, f->stream>>>(f->B, f->inhTmp, f->dInhParent, f->dInhLevel, L);
    launch_predict_kernels(f, add_birth, BirthLevel{f->dInhLevel, L, 0});
  }
  f->candUsed = f->candUsed || f->D == 3 || f->cfg.birthGaussianMeasurementCountThreshold != 1u;
  return RFSGPU_OK;
}

// One level of the level-ordered birth step, for hosts that move the per-slot birth lists between shards themselves (rfsgpu.h).
int rfsgpu_predict_map_level(rfsgpu_filter *f, int add_birth, const int *level_of_slot, int level, int do_static) {
  CHECK_HANDLE(f);
  if (!level_of_slot) return fail(f, RFSGPU_ERR_INVALID, "predict_map_level: null level array");
  f->externalAck = true;
  hipSetDevice(f->device);
  if (!f->dInhLevel) HIPCHK(hipMalloc(&f->dInhLevel, (size_t)f->Ncap * sizeof(int)));
  f->hInhLevel.assign(level_of_slot, level_of_slot + f->N);
  HIPCHK(hipMemcpyAsync(f->dInhLevel, f->hInhLevel.data(), (size_t)f->N * sizeof(int), hipMemcpyHostToDevice, f->stream));
  HIPCHK(hipEventRecord(f->ev[EV_P0], f->stream));
  launch_predict_kernels(f, add_birth, BirthLevel{f->dInhLevel, level, do_static ? 1 : 0});
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(f->ev[EV_P1], f->stream));
  const int rc = check_device_errors(f);
  accumulate(f->ev[EV_P0], f->ev[EV_P1], f->timing.predict_wall, nullptr);
  f->candUsed = f->candUsed || f->D == 3 || f->cfg.birthGaussianMeasurementCountThreshold != 1u;
  return rc;
}

int rfsgpu_predict_map(rfsgpu_filter *f, int add_birth) {
  CHECK_HANDLE(f);
  if (add_birth && f->resampleOccured && f->inheritMode == RFSGPU_INHERIT_EXTERNAL && !f->externalAck)
    return fail(f, RFSGPU_ERR_INVALID, "predict_map with births after a resampling in RFSGPU_INHERIT_EXTERNAL mode, but the host has not applied the "
                "inheritance rule (rfsgpu_set_unused_masks, rfsgpu_predict_map_level, or rfsgpu_set_birth_inheritance(EXTERNAL) again to acknowledge): "
                "go through the multi-GPU host's own predict (ShardedRBPHDFilter.predict_map / rfsgpu_group_predict_map)");
  long long t0 = now_ns();
  hipSetDevice(f->device);
  HIPCHK(hipEventRecord(f->ev[EV_P0], f->stream));
  { const int rcl = predict_launch(f, add_birth); if (rcl != RFSGPU_OK) return rcl; }
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(f->ev[EV_P1], f->stream));
  int rc = check_device_errors(f);
  accumulate(f->ev[EV_P0], f->ev[EV_P1], f->timing.predict_wall, nullptr);
  f->timing.predict_cpu += now_ns() - t0;
  return rc;
}

// RBPHDFilter::predict's map part without the error-word readback: stream-ordered, errors surface at the next synchronising call.
int rfsgpu_predict_map_async(rfsgpu_filter *f, int add_birth) {
  CHECK_HANDLE(f);
  if (add_birth && f->resampleOccured && f->inheritMode == RFSGPU_INHERIT_EXTERNAL && !f->externalAck)
    return fail(f, RFSGPU_ERR_INVALID, "predict_map with births after a resampling in RFSGPU_INHERIT_EXTERNAL mode, but the host has not applied the "
                "inheritance rule (rfsgpu_set_unused_masks, rfsgpu_predict_map_level, or rfsgpu_set_birth_inheritance(EXTERNAL) again to acknowledge): "
                "go through the multi-GPU host's own predict (ShardedRBPHDFilter.predict_map / rfsgpu_group_predict_map)");
  long long t0 = now_ns();
  hipSetDevice(f->device);
  if (f->predPending && hipEventQuery(f->ev[EV_P1]) == hipSuccess) {   // the previous async predict has long finished: book it
    accumulate(f->ev[EV_P0], f->ev[EV_P1], f->timing.predict_wall, nullptr);
    f->predPending = false;
  }
  const bool rec = !f->predPending;
  if (rec) HIPCHK(hipEventRecord(f->ev[EV_P0], f->stream));
  { const int rcl = predict_launch(f, add_birth); if (rcl != RFSGPU_OK) return rcl; }
  HIPCHK(hipGetLastError());
  if (rec) { HIPCHK(hipEventRecord(f->ev[EV_P1], f->stream)); f->predPending = true; }
  f->timing.predict_cpu += now_ns() - t0;
  return RFSGPU_OK;
}

// A run of `n` predicts that add no births, in one launch (birth.h, static_steps_kernel): the same bits as n calls of
// rfsgpu_set_lmk_process_noise(Q_k) + rfsgpu_predict_map_async(f, 0).  noises: [n][D*D] row-major, or NULL = the noise that is set
// now, n times.  Stream-ordered.
int rfsgpu_static_steps_async(rfsgpu_filter *f, int n, const double *noises) {
  CHECK_HANDLE(f);
  if (n < 0) return fail(f, RFSGPU_ERR_INVALID, "static_steps: negative count");
  hipSetDevice(f->device);
  const int D = f->D;
  for (int k0 = 0; k0 < n; k0 += STATIC_RUN_MAX) {
    StaticRun R;
    R.n = std::min(STATIC_RUN_MAX, n - k0);
    for (int r = 0; r < R.n; r++) {
      const double *Q = noises ? noises + (size_t)(k0 + r) * D * D : f->Qlm;
      if (D == 2) { R.q[r][0] = Q[0]; R.q[r][1] = Q[1]; R.q[r][2] = Q[3]; R.q[r][3] = R.q[r][4] = R.q[r][5] = 0.0; }
      else { R.q[r][0] = Q[0]; R.q[r][1] = Q[1]; R.q[r][2] = Q[2]; R.q[r][3] = Q[4]; R.q[r][4] = Q[5]; R.q[r][5] = Q[8]; }
    }
    if (D == 3) static_steps_kernel<3, 4><<<(f->N + 3) / 4, 256, 0, f->stream>>>(f->B, f->cur, R);
    else static_steps_kernel<2, 4><<<(f->N + 3) / 4, 256, 0, f->stream>>>(f->B, f->cur, R);
    HIPCHK(hipGetLastError());
  }
  return RFSGPU_OK;
}

// ParticleFilter::propagate for the Ackerman model on the device (motion.h): stream-ordered, no host wait.  u = {speed,
// steering}, var = their noise variances (null or zeros: noise-free), geom = {h, l, dx, dy}; (seed, call) key the generator.
int rfsgpu_propagate_ackerman_async(rfsgpu_filter *f, const double *u, const double *var, double dt, const double *geom, unsigned long long seed,
                                    unsigned long long call) {
  CHECK_HANDLE(f);
  if (!u || !geom || !(dt == dt)) return fail(f, RFSGPU_ERR_INVALID, "propagate_ackerman: bad arguments");
  if (var && (var[0] < 0 || var[1] < 0)) return fail(f, RFSGPU_ERR_INVALID, "propagate_ackerman: negative variance");
  hipSetDevice(f->device);
  AckermanStep A;
  A.uv = u[0]; A.ur = u[1];
  A.sv = var ? std::sqrt(var[0]) : 0.0; A.sr = var ? std::sqrt(var[1]) : 0.0;
  A.dt = dt;
  A.h = geom[0]; A.l = geom[1]; A.dx = geom[2]; A.dy = geom[3];
  A.seed = seed; A.call = call;
  propagate_ackerman_kernel<<<(f->N + 255) / 256, 256, 0, f->stream>>>(f->B.pose, f->N, A);
  HIPCHK(hipGetLastError());
  return RFSGPU_OK;
}

// A run of n propagations (consecutive odometry messages) in one launch: u [n][2], var [n][2] or NULL, dt [n]; the calls are
// numbered call0, call0 + 1, ...  Same poses, bit for bit, as n calls of rfsgpu_propagate_ackerman_async.
int rfsgpu_propagate_ackerman_run_async(rfsgpu_filter *f, int n, const double *u, const double *var, const double *dt, const double *geom,
                                        unsigned long long seed, unsigned long long call0) {
  CHECK_HANDLE(f);
  if (n < 0 || (n > 0 && (!u || !dt || !geom))) return fail(f, RFSGPU_ERR_INVALID, "propagate_ackerman_run: bad arguments");
  for (int k = 0; k < n; k++)
    if (!(dt[k] == dt[k]) || (var && (var[2 * k] < 0 || var[2 * k + 1] < 0))) return fail(f, RFSGPU_ERR_INVALID, "propagate_ackerman_run: bad interval / negative variance");
  hipSetDevice(f->device);
  for (int k0 = 0; k0 < n; k0 += ACKERMAN_RUN_MAX) {
    AckermanRun R;
    R.n = std::min(ACKERMAN_RUN_MAX, n - k0);
    for (int r = 0; r < R.n; r++) {
      const int k = k0 + r;
      R.uv[r] = u[2 * k]; R.ur[r] = u[2 * k + 1];
      R.sv[r] = var ? std::sqrt(var[2 * k]) : 0.0; R.sr[r] = var ? std::sqrt(var[2 * k + 1]) : 0.0;
      R.dt[r] = dt[k];
    }
    R.h = geom[0]; R.l = geom[1]; R.dx = geom[2]; R.dy = geom[3];
    R.seed = seed; R.call0 = call0 + (unsigned long long)k0;
    propagate_ackerman_run_kernel<<<(f->N + 255) / 256, 256, 0, f->stream>>>(f->B.pose, f->N, R);
    HIPCHK(hipGetLastError());
  }
  return RFSGPU_OK;
}

// Everything a step consumes from the host in ONE stream-ordered call: poses (+ covariance) and, for the Victoria Park model,
// the laser scan (MeasurementModel_VictoriaPark::setLaserScan, src/MeasurementModel_VictoriaPark.cpp:267-281).  The caller's
// buffers are copied into a pinned staging ring before the call returns; the host never waits for the device (except when all
// four ring slots are still in flight).  x may be null (poses unchanged), scan may be null (scan unchanged / 2-D model).
int rfsgpu_set_step_inputs_async(rfsgpu_filter *f, const double *x, const double *cov, int cov_stride, const double *scan, int n_scan) {
  CHECK_HANDLE(f);
  if (cov && cov_stride != 0 && cov_stride != 9) return fail(f, RFSGPU_ERR_INVALID, "set_step_inputs: cov_stride must be 0 or 9");
  if (scan && (f->model != RFSGPU_MODEL_VICTORIAPARK_3D || n_scan < 2 || n_scan > RFSGPU_VP_MAX_SCAN)) return fail(f, RFSGPU_ERR_INVALID, "set_step_inputs: laser scan needs the Victoria Park model and 2..RFSGPU_VP_MAX_SCAN beams");
  if (!x && !scan) return RFSGPU_OK;
  hipSetDevice(f->device);
  double *h = nullptr;
  int k = 0;
  { const int rc = stage_slot(f, &h, &k); if (rc != RFSGPU_OK) return rc; }
  if (x) {
    memcpy(h, x, (size_t)f->N * 3 * sizeof(double));
    HIPCHK(hipMemcpyAsync(f->B.pose, h, (size_t)f->N * 3 * sizeof(double), hipMemcpyHostToDevice, f->stream));
    double *hc = h + (size_t)f->Ncap * 3;
    if (cov) {
      const size_t n = cov_stride == 9 ? (size_t)f->N * 9 : 9;
      memcpy(hc, cov, n * sizeof(double));
      HIPCHK(hipMemcpyAsync(f->B.poseCov, hc, n * sizeof(double), hipMemcpyHostToDevice, f->stream));
      f->P.poseCovStride = cov_stride;
      f->poseCovZero = false;
    } else {
      if (!f->poseCovZero) {   // (the shared all-zero covariance is written once, not per call)
        memset(hc, 0, 9 * sizeof(double));
        HIPCHK(hipMemcpyAsync(f->B.poseCov, hc, 9 * sizeof(double), hipMemcpyHostToDevice, f->stream));
        f->poseCovZero = true;
      }
      f->P.poseCovStride = 0;
    }
  }
  if (scan) {
    double area = 0;
    for (int q = 1; q < n_scan; q++) area += scan[q] * scan[q - 1];
    area += scan[0] * scan[n_scan - 1];
    area *= sin(acos(-1) / 360) / 2;
    f->vpClutter = f->vp.expectedClutterNumber / area;
    double *hs = h + (size_t)f->Ncap * 22;
    memcpy(hs, scan, (size_t)n_scan * sizeof(double));
    HIPCHK(hipMemcpyAsync(f->B.scan, hs, (size_t)n_scan * sizeof(double), hipMemcpyHostToDevice, f->stream));
    f->B.nScan = n_scan;
    rebuild_params(f);
  }
  HIPCHK(hipEventRecord(f->evStage[k], f->stream));
  return RFSGPU_OK;
}

int rfsgpu_get_unused(rfsgpu_filter *f, int slot, int *idx, int max_n, int *n_out) {
  CHECK_HANDLE(f);
  if (slot < 0 || slot >= f->N) return RFSGPU_ERR_INVALID;
  hipSetDevice(f->device);
  unsigned long long m = 0;
  HIPCHK(hipMemcpyAsync(&m, f->B.unusedMask + slot, sizeof(m), hipMemcpyDeviceToHost, f->stream));
  HIPCHK(hipStreamSynchronize(f->stream));
  int k = 0;
  for (int z = 0; z < 64; z++)
    if ((m >> z) & 1ull) { if (k < max_n && idx) idx[k] = z; k++; }
  if (n_out) *n_out = k;
  return RFSGPU_OK;
}
int rfsgpu_landmarks_in_fov(rfsgpu_filter *f, int slot, int *n_out) {
  CHECK_HANDLE(f);
  if (slot < 0 || slot >= f->N || !n_out) return RFSGPU_ERR_INVALID;
  hipSetDevice(f->device);
  HIPCHK(hipMemcpyAsync(n_out, f->B.nInFov + slot, sizeof(int), hipMemcpyDeviceToHost, f->stream));
  HIPCHK(hipStreamSynchronize(f->stream));
  return RFSGPU_OK;
}

// ---- weights / resampling ---------------------------------------------------------------------------

int rfsgpu_weight_sums_async(rfsgpu_filter *f) {
  CHECK_HANDLE(f);
  hipSetDevice(f->device);
  weight_sums_kernel<<<1, 1024, 0, f->stream>>>(f->B.weight, f->N, f->dSums);
  HIPCHK(hipGetLastError());
  return RFSGPU_OK;
}
void *rfsgpu_weight_sums_device_ptr(rfsgpu_filter *f) { return f ? (void *)f->dSums : nullptr; }
int rfsgpu_weight_sums(rfsgpu_filter *f, double *out) {
  CHECK_HANDLE(f);
  int rc = rfsgpu_weight_sums_async(f);
  if (rc != RFSGPU_OK) return rc;
  HIPCHK(hipMemcpyAsync(f->hSums, f->dSums, 2 * sizeof(double), hipMemcpyDeviceToHost, f->stream));
  HIPCHK(hipStreamSynchronize(f->stream));
  out[0] = f->hSums[0];
  out[1] = f->hSums[1];
  return RFSGPU_OK;
}
int rfsgpu_normalize_weights(rfsgpu_filter *f, double sum, const void *sum_dev) { return rfsgpu_normalize_weights_parts(f, sum, sum_dev, 1); }
int rfsgpu_normalize_weights_parts(rfsgpu_filter *f, double sum, const void *sum_dev, int n_parts) {
  CHECK_HANDLE(f);
  if (n_parts < 1) return RFSGPU_ERR_INVALID;
  hipSetDevice(f->device);
  long long t0 = now_ns();
  HIPCHK(hipEventRecord(f->ev[EV_R0], f->stream));
  normalize_kernel<<<(f->N + 255) / 256, 256, 0, f->stream>>>(f->B.weight, f->N, sum, (const double *)sum_dev, n_parts);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(f->ev[EV_R1], f->stream));
  // stream-ordered, no host sync: the next call that syncs (or rfsgpu_synchronize) completes it; its event pair is
  // folded into the Resampling bucket by rfsgpu_get_timing
  f->normPending = true;
  f->timing.particleResample_cpu += now_ns() - t0;
  return RFSGPU_OK;
}
int rfsgpu_n_particles(const rfsgpu_filter *f) { return f ? f->N : -1; }
int rfsgpu_max_particles(const rfsgpu_filter *f) { return f ? f->Ncap : -1; }
int rfsgpu_resample_apply(rfsgpu_filter *f, const int *src_slot) { return f ? rfsgpu_resample_apply_n(f, src_slot, f->N) : RFSGPU_ERR_INVALID; }
int rfsgpu_resample_apply_n(rfsgpu_filter *f, const int *src_slot, int n_out) {
  CHECK_HANDLE(f);
  if (!src_slot || n_out < 1 || n_out > f->N) return RFSGPU_ERR_INVALID;
  const int nIn = f->N;
  for (int k = 0; k < n_out; k++) {
    const int s = src_slot[k];
    // a source is either a surviving slot that keeps itself or a slot beyond the new count (dropped after the copy)
    if (s < 0 || s >= nIn || (s < n_out && src_slot[s] != s)) return fail(f, RFSGPU_ERR_INVALID, "resample_apply: a source slot below the new count must keep itself");
  }
  f->N = n_out;
  f->B.N = n_out;
  // ids as ParticleFilter::resample leaves them (:446-479): a copy has its source's id (Particle::copy) and idParent_ = that id;
  // a slot that keeps its particle gets idParent_ = its own id (case 1).  Sources are never destinations, so in place is safe.
  ensure_ids(f);
  for (int k = 0; k < n_out; k++) {
    const int s = src_slot[k];
    if (s != k) { f->pid[k] = f->pid[s]; f->ppid[k] = f->pid[s]; }
    else f->ppid[k] = f->pid[k];
  }
  f->resampleOccured = true;
  f->externalAck = false;
  hipSetDevice(f->device);
  long long t0 = now_ns();
  // Round 6: stream-ordered.  The plan goes into a slot of the pinned staging ring (the caller's buffer is free when the call returns) and
  // the gather kernel reads it from there -- one 4-byte PCIe read per workgroup -- instead of a pageable host-to-device copy (staged and
  // waited for by the runtime), two timing events and a stream synchronisation: 125 -> 84 us per call at 2000 particles under the
  // unmodified 2-D driver.  Nothing after it needs the host to wait: every reader of the maps synchronises on the stream itself, the next
  // predict / update is ordered behind the gather.  TimingInfo's particleResample_wall books the host time of the call (the device part
  // overlaps whatever the host does next).  RFSGPU_RESAMPLE_SYNC=1 keeps the synchronous form (A/B).
  static const bool syncForm = [] { const char *e = getenv("RFSGPU_RESAMPLE_SYNC"); return e && atoi(e) != 0; }();
  if (syncForm) {
    HIPCHK(hipMemcpyAsync(f->dSrcSlot, src_slot, (size_t)f->N * sizeof(int), hipMemcpyHostToDevice, f->stream));
    HIPCHK(hipEventRecord(f->ev[EV_R0], f->stream));
    resample_gather_kernel<<<f->N, 256, 0, f->stream>>>(f->B, f->cur, f->dSrcSlot, f->P.poseCovStride, map_only(f));
    set_weights_kernel<<<(f->N + 255) / 256, 256, 0, f->stream>>>(f->B.weight, f->N, 1.0);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(f->ev[EV_R1], f->stream));
    HIPCHK(hipStreamSynchronize(f->stream));
    accumulate(f->ev[EV_R0], f->ev[EV_R1], f->timing.particleResample_wall, nullptr);
    f->timing.particleResample_cpu += now_ns() - t0;
    return RFSGPU_OK;
  }
  double *hs;
  int ks;
  int rc = stage_slot(f, &hs, &ks);
  if (rc != RFSGPU_OK) return rc;
  memcpy(hs, src_slot, (size_t)f->N * sizeof(int));
  resample_gather_kernel<<<f->N, 256, 0, f->stream>>>(f->B, f->cur, reinterpret_cast<const int *>(hs), f->P.poseCovStride, map_only(f));
  set_weights_kernel<<<(f->N + 255) / 256, 256, 0, f->stream>>>(f->B.weight, f->N, 1.0);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(f->evStage[ks], f->stream));      // the slot is free again once the gather has read it
  const long long dtn = now_ns() - t0;
  f->timing.particleResample_wall += dtn;
  f->timing.particleResample_cpu += dtn;
  return RFSGPU_OK;
}

int rfsgpu_set_birth_inheritance(rfsgpu_filter *f, int mode) {
  CHECK_HANDLE(f);
  if (mode != RFSGPU_INHERIT_REFERENCE && mode != RFSGPU_INHERIT_EAGER && mode != RFSGPU_INHERIT_EXTERNAL) return fail(f, RFSGPU_ERR_INVALID, "set_birth_inheritance: unknown mode");
  f->inheritMode = mode;
  f->externalAck = true;    // (EXTERNAL: the caller asserts that it owns the rule from here on, incl. for the coming predict)
  return RFSGPU_OK;
}
int rfsgpu_get_birth_inheritance(const rfsgpu_filter *f) { return f ? f->inheritMode : -1; }
int rfsgpu_get_particle_ids(rfsgpu_filter *f, int *id, int *parent_id) {
  CHECK_HANDLE(f);
  ensure_ids(f);
  for (int k = 0; k < f->N; k++) { if (id) id[k] = f->pid[k]; if (parent_id) parent_id[k] = f->ppid[k]; }
  return RFSGPU_OK;
}
int rfsgpu_set_particle_ids(rfsgpu_filter *f, const int *id, const int *parent_id) {
  CHECK_HANDLE(f);
  ensure_ids(f);
  for (int k = 0; k < f->N; k++) { if (id) f->pid[k] = id[k]; if (parent_id) f->ppid[k] = parent_id[k]; }
  return RFSGPU_OK;
}
int rfsgpu_resample_occured(const rfsgpu_filter *f) { return f ? (f->resampleOccured ? 1 : 0) : -1; }
int rfsgpu_get_unused_masks(rfsgpu_filter *f, unsigned long long *masks) {
  CHECK_HANDLE(f);
  if (!masks) return RFSGPU_ERR_INVALID;
  hipSetDevice(f->device);
  HIPCHK(hipMemcpyAsync(masks, f->B.unusedMask, (size_t)f->N * sizeof(unsigned long long), hipMemcpyDeviceToHost, f->stream));
  HIPCHK(hipStreamSynchronize(f->stream));
  return RFSGPU_OK;   // (a read acknowledges nothing: only set_unused_masks / predict_map_level / set_birth_inheritance do)
}
int rfsgpu_has_birth_candidates(const rfsgpu_filter *f) { return f ? (f->candUsed ? 1 : 0) : -1; }
int rfsgpu_set_unused_masks(rfsgpu_filter *f, const unsigned long long *masks) {
  CHECK_HANDLE(f);
  if (!masks) return RFSGPU_ERR_INVALID;
  hipSetDevice(f->device);
  HIPCHK(hipMemcpyAsync(f->B.unusedMask, masks, (size_t)f->N * sizeof(unsigned long long), hipMemcpyHostToDevice, f->stream));
  HIPCHK(hipStreamSynchronize(f->stream));
  f->externalAck = true;
  return RFSGPU_OK;
}

// ---- cross-shard migration (multi-GPU resampling): packed rows in DEVICE buffers, stream-ordered, no host sync -----------
// candidates a migration row carries: the whole list for a filter that keeps one, nothing for a configuration that never does
static int row_cand(const rfsgpu_filter *f) {
  if (map_only(f)) return 0;   // a migrant carries what Particle::copy carries: pose + mixture
  return (f->model == RFSGPU_MODEL_VICTORIAPARK_3D || f->cfg.birthGaussianMeasurementCountThreshold != 1u || f->candUsed) ? RFSGPU_MAX_CANDIDATES : 0;
}
size_t rfsgpu_slab_row_bytes(const rfsgpu_filter *f) { return f ? slab_row_bytes(f->B.npl, f->cap, row_cand(f)) : 0; }
static int slab_rows(rfsgpu_filter *f, const int *slots, int n, void *dev_rows, bool exporting) {
  if (n == 0) return RFSGPU_OK;
  if (!slots || !dev_rows || n < 0) return fail(f, RFSGPU_ERR_INVALID, "slab rows: bad arguments");
  for (int k = 0; k < n; k++)
    if (slots[k] < 0 || slots[k] >= f->N) return fail(f, RFSGPU_ERR_INVALID, "slab rows: slot out of range");
  hipSetDevice(f->device);
  if (n > f->rowSlotsCap) {   // (a slot may be exported many times -- one heavy parent, many children elsewhere -- so n is not bounded by N)
    HIPCHK(hipStreamSynchronize(f->stream));
    if (f->dRowSlots) HIPCHK(hipFree(f->dRowSlots));
    f->dRowSlots = nullptr;
    f->rowSlotsCap = 0;
    const int want = std::max(n, f->Ncap);
    HIPCHK(hipMalloc(&f->dRowSlots, (size_t)want * sizeof(int)));
    f->rowSlotsCap = want;
  }
  HIPCHK(hipMemcpyAsync(f->dRowSlots, slots, (size_t)n * sizeof(int), hipMemcpyHostToDevice, f->stream));
  if (exporting) slab_rows_kernel<true><<<n, 256, 0, f->stream>>>(f->B, f->cur, f->dRowSlots, (unsigned char *)dev_rows, f->P.poseCovStride, row_cand(f), map_only(f));
  else slab_rows_kernel<false><<<n, 256, 0, f->stream>>>(f->B, f->cur, f->dRowSlots, (unsigned char *)dev_rows, f->P.poseCovStride, row_cand(f), map_only(f));
  HIPCHK(hipGetLastError());
  return RFSGPU_OK;
}
int rfsgpu_export_slab_rows(rfsgpu_filter *f, const int *slots, int n, void *dev_rows) {
  CHECK_HANDLE(f);
  return slab_rows(f, slots, n, dev_rows, true);
}
int rfsgpu_import_slab_rows(rfsgpu_filter *f, const int *slots, int n, const void *dev_rows) {
  CHECK_HANDLE(f);
  return slab_rows(f, slots, n, const_cast<void *>(dev_rows), false);
}
void *rfsgpu_weights_device_ptr(rfsgpu_filter *f) { return f ? (void *)f->B.weight : nullptr; }

// ---- timing / misc -----------------------------------------------------------------------------------

int rfsgpu_get_timing(rfsgpu_filter *f, rfsgpu_timing *t) {
  CHECK_HANDLE(f);
  if (!t) return RFSGPU_ERR_INVALID;
  if (f->normPending || f->predPending || f->ringCount > 0) {   // (steps whose results were delivered through pinned memory have not been harvested yet)
    hipSetDevice(f->device);
    HIPCHK(hipStreamSynchronize(f->stream));
    harvest_async(f);
    if (f->normPending) accumulate(f->ev[EV_R0], f->ev[EV_R1], f->timing.particleResample_wall, nullptr);
    if (f->predPending) accumulate(f->ev[EV_P0], f->ev[EV_P1], f->timing.predict_wall, nullptr);
    f->normPending = f->predPending = false;
  }
  *t = f->timing;
  return RFSGPU_OK;
}
int rfsgpu_reset_timing(rfsgpu_filter *f) {
  CHECK_HANDLE(f);
  memset(&f->timing, 0, sizeof(f->timing));
  return RFSGPU_OK;
}
int rfsgpu_synchronize(rfsgpu_filter *f) {
  CHECK_HANDLE(f);
  hipSetDevice(f->device);
  const int rc = check_device_errors(f);  // syncs the stream, reports errors of async steps
  harvest_async(f);
  return rc;
}
int rfsgpu_set_phase_timing(rfsgpu_filter *f, int on) {
  CHECK_HANDLE(f);
  f->phaseTiming = on != 0;
  return RFSGPU_OK;
}
void *rfsgpu_stream(rfsgpu_filter *f) { return f ? (void *)f->stream : nullptr; }
int rfsgpu_set_stream(rfsgpu_filter *f, void *hip_stream) {
  CHECK_HANDLE(f);
  hipSetDevice(f->device);
  HIPCHK(hipStreamSynchronize(f->stream));
  f->stream = hip_stream ? (hipStream_t)hip_stream : f->ownStream;
  return RFSGPU_OK;
}
int rfsgpu_bind_weight_sums_buffer(rfsgpu_filter *f, void *dev_ptr) {
  CHECK_HANDLE(f);
  f->dSums = dev_ptr ? (double *)dev_ptr : f->ownSums;
  return RFSGPU_OK;
}

int rfsgpu_save_state(rfsgpu_filter *f) {
  CHECK_HANDLE(f);
  hipSetDevice(f->device);
  const size_t slabBytes = (size_t)f->Ncap * f->B.npl * f->cap * sizeof(double);
  if (!f->snapSlab) {
    bool ok = true;
    ok &= hipMalloc(&f->snapSlab, slabBytes) == hipSuccess;
    ok &= hipMalloc(&f->snapWeight, f->Ncap * sizeof(double)) == hipSuccess;
    ok &= hipMalloc(&f->snapCount, f->Ncap * sizeof(int)) == hipSuccess;
    ok &= hipMalloc(&f->snapFov, f->Ncap * sizeof(int)) == hipSuccess;
    ok &= hipMalloc(&f->snapUnused, f->Ncap * sizeof(unsigned long long)) == hipSuccess;
    if (!ok) {
      hipFree(f->snapSlab); hipFree(f->snapWeight); hipFree(f->snapCount); hipFree(f->snapFov); hipFree(f->snapUnused);
      f->snapSlab = nullptr; f->snapWeight = nullptr; f->snapCount = nullptr; f->snapFov = nullptr; f->snapUnused = nullptr;
      return fail(f, RFSGPU_ERR_HIP, "save_state: out of device memory");
    }
  }
  HIPCHK(hipMemcpyAsync(f->snapSlab, f->B.slab[f->cur], slabBytes, hipMemcpyDeviceToDevice, f->stream));
  HIPCHK(hipMemcpyAsync(f->snapWeight, f->B.weight, f->N * sizeof(double), hipMemcpyDeviceToDevice, f->stream));
  HIPCHK(hipMemcpyAsync(f->snapCount, f->B.count, f->N * sizeof(int), hipMemcpyDeviceToDevice, f->stream));
  HIPCHK(hipMemcpyAsync(f->snapFov, f->B.nInFov, f->N * sizeof(int), hipMemcpyDeviceToDevice, f->stream));
  HIPCHK(hipMemcpyAsync(f->snapUnused, f->B.unusedMask, f->N * sizeof(unsigned long long), hipMemcpyDeviceToDevice, f->stream));
  HIPCHK(hipStreamSynchronize(f->stream));
  f->snapNZ = f->nZ;
  f->snapN = f->N;
  return RFSGPU_OK;
}
int rfsgpu_restore_state(rfsgpu_filter *f) {
  CHECK_HANDLE(f);
  if (!f->snapSlab) return fail(f, RFSGPU_ERR_INVALID, "restore_state: nothing saved");
  hipSetDevice(f->device);
  f->N = f->snapN;  // (the particle count is part of the state: the multi-hypothesis FastSLAM update changes it)
  f->B.N = f->N;
  // only the live entries are copied (one block per particle); asynchronous on the handle's stream
  if (f->B.npl == 7) restore_state_kernel<7><<<f->N, 256, 0, f->stream>>>(f->B, f->cur, f->snapSlab, f->snapWeight, f->snapCount, f->snapFov, f->snapUnused);
  else restore_state_kernel<11><<<f->N, 256, 0, f->stream>>>(f->B, f->cur, f->snapSlab, f->snapWeight, f->snapCount, f->snapFov, f->snapUnused);
  HIPCHK(hipGetLastError());
  f->nZ = f->snapNZ;
  return RFSGPU_OK;
}
// ---- ring of pre-seeded states (bench.py): the input of a timed step is resident BEFORE the timed region -----------------------
// rfsgpu_restore_state inside a timed loop is a 45 MB copy per step at configs[1] (restore_state_kernel 8.8 us of a 124 us step) that
// is not part of the path.  Instead: n slots, each a full copy of what restore_state writes (slab, particle weights, sizes, FOV
// counts, unused lists), filled from the snapshot once; rfsgpu_state_ring_next swaps the handle's current-state pointers with the
// next slot's -- host work only, nothing is launched.  A slot is consumed by the step that runs on it (the map update works in
// place); rfsgpu_state_ring_seed fills all of them again.
static void ring_seed_slot(rfsgpu_filter *f, rfsgpu_filter::RingSlot &r) {
  Buffers Bs = f->B;
  Bs.slab[f->cur] = r.slab; Bs.weight = r.weight; Bs.count = r.count; Bs.nInFov = r.fov; Bs.unusedMask = r.unused;
  Bs.N = f->snapN;
  if (f->B.npl == 7) restore_state_kernel<7><<<f->snapN, 256, 0, f->stream>>>(Bs, f->cur, f->snapSlab, f->snapWeight, f->snapCount, f->snapFov, f->snapUnused);
  else restore_state_kernel<11><<<f->snapN, 256, 0, f->stream>>>(Bs, f->cur, f->snapSlab, f->snapWeight, f->snapCount, f->snapFov, f->snapUnused);
}
int rfsgpu_state_ring_seed(rfsgpu_filter *f) {
  CHECK_HANDLE(f);
  if (!f->snapSlab || f->stateRing.empty()) return fail(f, RFSGPU_ERR_INVALID, "state_ring_seed: no snapshot / no ring");
  hipSetDevice(f->device);
  for (auto &r : f->stateRing) ring_seed_slot(f, r);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(f->stream));
  f->stateRingPos = 0;
  return RFSGPU_OK;
}
int rfsgpu_state_ring_create(rfsgpu_filter *f, int n_slots) {
  CHECK_HANDLE(f);
  if (!f->snapSlab) return fail(f, RFSGPU_ERR_INVALID, "state_ring_create: rfsgpu_save_state first");
  if (n_slots < 0) return fail(f, RFSGPU_ERR_INVALID, "state_ring_create: negative slot count");
  hipSetDevice(f->device);
  HIPCHK(hipStreamSynchronize(f->stream));
  const size_t slabBytes = (size_t)f->Ncap * f->B.npl * f->cap * sizeof(double);
  while ((int)f->stateRing.size() > n_slots) {
    auto &r = f->stateRing.back();
    hipFree(r.slab); hipFree(r.weight); hipFree(r.count); hipFree(r.fov); hipFree(r.unused);
    f->stateRing.pop_back();
  }
  while ((int)f->stateRing.size() < n_slots) {
    rfsgpu_filter::RingSlot r;
    bool ok = true;
    ok &= hipMalloc(&r.slab, slabBytes) == hipSuccess;
    ok &= hipMalloc(&r.weight, f->Ncap * sizeof(double)) == hipSuccess;
    ok &= hipMalloc(&r.count, f->Ncap * sizeof(int)) == hipSuccess;
    ok &= hipMalloc(&r.fov, f->Ncap * sizeof(int)) == hipSuccess;
    ok &= hipMalloc(&r.unused, f->Ncap * sizeof(unsigned long long)) == hipSuccess;
    if (!ok) {
      hipFree(r.slab); hipFree(r.weight); hipFree(r.count); hipFree(r.fov); hipFree(r.unused);
      (void)hipGetLastError();
      return fail(f, RFSGPU_ERR_HIP, "state_ring_create: out of device memory");
    }
    // (rows beyond the live entries are never read before they are written; the small arrays are written in full by the seed)
    f->stateRing.push_back(r);
  }
  return n_slots > 0 ? rfsgpu_state_ring_seed(f) : RFSGPU_OK;
}
int rfsgpu_state_ring_next(rfsgpu_filter *f) {
  CHECK_HANDLE(f);
  if (f->stateRing.empty()) return fail(f, RFSGPU_ERR_INVALID, "state_ring_next: no ring");
  if (f->stateRingPos >= f->stateRing.size()) return fail(f, RFSGPU_ERR_INVALID, "state_ring_next: every slot has been consumed (rfsgpu_state_ring_seed fills them again)");
  auto &r = f->stateRing[f->stateRingPos++];
  std::swap(f->B.slab[f->cur], r.slab);
  std::swap(f->B.weight, r.weight);
  std::swap(f->B.count, r.count);
  std::swap(f->B.nInFov, r.fov);
  std::swap(f->B.unusedMask, r.unused);
  f->N = f->snapN;
  f->B.N = f->N;
  f->nZ = f->snapNZ;
  return RFSGPU_OK;
}
int rfsgpu_last_step_variant(const rfsgpu_filter *f, int *out4) {
  if (!f || !out4) return RFSGPU_ERR_INVALID;
  for (int k = 0; k < 4; k++) out4[k] = f->lastStepVariant[k];
  return RFSGPU_OK;
}
int rfsgpu_last_kernel_ns(rfsgpu_filter *f, long long *ns4) {
  CHECK_HANDLE(f);
  if (!ns4) return RFSGPU_ERR_INVALID;
  for (int k = 0; k < 4; k++) ns4[k] = f->lastKernelNs[k];
  return RFSGPU_OK;
}

#ifdef RFS_PROFILE
// tuning builds only (not declared in rfsgpu.h): 64 s_memtime stamps of one particle's last kernels
int rfsgpu_debug_sections(rfsgpu_filter *f, long long *out64) {
  CHECK_HANDLE(f);
  hipSetDevice(f->device);
  if (!f->B.dbg) {
    HIPCHK(hipMalloc(&f->B.dbg, (64 + 8 * (size_t)f->Ncap) * sizeof(long long)));
    HIPCHK(hipMemset(f->B.dbg, 0, (64 + 8 * (size_t)f->Ncap) * sizeof(long long)));
  }
  HIPCHK(hipMemcpy(out64, f->B.dbg, 64 * sizeof(long long), hipMemcpyDeviceToHost));
  return RFSGPU_OK;
}
// four counters per particle written by the last instrumented kernel (see the RFS_PROFILE blocks in the kernels)
int rfsgpu_debug_per_particle(rfsgpu_filter *f, long long *out4n) {
  CHECK_HANDLE(f);
  if (!f->B.dbg) return RFSGPU_ERR_INVALID;
  hipSetDevice(f->device);
  HIPCHK(hipMemcpy(out4n, f->B.dbg + 64, 4 * (size_t)f->N * sizeof(long long), hipMemcpyDeviceToHost));
  return RFSGPU_OK;
}
// second block of four: the fused step kernel's phase clock per particle (start, after map update, after weighting, end)
int rfsgpu_debug_per_particle_fused(rfsgpu_filter *f, long long *out4n) {
  CHECK_HANDLE(f);
  if (!f->B.dbg) return RFSGPU_ERR_INVALID;
  hipSetDevice(f->device);
  HIPCHK(hipMemcpy(out4n, f->B.dbg + 64 + 4 * (size_t)f->N, 4 * (size_t)f->N * sizeof(long long), hipMemcpyDeviceToHost));
  return RFSGPU_OK;
}
#endif

// ---- FastSLAM 1.0 (include/FastSLAM.hpp) on the same handle ----------------------------------------------------------
void rfsgpu_default_fastslam_config(rfsgpu_fastslam_config *c) {  // constructor defaults, :243-257 (nParticlesMax = 3n is set per handle)
  if (!c) return;
  c->minUpdatesBeforeResample = 1;
  c->minMeasurementsBeforeResample = 1;
  c->landmarkExistencePrior = 0.5;
  c->mapExistencePruneThreshold = -3.0;
  c->minLogMeasurementLikelihood = -10.0;
  c->nParticlesMax = 0;
  c->maxNDataAssocHypotheses = 1;
  c->maxDataAssocLogLikelihoodDiff = 5;
  c->landmarkCandidateMeasurementSupportDist = 1;
  c->landmarkCandidateMeasurementCountThreshold = 1;
  c->landmarkCandidateCurrentMeasurementCountThreshold = 1;
  c->landmarkCandidateMeasurementCheckThreshold = 2;
  c->landmarkLockWeight = 10;
  c->pruningMeasurementsThreshold = 0;
}
int rfsgpu_set_fastslam_config(rfsgpu_filter *f, const rfsgpu_fastslam_config *cfg) {
  CHECK_HANDLE(f);
  if (!cfg) return RFSGPU_ERR_INVALID;
  f->fs = *cfg;
  return RFSGPU_OK;
}
int rfsgpu_get_fastslam_config(const rfsgpu_filter *f, rfsgpu_fastslam_config *cfg) {
  if (!f || !cfg) return RFSGPU_ERR_INVALID;
  *cfg = f->fs;
  return RFSGPU_OK;
}
static FsParams fs_params(const rfsgpu_filter *f, int n_z) {
  FsParams F;
  F.prior = f->fs.landmarkExistencePrior;
  F.minLog = f->fs.minLogMeasurementLikelihood;
  F.lockW = f->fs.landmarkLockWeight;
  // clutterIntensityIntegral(nZ) / nZ (:561-562): c * sensing area (RngBrg.cpp:175-178) or the expected clutter number (VictoriaPark)
  F.pfa = ((f->D == 2) ? f->P.clutter * (2 * RFS_PI * (f->P.rmax - f->P.rmin)) : f->P.vpExpClutter) / n_z;
  F.newW = log(F.prior / (1 - F.prior));
  F.supportD2 = f->fs.landmarkCandidateMeasurementSupportDist * f->fs.landmarkCandidateMeasurementSupportDist;
  F.countThr = f->fs.landmarkCandidateMeasurementCountThreshold;
  if (F.countThr != 1u) const_cast<rfsgpu_filter *>(f)->candUsed = true;   // (migration rows carry the lists from now on)
  F.curThr = f->fs.landmarkCandidateCurrentMeasurementCountThreshold;
  F.checkThr = f->fs.landmarkCandidateMeasurementCheckThreshold;
  return F;
}
// prune by the existence threshold (:611-612) + new landmarks / candidates (:615-690) over all f->N particles
static int fastslam_map_management(rfsgpu_filter *f, const FsParams &F, int n_z) {
  int rc;
  if ((unsigned)n_z >= f->fs.pruningMeasurementsThreshold) {
    Params Pp = f->P;
    Pp.pruneT = f->fs.mapExistencePruneThreshold;
    const size_t pb = gm_prune_lds_bytes_per_wave(f->cap);
    if ((rc = set_lds(f, (gm_prune_kernel<4, false>), 4 * pb)) != RFSGPU_OK) return rc;
    gm_prune_kernel<4, false><<<(f->N + 3) / 4, 256, 4 * pb, f->stream>>>(f->B, Pp, f->cur, f->cur ^ 1);
    HIPCHK(hipGetLastError());
    f->cur ^= 1;
  }
  if (f->D == 2) fs_new_landmarks_kernel<2><<<(f->N + FS_NEWLM_WPB - 1) / FS_NEWLM_WPB, 64 * FS_NEWLM_WPB, 0, f->stream>>>(f->B, f->P, F, f->cur, n_z);
  else fs_new_landmarks_kernel<3><<<(f->N + FS_NEWLM_WPB - 1) / FS_NEWLM_WPB, 64 * FS_NEWLM_WPB, 0, f->stream>>>(f->B, f->P, F, f->cur, n_z);
  HIPCHK(hipGetLastError());
  return RFSGPU_OK;
}
// Multi-hypothesis update (fastslam_mh.h).  Host part: the slots of the particle copies, in particle order.
static int fastslam_update_mh(rfsgpu_filter *f, const double *z, int n_z) {
  if (f->D == 3 && f->B.nScan < 2) return fail(f, RFSGPU_ERR_INVALID, "Victoria Park model: rfsgpu_set_laser_scan must precede the update");
  int rc = stage_measurements(f, z, n_z);
  if (rc != RFSGPU_OK) return rc;
  hipSetDevice(f->device);
  const long long t0 = now_ns();
  const FsMhLayout L = fs_mh_layout();
  if (!f->mhArena) HIPCHK(hipMalloc(&f->mhArena, (size_t)f->Ncap * L.total));
  if (!f->mhInts) HIPCHK(hipMalloc(&f->mhInts, (size_t)5 * f->Ncap * sizeof(int)));
  const FsParams F = fs_params(f, n_z);
  const int N0 = f->N, kmax = (int)f->fs.maxNDataAssocHypotheses;
  HIPCHK(hipEventRecord(f->ev[EV_UM0], f->stream));
  if (f->D == 2) fs_mh_associate_kernel<2><<<N0, 64 * FSMH_WAVES, 0, f->stream>>>(f->B, f->P, F, f->cur, n_z, kmax, f->fs.maxDataAssocLogLikelihoodDiff, f->mhArena);
  else fs_mh_associate_kernel<3><<<N0, 64 * FSMH_WAVES, 0, f->stream>>>(f->B, f->P, F, f->cur, n_z, kmax, f->fs.maxDataAssocLogLikelihoodDiff, f->mhArena);
  HIPCHK(hipGetLastError());
  std::vector<int> nH(N0);
  HIPCHK(hipMemcpy2DAsync(nH.data(), sizeof(int), f->mhArena + L.offHdr + 2 * sizeof(int), L.total, sizeof(int), N0, hipMemcpyDeviceToHost, f->stream));
  if ((rc = check_device_errors(f)) != RFSGPU_OK) return rc;  // syncs
  // pi[0] = i; after particle i's nH - 1 copies were appended: pi[h] = nParticles_ - h (:543-556)
  std::vector<int> slotSrc(N0), slotHyp(N0), slotNH(N0), cDst, cSrc;
  int n = N0;
  for (int i = 0; i < N0; i++) { slotSrc[i] = i; slotHyp[i] = nH[i] > 0 ? 0 : -1; slotNH[i] = nH[i]; }
  for (int i = 0; i < N0; i++) {
    if (nH[i] <= 1) continue;
    const int first = n;
    n += nH[i] - 1;
    if (n > f->Ncap) return fail(f, RFSGPU_ERR_CAPACITY, "the particle copies of the multi-hypothesis update exceed max_particles (rfsgpu_create_ex)");
    slotSrc.resize(n); slotHyp.resize(n); slotNH.resize(n);
    for (int h = 1; h < nH[i]; h++) {
      const int slot = n - h;
      slotSrc[slot] = i; slotHyp[slot] = h; slotNH[slot] = nH[i];
      cDst.push_back(slot); cSrc.push_back(i);
    }
    (void)first;
  }
  int *dSlotSrc = f->mhInts, *dSlotHyp = dSlotSrc + f->Ncap, *dSlotNH = dSlotHyp + f->Ncap, *dDst = dSlotNH + f->Ncap, *dSrc = dDst + f->Ncap;
  HIPCHK(hipMemcpyAsync(dSlotSrc, slotSrc.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, f->stream));
  HIPCHK(hipMemcpyAsync(dSlotHyp, slotHyp.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, f->stream));
  HIPCHK(hipMemcpyAsync(dSlotNH, slotNH.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, f->stream));
  const int nNew = n - N0;
  if (nNew > 0) {
    HIPCHK(hipMemcpyAsync(dDst, cDst.data(), (size_t)nNew * sizeof(int), hipMemcpyHostToDevice, f->stream));
    HIPCHK(hipMemcpyAsync(dSrc, cSrc.data(), (size_t)nNew * sizeof(int), hipMemcpyHostToDevice, f->stream));
    fs_mh_copy_kernel<<<nNew, 256, 0, f->stream>>>(f->B, f->cur, dDst, dSrc, f->fsResampleOccured ? 1 : 0, f->P.poseCovStride);
    HIPCHK(hipGetLastError());
    f->N = n;
    f->B.N = n;
    fs_mh_split_weights_kernel<<<(n + 255) / 256, 256, 0, f->stream>>>(f->B.weight, dSlotSrc, dSlotNH, n, 0);
    fs_mh_split_weights_kernel<<<(n + 255) / 256, 256, 0, f->stream>>>(f->B.weight, dSlotSrc, dSlotNH, n, 1);
    HIPCHK(hipGetLastError());
  }
  if (f->D == 2) fs_mh_apply_kernel<2><<<n, 64, 0, f->stream>>>(f->B, f->P, F, f->cur, n_z, dSlotSrc, dSlotHyp, f->mhArena);
  else fs_mh_apply_kernel<3><<<n, 64, 0, f->stream>>>(f->B, f->P, F, f->cur, n_z, dSlotSrc, dSlotHyp, f->mhArena);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(f->ev[EV_UM1], f->stream));
  if ((rc = fastslam_map_management(f, F, n_z)) != RFSGPU_OK) return rc;
  HIPCHK(hipEventRecord(f->ev[EV_PR1], f->stream));
  f->parents = slotSrc;
  rc = check_device_errors(f);
  accumulate(f->ev[EV_UM0], f->ev[EV_UM1], f->timing.mapUpdate_wall, &f->lastKernelNs[0]);
  f->timing.mapUpdate_kf_wall = f->timing.mapUpdate_wall;
  accumulate(f->ev[EV_UM1], f->ev[EV_PR1], f->timing.mapPrune_wall, &f->lastKernelNs[3]);
  f->timing.mapUpdate_cpu += now_ns() - t0;
  return rc;
}
int rfsgpu_fastslam_set_resample_occured(rfsgpu_filter *f, int flag) {
  CHECK_HANDLE(f);
  f->fsResampleOccured = flag != 0;
  return RFSGPU_OK;
}
int rfsgpu_particle_parents(rfsgpu_filter *f, int *parent, int max_n) {
  CHECK_HANDLE(f);
  if (!parent || max_n < f->N) return RFSGPU_ERR_INVALID;
  for (int k = 0; k < f->N; k++) parent[k] = (k < (int)f->parents.size()) ? f->parents[k] : k;
  return RFSGPU_OK;
}
int rfsgpu_fastslam_update(rfsgpu_filter *f, const double *z, int n_z) {
  CHECK_HANDLE(f);
  f->holes = false;
  f->fastSlamHandle = true;
  if (n_z == 0) return RFSGPU_OK;  // :401-402
  if (f->D == 3 && f->B.nScan < 2) return fail(f, RFSGPU_ERR_INVALID, "Victoria Park model: rfsgpu_set_laser_scan must precede the update");
  if (f->fs.maxNDataAssocHypotheses < 1 || f->fs.maxNDataAssocHypotheses > FSMH_MAX_HYP)
    return fail(f, RFSGPU_ERR_UNSUPPORTED, "maxNDataAssocHypotheses must be in [1, 16]");
  if (f->fs.maxNDataAssocHypotheses > 1) return fastslam_update_mh(f, z, n_z);
  int rc = stage_measurements(f, z, n_z);
  if (rc != RFSGPU_OK) return rc;
  hipSetDevice(f->device);
  if (!f->fsArena) HIPCHK(hipMalloc(&f->fsArena, (size_t)f->Ncap * fs_arena_bytes()));
  const FsParams F = fs_params(f, n_z);
  const long long t0 = now_ns();
  HIPCHK(hipEventRecord(f->ev[EV_UM0], f->stream));
  const size_t per = fs_lds_bytes_per_wave(f->cap);
  const size_t b2 = (size_t)(3 * RFSGPU_MAX_Z * 8) + 2 * per, b1 = (size_t)(3 * RFSGPU_MAX_Z * 8) + per;
  if (f->D == 2) {
    if (b2 <= 64 * 1024) {
      if ((rc = set_lds(f, (fs_associate_update_kernel<2, 2>), b2)) != RFSGPU_OK) return rc;
      fs_associate_update_kernel<2, 2><<<(f->N + 1) / 2, 128, b2, f->stream>>>(f->B, f->P, F, f->cur, n_z, f->fsArena);
    } else {
      if ((rc = set_lds(f, (fs_associate_update_kernel<1, 2>), b1)) != RFSGPU_OK) return rc;
      fs_associate_update_kernel<1, 2><<<f->N, 64, b1, f->stream>>>(f->B, f->P, F, f->cur, n_z, f->fsArena);
    }
  } else {
    if ((rc = set_lds(f, (fs_associate_update_kernel<1, 3>), b1)) != RFSGPU_OK) return rc;
    fs_associate_update_kernel<1, 3><<<f->N, 64, b1, f->stream>>>(f->B, f->P, F, f->cur, n_z, f->fsArena);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(f->ev[EV_UM1], f->stream));
  if ((rc = fastslam_map_management(f, F, n_z)) != RFSGPU_OK) return rc;
  HIPCHK(hipEventRecord(f->ev[EV_PR1], f->stream));
  f->parents.clear();
  rc = check_device_errors(f);  // syncs
  // FastSLAM::TimingInfo buckets folded onto the handle's: data association + KF + weighting (one kernel) under
  // mapUpdate / mapUpdate_kf, prune + new landmarks ("map management", :606-693) under mapPrune
  accumulate(f->ev[EV_UM0], f->ev[EV_UM1], f->timing.mapUpdate_wall, &f->lastKernelNs[0]);
  f->timing.mapUpdate_kf_wall = f->timing.mapUpdate_wall;
  accumulate(f->ev[EV_UM1], f->ev[EV_PR1], f->timing.mapPrune_wall, &f->lastKernelNs[3]);
  f->timing.mapUpdate_cpu += now_ns() - t0;
  return rc;
}

static double g_matPermLastKernelMs = 0.0;
double rfsgpu_mat_perm_last_kernel_ms(void) { return g_matPermLastKernelMs; }
int rfsgpu_mat_perm(const double *A, int n, int batch, double *out, int device_id) {
  if (!A || !out || n < 1 || n > 24 || batch < 0) return RFSGPU_ERR_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return RFSGPU_ERR_NO_DEVICE;
  if (batch == 0) return RFSGPU_OK;
  if (hipSetDevice(device_id) != hipSuccess) return RFSGPU_ERR_NO_DEVICE;
  double *dA = nullptr, *dO = nullptr;
  const size_t bytes = (size_t)batch * n * n * sizeof(double);
  if (hipMalloc(&dA, bytes) != hipSuccess) return RFSGPU_ERR_HIP;
  if (hipMalloc(&dO, (size_t)batch * sizeof(double)) != hipSuccess) { hipFree(dA); return RFSGPU_ERR_HIP; }
  int rc = RFSGPU_OK;
  if (hipMemcpy(dA, A, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = RFSGPU_ERR_HIP;
  if (rc == RFSGPU_OK) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool timed = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
    if (timed) hipEventRecord(e0, 0);
    if (mat_perm_launch(dA, n, batch, dO) != hipSuccess) rc = RFSGPU_ERR_HIP;
    if (timed) hipEventRecord(e1, 0);
    if (hipDeviceSynchronize() != hipSuccess) rc = RFSGPU_ERR_HIP;
    float ms = 0.f;
    if (timed && rc == RFSGPU_OK && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) g_matPermLastKernelMs = (double)ms;
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
  }
  if (rc == RFSGPU_OK && hipMemcpy(out, dO, (size_t)batch * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = RFSGPU_ERR_HIP;
  hipFree(dA);
  hipFree(dO);
  return rc;
}

}  // extern "C"

#include "group.h"
