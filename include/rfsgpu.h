/*
 * rfsgpu.h -- C ABI of the MI355X-native RB-PHD SLAM update engine.
 *
 * This is the drop-in boundary for ONE hot path of kykleung/RFS-SLAM: the per-particle
 * Gaussian-mixture PHD map update + particle weighting + GM merge/prune that runs inside
 * rfs::RBPHDFilter<...>::update() (reference include/RBPHDFilter.hpp:444-541), plus the few
 * map-side pieces either side of it (static landmark predict + birth Gaussians, resample copy).
 *
 * The reference has no FFI / plugin registry; the surface a replacement has to sit behind is the
 * public interface of the RBPHDFilter class template (include/RBPHDFilter.hpp:72-251) and what it
 * inherits from ParticleFilter (include/ParticleFilter.hpp:72-186).  Every entry point below
 * names the reference member it replaces.  INTEGRATION.md shows the binding a reference
 * maintainer would add.
 *
 * Conventions
 *   - plain C, opaque handle, int status returns (0 = RFSGPU_OK), no exceptions / STL / torch types.
 *   - all host buffers are caller-owned; the engine owns its device memory (one handle == one GPU,
 *     one host thread per handle, one process per GPU).
 *   - everything is fp64; matrices are row-major; d_m = landmark dim, d_z = measurement dim
 *     (both 2 for RFSGPU_MODEL_RNGBRG_2D, both 3 for RFSGPU_MODEL_VICTORIAPARK_3D).
 *   - "slot" = particle index 0..n_particles-1.
 *   - the engine fails loudly: there is NO CPU fallback anywhere behind this ABI.
 *
 * STABLE CORE -- the 24 entry points a reference-side binding needs (round 5: + rfsgpu_update_io, the update with its inputs and outputs in one call); integration/RBPHDFilter_rfsgpu.hpp (the class template that
 * stands where rfs::RBPHDFilter stands, compiled under the unmodified reference drivers) uses 20 of them and nothing else
 * (tests/test_abi.py checks the list against the linked drivers):
 *   RFSGPU_CORE: rfsgpu_abi_version rfsgpu_create rfsgpu_destroy rfsgpu_last_error rfsgpu_default_filter_config
 *   RFSGPU_CORE: rfsgpu_set_filter_config rfsgpu_set_model_rngbrg rfsgpu_set_model_victoriapark rfsgpu_set_laser_scan
 *   RFSGPU_CORE: rfsgpu_set_kf_config rfsgpu_set_lmk_process_noise rfsgpu_set_poses rfsgpu_set_weights rfsgpu_get_weights
 *   RFSGPU_CORE: rfsgpu_predict_map rfsgpu_update rfsgpu_resample_apply rfsgpu_set_birth_inheritance rfsgpu_gm_size
 *   RFSGPU_CORE: rfsgpu_get_landmark rfsgpu_get_timing rfsgpu_set_phase_timing rfsgpu_synchronize rfsgpu_update_io
 * The same binding over SEVERAL GPUs (RFSGPU_DEVICES=0,1,...: integration/RBPHDFilter_rfsgpu.hpp, rfsgpu_engine_facade) uses the group
 * counterparts of those calls and nothing else:
 *   RFSGPU_CORE_MULTI: rfsgpu_group_create rfsgpu_group_destroy rfsgpu_group_last_error rfsgpu_group_set_filter_config
 *   RFSGPU_CORE_MULTI: rfsgpu_group_set_model_rngbrg rfsgpu_group_set_model_victoriapark rfsgpu_group_set_laser_scan rfsgpu_group_set_kf_config
 *   RFSGPU_CORE_MULTI: rfsgpu_group_set_lmk_process_noise rfsgpu_group_set_poses rfsgpu_group_set_weights rfsgpu_group_get_weights
 *   RFSGPU_CORE_MULTI: rfsgpu_group_predict_map rfsgpu_group_update rfsgpu_group_apply_plan rfsgpu_group_gm_size rfsgpu_group_get_landmark
 *   RFSGPU_CORE_MULTI: rfsgpu_group_get_timing rfsgpu_group_set_phase_timing rfsgpu_group_update_io
 * Everything else is OPTIONAL and grouped below by who needs it:
 *   [async]    stream-ordered forms for host loops that pipeline (rfsgpu_*_async, rfsgpu_step_async, rfsgpu_set_stream, ...);
 *   [multi]    several GPUs: rfsgpu_group_* (one host thread), slab rows / device pointers (one process per GPU over RCCL);
 *   [state]    state injection and probes for tests and tools (import / export of mixtures, candidate lists, ids, masks);
 *   [bench]    snapshot / restore, kernel timing statistics, debug counters, test probes: declared only under
 *              RFSGPU_ENABLE_BENCH_API (a maintainer reading this header for the binding never meets them);
 *   [fastslam] FastSLAM / MH-FastSLAM on the same handle, the optional device-side Ackerman propagation, rfsgpu_mat_perm.
 * A maintainer wiring the reference to the library reads the CORE entries and INTEGRATION.md; nothing optional is required
 * for correct results.
 */
#ifndef RFSGPU_H
#define RFSGPU_H

#include <stddef.h>   /* size_t */

#ifdef __cplusplus
extern "C" {
#endif

#define RFSGPU_ABI_VERSION 1

/* Maximum measurements per update() (one 64-bit association mask per landmark). */
/* Hard limits of the device path (the reference has none; all are refused LOUDLY -- an error status, never a silent
 * truncation -- and all are far above BASELINE.json's configurations): RFSGPU_MAX_Z measurements per update,
 * RFSGPU_MAX_EVAL evaluation points, Murty extended dimension nR + nC <= 64, gm_capacity <= 2048 Gaussians per particle,
 * RFSGPU_MAX_CANDIDATES birth candidates per particle on the RB-PHD path (64 landmark candidates on the FastSLAM path). */
#define RFSGPU_MAX_Z 64
/* Maximum evaluation points for the multi-feature particle weight. */
#define RFSGPU_MAX_EVAL 64

typedef struct rfsgpu_filter rfsgpu_filter;

enum rfsgpu_status {
  RFSGPU_OK = 0,
  RFSGPU_ERR_INVALID = 1,     /* bad argument / bad index                                      */
  RFSGPU_ERR_HIP = 2,         /* a HIP runtime call failed (rfsgpu_last_error has the string)  */
  RFSGPU_ERR_CAPACITY = 3,    /* a particle's Gaussian mixture outgrew gm_capacity             */
  RFSGPU_ERR_NO_DEVICE = 4,   /* no gfx950 device visible                                      */
  RFSGPU_ERR_UNSUPPORTED = 5  /* configuration outside what the device path implements         */
};

enum rfsgpu_model {
  RFSGPU_MODEL_RNGBRG_2D = 0,       /* MeasurementModel_RngBrg + KalmanFilter_RngBrg + Landmark2d (d_m = d_z = 2)            */
  RFSGPU_MODEL_VICTORIAPARK_3D = 1  /* MeasurementModel_VictoriaPark + KalmanFilter_VictoriaPark + Landmark3d (d_m = d_z = 3:
                                       landmark (x, y, diameter), measurement (range, bearing, diameter))                    */
};

/* Maximum entries of the Victoria Park Pd table / beams of a laser scan / birth candidates per particle. */
#define RFSGPU_VP_MAX_PD 16
#define RFSGPU_VP_MAX_SCAN 720
#define RFSGPU_MAX_CANDIDATES 256

/* Mirrors RBPHDFilter::Config, include/RBPHDFilter.hpp:90-146 (same meaning, same defaults
 * :370-382 when filled by rfsgpu_default_filter_config). */
typedef struct rfsgpu_filter_config {
  double birthGaussianWeight;
  unsigned int birthGaussianMeasurementCountThreshold;
  unsigned int birthGaussianMeasurementCheckThreshold;
  double birthGaussianMeasurementSupportDist;
  unsigned int birthGaussianCurrentMeasurementCountThreshold;
  double newGaussianCreateInnovMDThreshold;
  int importanceWeightingEvalPointCount;          /* -1 => all (reference :737 unsigned compare) */
  double importanceWeightingEvalPointGuassianWeight;
  double importanceWeightingMeasurementLikelihoodMDThreshold;
  double gaussianMergingThreshold;
  double gaussianMergingCovarianceInflationFactor;
  double gaussianPruningThreshold;
  int minUpdatesBeforeResample;
  int minMeasurementsBeforeResample;
  int useClusterProcess;                          /* bool in the reference */
} rfsgpu_filter_config;

/* Mirrors MeasurementModel_RngBrg::Config (include/MeasurementModel_RngBrg.hpp:65-71) plus the
 * additive noise R set through MeasurementModel::setNoise (src/rbphdslam2dSim.cpp:466-469). */
typedef struct rfsgpu_rngbrg_config {
  double R[4];                     /* 2x2 row-major measurement covariance */
  double probabilityOfDetection;
  double uniformClutterIntensity;
  double rangeLimMax;
  double rangeLimMin;
  double rangeLimBuffer;
} rfsgpu_rngbrg_config;

/* Mirrors MeasurementModel_VictoriaPark::Config (include/MeasurementModel_VictoriaPark.hpp:150-158) plus the noise set
 * through setNoise(R, Slb) (src/MeasurementModel_VictoriaPark.cpp:66-73).  Angles in radians. */
typedef struct rfsgpu_vp_config {
  double R[9];                       /* 3x3 row-major measurement covariance (range, bearing, diameter) */
  double Slb;                        /* variance of the lidar beam angle                                 */
  double PdTable[RFSGPU_VP_MAX_PD];  /* config.probabilityOfDetection_ (Pd by number of visible beams)   */
  int nPd;
  double expectedClutterNumber;
  double rangeLimMax, rangeLimMin;
  double bearingLimitMax, bearingLimitMin;
  double bufferZonePd;
} rfsgpu_vp_config;

/* Mirrors KalmanFilter_RngBrg::Config (include/KalmanFilter_RngBrg.hpp:55-60). <=0 disables. */
typedef struct rfsgpu_kf_config {
  double rangeInnovationThreshold;
  double bearingInnovationThreshold;
} rfsgpu_kf_config;

/* Mirrors FastSLAM::Config (include/FastSLAM.hpp:106-132), same member names without the trailing '_'.
 * Defaults: FastSLAM constructor (:243-257). */
typedef struct rfsgpu_fastslam_config {
  int minUpdatesBeforeResample;
  int minMeasurementsBeforeResample;
  double landmarkExistencePrior;
  double mapExistencePruneThreshold;
  double minLogMeasurementLikelihood;
  int nParticlesMax;
  unsigned maxNDataAssocHypotheses;
  double maxDataAssocLogLikelihoodDiff;
  double landmarkCandidateMeasurementSupportDist;
  unsigned landmarkCandidateMeasurementCountThreshold;
  unsigned landmarkCandidateCurrentMeasurementCountThreshold;
  unsigned landmarkCandidateMeasurementCheckThreshold;
  double landmarkLockWeight;
  unsigned pruningMeasurementsThreshold;
} rfsgpu_fastslam_config;

/* Mirrors RBPHDFilter::TimingInfo (include/RBPHDFilter.hpp:152-167), nanoseconds, accumulated.
 * *_wall come from HIP events around each phase's kernels; *_cpu is the host time spent inside
 * the corresponding ABI calls (launch + sync). */
typedef struct rfsgpu_timing {
  long long predict_wall, predict_cpu;
  long long mapUpdate_wall, mapUpdate_cpu;
  long long mapUpdate_kf_wall, mapUpdate_kf_cpu;
  long long particleWeighting_wall, particleWeighting_cpu;
  long long mapMerge_wall, mapMerge_cpu;
  long long mapPrune_wall, mapPrune_cpu;
  long long particleResample_wall, particleResample_cpu;
} rfsgpu_timing;

/* ---- [core] lifetime ------------------------------------------------------------------------------ */

/* ABI version of the loaded library (== RFSGPU_ABI_VERSION). */
int rfsgpu_abi_version(void);

/* Replaces the RBPHDFilter constructor (include/RBPHDFilter.hpp:350-393): n_particles particles
 * of weight 1, pose 0, empty maps, default config.  device_id: HIP ordinal.  gm_capacity: per-particle
 * Gaussian slots (rounded up to a multiple of 64); must cover the transient size right after the
 * map update (nM + new Gaussians).  Exceeding it sets RFSGPU_ERR_CAPACITY, never corrupts memory. */
int rfsgpu_create(rfsgpu_filter **out, int model, int n_particles, int device_id, int gm_capacity);
/* The same with room for the particle set to GROW up to max_particles (>= n_particles): FastSLAM's multi-hypothesis
 * update multiplies particles (FastSLAM.hpp:543-551, ParticleFilter::copyParticle) until resampleWithMapCopy brings the
 * set back to its initial size.  rfsgpu_n_particles = nParticles_ now; every array argument of the other calls has that
 * many entries. */
int rfsgpu_create_ex(rfsgpu_filter **out, int model, int n_particles, int device_id, int gm_capacity, int max_particles);
int rfsgpu_n_particles(const rfsgpu_filter *f);
int rfsgpu_max_particles(const rfsgpu_filter *f);
/* Replaces ~RBPHDFilter (:395-403). */
void rfsgpu_destroy(rfsgpu_filter *f);
/* Human-readable text for the last non-OK status on this handle (never NULL). */
const char *rfsgpu_last_error(const rfsgpu_filter *f);

/* ---- [core] configuration (the three public config structs of the reference) ---------------------- */

void rfsgpu_default_filter_config(rfsgpu_filter_config *cfg);            /* RBPHDFilter.hpp:370-382 */
int rfsgpu_set_filter_config(rfsgpu_filter *f, const rfsgpu_filter_config *cfg);   /* filter.config.* */
int rfsgpu_get_filter_config(const rfsgpu_filter *f, rfsgpu_filter_config *cfg);
/* How rfsMeasurementLikelihood evaluates a partition with nR + nC > 8 (include/RBPHDFilter.hpp:920-959):
 *   RFSGPU_PARTITION_MURTY200 (default) the reference's way, bug-compatible: Murty's ranked assignments on the extended matrix,
 *                             sum of at most 200 terms, stop below score -1000 (murty.h);
 *   RFSGPU_PARTITION_EXACT    the untruncated sum over all partial assignments (the quantity the truncation approximates) by a
 *                             subset recurrence over the smaller side of the partition (<= 9 items; larger partitions still
 *                             go to Murty).  Not the reference's numbers: an opt-in, timed beside the default. */
#define RFSGPU_PARTITION_MURTY200 0
#define RFSGPU_PARTITION_EXACT 1
int rfsgpu_set_partition_mode(rfsgpu_filter *f, int mode);
int rfsgpu_get_partition_mode(const rfsgpu_filter *f);
int rfsgpu_set_model_rngbrg(rfsgpu_filter *f, const rfsgpu_rngbrg_config *cfg);    /* getMeasurementModel()->config / setNoise */
int rfsgpu_set_kf_config(rfsgpu_filter *f, const rfsgpu_kf_config *cfg);           /* getKalmanFilter()->config */
/* Victoria Park model: getMeasurementModel()->config / setNoise(R, Slb) (src/rbphdslam_VictoriaPark.cpp:371-378). */
int rfsgpu_set_model_victoriapark(rfsgpu_filter *f, const rfsgpu_vp_config *cfg);
/* MeasurementModel_VictoriaPark::setLaserScan (src/MeasurementModel_VictoriaPark.cpp:267-281): the raw scan used by the
 * occlusion-based Pd and by the clutter intensity (expected clutter / field-of-view area); n <= RFSGPU_VP_MAX_SCAN. */
int rfsgpu_set_laser_scan(rfsgpu_filter *f, const double *scan, int n);
#ifdef RFSGPU_ENABLE_BENCH_API   /* [bench] / [test]: exported by the library, declared only for callers that ask (bench.py, the tests) */
/* Probe for tests: MeasurementModel_VictoriaPark::probabilityOfDetection (src/MeasurementModel_VictoriaPark.cpp:153-199) of the
 * first max_n Gaussians of particle `slot` under the current pose and scan, as the kernels evaluate it: Pd and the
 * isCloseToSensingLimit flag. */
int rfsgpu_vp_probe_pd(rfsgpu_filter *f, int slot, double *pd, int *close_to_limit, int max_n);
#endif /* RFSGPU_ENABLE_BENCH_API */
/* getLmkProcessModel()->setNoise(Q) (include/ProcessModel.hpp:195-208); Q is d_m x d_m. */
int rfsgpu_set_lmk_process_noise(rfsgpu_filter *f, const double *Q);

/* ---- [core] particle state crossing the boundary --------------------------------------------------- */

/* Poses after the host-side ParticleFilter::propagate / setParticlePose
 * (include/ParticleFilter.hpp:322-341, RBPHDFilter.hpp:1181-1186).  x: 3 doubles per particle
 * (x, y, theta).  cov: 3x3 pose covariance (enters S, src/MeasurementModel_RngBrg.cpp:102);
 * cov_stride = 0 -> one shared 3x3 (or cov==NULL -> zero), 9 -> one per particle. */
int rfsgpu_set_poses(rfsgpu_filter *f, const double *x, const double *cov, int cov_stride);
int rfsgpu_get_poses(rfsgpu_filter *f, double *x);
/* Particle::setWeight / getWeight (include/Particle.hpp), n_particles doubles. */
int rfsgpu_set_weights(rfsgpu_filter *f, const double *w);
int rfsgpu_get_weights(rfsgpu_filter *f, double *w);

/* ---- [core] map access (getGMSize / getLandmark, RBPHDFilter.hpp:1152-1178) + [state] injection ------ */

int rfsgpu_gm_size(rfsgpu_filter *f, int slot);                        /* -1 on bad index */
/* returns RFSGPU_OK, or RFSGPU_ERR_INVALID on a bad index (reference returns false). */
int rfsgpu_get_landmark(rfsgpu_filter *f, int slot, int m, double *mean, double *cov, double *w);
/* Replace particle `slot`'s mixture with n Gaussians (GaussianMixture::addGaussian order). */
int rfsgpu_import_gm(rfsgpu_filter *f, int slot, int n, const double *w, const double *mean, const double *cov);
/* Copy out up to max_n Gaussians in storage order; *n_out = mixture size.  w_prev may be NULL. */
int rfsgpu_export_gm(rfsgpu_filter *f, int slot, int max_n, int *n_out, double *w, double *w_prev,
                     double *mean, double *cov);
/* Set particle `slot`'s birth bookkeeping (unused_measurements_[slot] as ascending indices, nLandmarksInFOV_[slot]);
 * needed when a particle migrates between shards during a multi-GPU resample (RBPHDFilter.hpp:1005-1011). */
int rfsgpu_import_aux(rfsgpu_filter *f, int slot, const int *unused_idx, int n_unused, int n_in_fov);
/* birthGaussians_[slot] (RBPHDFilter.hpp:253, :173-177): the particle's birth-Gaussian candidates in list order. */
int rfsgpu_export_birth_candidates(rfsgpu_filter *f, int slot, int max_n, int *n_out, double *mean, double *cov, int *n_support, int *n_checks);
int rfsgpu_import_birth_candidates(rfsgpu_filter *f, int slot, int n, const double *mean, const double *cov, const int *n_support, const int *n_checks);
/* All mixture sizes at once (n_particles ints). */
int rfsgpu_gm_sizes(rfsgpu_filter *f, int *sizes);

/* ---- [core] the hot path (+ its [async] forms) --------------------------------------------------------------------------- */

/* Map part of RBPHDFilter::predict (:415-442): if add_birth, addBirthGaussians (:1000-1084) from
 * the previous update's measurements / unused lists at the CURRENT (pre-propagation) poses, then
 * StaticProcessModel::staticStep (Sigma += Q) on every Gaussian (:433-439). */
int rfsgpu_predict_map(rfsgpu_filter *f, int add_birth);

/* The particle-parallel body of RBPHDFilter::update (:444-523): updateMap, importanceWeighting
 * (unless useClusterProcess), merge, prune.  z: n_z x d_z doubles.  n_z == 0 returns OK without
 * touching anything (:450-452).  Resampling / normalisation stay with the caller (below). */
int rfsgpu_update(rfsgpu_filter *f, const double *z, int n_z);
/* RBPHDFilter::update (:444-541) with everything it consumes and returns in ONE synchronous call: [optionally the predict that
 * precedes it, see rfsgpu_cycle_async: `predict` = RFSGPU_CYCLE_NO_PREDICT | 0 | 1], poses `x` (+ covariance; NULL = unchanged),
 * particle weights `w_in` (NULL = unchanged), the measurement set, and the updated (un-normalised) weights back in `w_out`
 * (NULL = not wanted).  One wait for the device, one error check: what set_poses + set_weights + update + get_weights do in four
 * calls and two waits.  The reference-side binding's update() is this call. */
int rfsgpu_update_io(rfsgpu_filter *f, int predict, const double *x, const double *x_cov, int cov_stride, const double *w_in,
                     const double *z, int n_z, double *w_out);
/* rfsgpu_update runs the 2-D model's step as ONE fused launch by default and books its time under TimingInfo::mapUpdate.
 * on != 0: the phases run as separate launches instead (updateMap, importanceWeighting, merge + prune), so that the
 * mapUpdate / particleWeighting / mapMerge buckets of RBPHDFilter::TimingInfo (:152-167) are filled separately, as the
 * reference's timing printout expects -- same results bit for bit, about a third more device time. */
int rfsgpu_set_phase_timing(rfsgpu_filter *f, int on);
/* Stream-ordered form of rfsgpu_update: enqueues the step and returns without waiting for the GPU.  With the 2-D model the
 * step is ONE kernel (a workgroup takes its particle through updateMap, importanceWeighting, merge and prune; results are
 * bit-identical to rfsgpu_update; the environment variable RFSGPU_FUSED_STEP=0 selects the three-kernel form).  Device-side
 * errors (capacity overflow, ...) surface at the next rfsgpu_synchronize() / synchronous call; per-phase HIP-event pairs of
 * up to RFSGPU_ASYNC_RING outstanding steps are harvested there into TimingInfo and rfsgpu_kernel_time_stats.  A host loop
 * that does not need the weights every step (no resampling decision pending) pipelines its steps with this. */
#define RFSGPU_ASYNC_RING 256
int rfsgpu_update_async(rfsgpu_filter *f, const double *z, int n_z);
/* The same step followed -- in the same post kernel that runs the Murty partitions -- by the weight reduction of
 * ParticleFilter::normalizeWeights (include/ParticleFilter.hpp:352-363): {sum w, sum w^2} of this shard go to the bound sums
 * buffer (rfsgpu_bind_weight_sums_buffer / rfsgpu_weight_sums_device_ptr); with normalize != 0 (a filter that lives on one
 * GPU) the weights are divided by the sum right there, otherwise the caller all-reduces the pair across its shards and calls
 * rfsgpu_normalize_weights.  Two launches per step (fused step + post) instead of five. */
int rfsgpu_step_async(rfsgpu_filter *f, const double *z, int n_z, int normalize);
/* One submission per predict + update cycle (round 5).  RBPHDFilter::predict's map part (include/RBPHDFilter.hpp:415-442: birth
 * Gaussians at the poses the previous update used, then StaticProcessModel::staticStep) -- `predict` = 1 with births, 0 without,
 * RFSGPU_CYCLE_NO_PREDICT for none --, then the host's new poses `x` (+ covariance; NULL = unchanged) and particle weights `w_in`
 * (NULL = unchanged), then RBPHDFilter::update's body (:444-523) with the post kernel's weight sums / division as in
 * rfsgpu_step_async.  Stream-ordered, the caller's buffers are copied before the call returns.  Where the configuration allows it
 * (2-D model, immediate births, no inheritance walk pending, n_z > 0) the predict runs at the head of the fused step kernel --
 * one launch chain per cycle, no separate predict launch; otherwise the stand-alone kernels are enqueued in the same order.
 * Same results either way as rfsgpu_predict_map + rfsgpu_set_poses + rfsgpu_set_weights + rfsgpu_step_async. */
/* [multi] rfsgpu_step_async(normalize = 0) for hosts that shard the particles over several GPUs, with the weight normalisation
 * trailing by one step: the post kernel divides the weights by *prev_total_dev ({sum w, sum w^2} of the PREVIOUS step over all
 * shards, device memory, NULL = none) before it takes this step's sums into the bound sums buffer, and the stream waits for
 * `wait_event` (a hipEvent_t recorded on another stream behind the collective that wrote that total; NULL = none) only between
 * the step kernel and the post kernel.  The one collective of the path then runs beside the next step kernel, not in front of
 * it.  (w L) / T instead of (w / T) L: an ulp apart from the call-by-call order.  A host that needs the normalised weights or
 * N_eff (the resample test, include/ParticleFilter.hpp:405-415) finishes with rfsgpu_normalize_weights(f, 0, total_dev). */
int rfsgpu_step_async_deferred(rfsgpu_filter *f, const double *z, int n_z, const void *prev_total_dev, void *wait_event);
/* [multi] The same hand-over without stream events (an event record and an event wait are a marker and a barrier packet on the step's
 * stream).  Per step: rfsgpu_step_async_trailing(h, z, n, total_dev, have_prev) on the engine's stream, then ON THE SIDE STREAM
 * rfsgpu_collective_gate(h, side) -> the collective of the shards' sums (the bound sums buffer) into total_dev ->
 * rfsgpu_collective_publish(h, side).  The gate kernel waits on the device for this step's sums, the next step's post kernel waits on
 * the device for the published total (bounded spins; a protocol that is never completed raises RFSGPU_ERR_UNSUPPORTED at the next
 * synchronising call instead of hanging).  have_prev = 0 for the first step of a run or after the pending total has been applied. */
int rfsgpu_step_async_trailing(rfsgpu_filter *f, const double *z, int n_z, const void *total_dev, int have_prev);
int rfsgpu_collective_gate(rfsgpu_filter *f, void *hip_stream);
int rfsgpu_collective_publish(rfsgpu_filter *f, void *hip_stream);
/* [multi] Whether the engine's stream and `hip_stream` make progress side by side on this runtime -- what the sequence-number form
 * above needs (its post kernel waits for a word a LATER submission on the other stream publishes; two streams mapped onto one
 * hardware queue serialise, and the wait can only run out: RFSGPU_ERR_UNSUPPORTED, "collective hand-over timed out").  Plays the
 * hand-over once with nothing at stake (bounded 0.2 s, synchronises both streams); *side_by_side = 1 | 0.  Probe once per stream
 * pair; on 0 use rfsgpu_step_async_deferred (stream events: two packets on the step's stream, ~+9 us per step) -- which is why both
 * forms stay.  rfsgpu_group_update_deferred and the Python hosts do exactly that. */
int rfsgpu_collective_probe(rfsgpu_filter *f, void *hip_stream, int *side_by_side);
#define RFSGPU_CYCLE_NO_PREDICT (-1)
int rfsgpu_cycle_async(rfsgpu_filter *f, int predict, const double *x, const double *x_cov, int cov_stride, const double *w_in,
                       const double *z, int n_z, int normalize);
/* The host side of an asynchronous filter loop (what the Victoria Park driver needs per lidar message: one call for the
 * inputs, one for the step, no host wait -- src/rbphdslam_VictoriaPark.cpp:555-583):
 *   rfsgpu_set_step_inputs_async  poses (+ covariance; x == NULL leaves them) and, Victoria Park model, the laser scan
 *                                 (MeasurementModel_VictoriaPark::setLaserScan; scan == NULL leaves it) through a pinned ring;
 *   rfsgpu_predict_map_async      rfsgpu_predict_map without the error-word readback.
 * Device-side errors (capacity, ...) of asynchronous calls surface at the next synchronising call (rfsgpu_synchronize,
 * rfsgpu_weight_sums, rfsgpu_get_weights, ...). */
int rfsgpu_set_step_inputs_async(rfsgpu_filter *f, const double *x, const double *cov, int cov_stride, const double *scan, int n_scan);
int rfsgpu_predict_map_async(rfsgpu_filter *f, int add_birth);
/* A run of n consecutive predicts that add no births (each one: rfsgpu_set_lmk_process_noise(Q_k) + rfsgpu_predict_map_async(f, 0),
 * i.e. StaticProcessModel::staticStep, Sigma += Q_k, include/ProcessModel.hpp:195-208) in ONE launch -- the additions are made
 * one after the other, so the result has the same bits.  noises: [n][D*D] row-major, one per predict of the run (the drivers
 * scale Q with the message interval), or NULL = the noise that is set now, n times.  A driver whose odometry messages outnumber
 * its sensor messages (Victoria Park: 9 to 1) collects the birth-less predicts and flushes them before the next call that
 * reads the maps. */
int rfsgpu_static_steps_async(rfsgpu_filter *f, int n, const double *noises);
/* ParticleFilter::propagate (include/ParticleFilter.hpp:322-339) for the Victoria Park driver's process model, on the device:
 * MotionModel_Ackerman2d::step (src/ProcessModel_Ackerman2D.cpp:47-78) applied to every particle's pose with its own noisy
 * input u + N(0, diag(var)) (ProcessModel::sample's input-noise branch, include/ProcessModel.hpp:126-150).  u = {speed,
 * steering angle}; var = their variances (NULL: no noise); geom = {h, l, dx, dy} (setAckermanParams).  The normal deviates come
 * from Philox4x32-10 keyed by `seed` with counter (particle, call): the reference's single serial boost stream has no parallel
 * form, the distribution is what is kept (csrc/motion.h).  Stream-ordered, no host wait.  OPTIONAL: a host that keeps the
 * reference's own ParticleFilter::propagate sends poses with rfsgpu_set_poses / rfsgpu_set_step_inputs_async instead; the
 * Victoria Park driver of this repository uses it because its host loop, not a kernel, was what bounded a run. */
int rfsgpu_propagate_ackerman_async(rfsgpu_filter *f, const double *u, const double *var, double dt, const double *geom, unsigned long long seed,
                                    unsigned long long call);
/* A run of n such propagations (consecutive odometry messages: u [n][2], var [n][2] or NULL, dt [n], calls numbered call0,
 * call0 + 1, ...) in ONE launch; every particle takes the steps one after the other: the same poses, bit for bit. */
int rfsgpu_propagate_ackerman_run_async(rfsgpu_filter *f, int n, const double *u, const double *var, const double *dt, const double *geom,
                                        unsigned long long seed, unsigned long long call0);
#ifdef RFSGPU_ENABLE_BENCH_API   /* [bench] / [test]: exported by the library, declared only for callers that ask (bench.py, the tests) */
/* [bench] Average duration (ns) of each hot-path kernel group over the async steps harvested since the last call / reset:
 * [0]=phd_update_map [1]=phd_weight_multifeature [2]=gm_merge(+prune); *n_steps = steps averaged.  Steps that ran as one
 * fused kernel report its duration in [0] and 0 in [1], [2] (their TimingInfo share is booked under mapUpdate). */
int rfsgpu_kernel_time_stats(rfsgpu_filter *f, double *avg_ns3, int *n_steps);
/* [bench] Average duration (ns) of the step's post kernel (Murty-200 partitions when any were queued, queue reset, weight sums /
 * division) over the fused steps the last rfsgpu_kernel_time_stats call covered. */
double rfsgpu_post_kernel_avg_ns(const rfsgpu_filter *f);
/* [bench] The HIP events behind the two calls above (and behind TimingInfo's device buckets) ride on every `every`-th fused step only
 * (default 8 since round 5; a filter's first step always carries them).  Three event records per step cost a step of configs[1] 8 us
 * of its 144 (each is a marker packet the queue drains before the next kernel starts).  Statistics average over the sampled steps;
 * TimingInfo (rfsgpu_get_timing) books a sampled step once for itself and once for every un-sampled step since the previous
 * sample, i.e. it stays an estimate of the whole run's device time.  `every` = 1 restores per-step events. */
int rfsgpu_set_step_timing_stride(rfsgpu_filter *f, int every);
/* [bench] Launch order of the fused step's particles, for launches with more workgroups than the GPU holds at once (the Victoria Park
 * step: one wavefront per particle; the 2-D step when its instantiation runs without phase priorities): such a launch lasts as long as
 * whatever started last, so the step kernel records every particle's duration and the post kernel sorts them for the next step, longest
 * first (mode 2, the default; RFSGPU_VP_COST_ORDER=0 in the environment: mode 0).  cost_out (N floats, may be NULL) receives every
 * particle's duration in the last step (100 MHz ticks; 0 where the launch did not capture them); mode 1 + order_in (a permutation of
 * 0..N-1: launch slot -> particle) freezes that order for the following steps, mode 0 goes back to slot == particle.  Results do not
 * depend on the order. */
int rfsgpu_step_launch_order(rfsgpu_filter *f, int mode, const int *order_in, float *cost_out);
#endif /* RFSGPU_ENABLE_BENCH_API */
/* The same four phases one at a time (used by the parity tests and by profiling):          */
/* [test] Murty-200 partition sums of n_jobs given extended tables (n_k x n_k row-major, back to back, n_k = nR[k] + nC[k] <= 64) by the
 * step's own post kernel: sums_out[k] = the sum of exp(score) over the <= 200 best assignments Murty returns with
 * setRealAssignmentBlock(nR, nC) (include/RBPHDFilter.hpp:942-959, src/MurtyAlgorithm.cpp:141-336).  The handle's weights are
 * left as they were.  (tests/test_gpu_parity.py pins it to the reference's BruteForceLinearAssignment fixture.) */
int rfsgpu_murty_partition_sums(rfsgpu_filter *f, const double *mats, const int *nR, const int *nC, int n_jobs, double *sums_out);
int rfsgpu_update_map(rfsgpu_filter *f, const double *z, int n_z);      /* updateMap       :543-725 */
int rfsgpu_importance_weighting(rfsgpu_filter *f);                       /* importanceWeighting :728-997 */
int rfsgpu_merge(rfsgpu_filter *f);                                      /* GaussianMixture::merge  GaussianMixture.hpp:394-475 */
int rfsgpu_prune(rfsgpu_filter *f);                                      /* GaussianMixture::prune  :477-534 */

/* unused_measurements_[slot] (:709-720) as indices in ascending order; returns count via *n_out. */
int rfsgpu_get_unused(rfsgpu_filter *f, int slot, int *idx, int max_n, int *n_out);
/* nLandmarksInFOV_[slot] (:601-613). */
int rfsgpu_landmarks_in_fov(rfsgpu_filter *f, int slot, int *n_out);

/* ---- [core] resampling, [multi] weight normalisation (ParticleFilter.hpp:352-363, 399-492) ---------------- */

/* Device reduction of this shard's {sum w, sum w^2}; out[2] on the host. */
int rfsgpu_weight_sums(rfsgpu_filter *f, double *out);
/* Device pointer to the same two doubles (valid after rfsgpu_weight_sums_async) so a multi-GPU
 * host can all-reduce them in place over RCCL without a host round trip. */
int rfsgpu_weight_sums_async(rfsgpu_filter *f);
void *rfsgpu_weight_sums_device_ptr(rfsgpu_filter *f);
/* w_i /= sum (sum = global sum over all shards).  If sum_dev != NULL the divisor is read on the
 * device from sum_dev[0] (after an in-place all-reduce) and `sum` is ignored. */
int rfsgpu_normalize_weights(rfsgpu_filter *f, double sum, const void *sum_dev);
/* The same when several shards (handles on one GPU and / or other GPUs) share the normalisation: sum_dev points to
 * n_parts consecutive {sum w, sum w^2} pairs (each shard's rfsgpu_bind_weight_sums_buffer slot, all-reduced in place
 * across GPUs); the divisor is the sum of their first elements, added in index order on the device. */
int rfsgpu_normalize_weights_parts(rfsgpu_filter *f, double sum, const void *sum_dev, int n_parts);
/* Apply a resampling decision (the copy loop of ParticleFilter::resample, include/ParticleFilter.hpp:446-479): slot k takes what
 * Particle::copy (include/Particle.hpp:218-223) carries from slot src_slot[k] -- the pose and a deep copy of the mixture -- and its
 * id; src_slot[k] == k keeps the slot (case 1, :466-467).  All weights are reset to 1 (:486-489).  Sources must be slots that
 * keep themselves.  The engine keeps the particles' id_ / idParent_ per slot exactly as that loop leaves them (a copy has its
 * SOURCE's id; idParent_ = the source's id; a survivor's idParent_ = its own id, which differs from its slot once it has been
 * a copy itself) and remembers that a resampling occurred (RBPHDFilter::resampleOccured_, cleared by the next update with
 * measurements, include/RBPHDFilter.hpp:526).  What happens to the three per-SLOT arrays of RBPHDFilter -- unused_measurements_,
 * birthGaussians_, nLandmarksInFOV_ -- is selected by rfsgpu_set_birth_inheritance (below).
 * Stream-ordered since round 6: src_slot is copied before the call returns, the device copy is enqueued behind whatever is on the handle's
 * stream and nothing waits for it (every reader of maps / weights does, on that stream); TimingInfo's particleResample buckets hold the
 * host time of the call. */
int rfsgpu_resample_apply(rfsgpu_filter *f, const int *src_slot);
/* Birth-state inheritance after a resampling (include/RBPHDFilter.hpp:1005-1011).
 *   RFSGPU_INHERIT_REFERENCE (default)  the reference, statement for statement: nothing but pose + mixture moves at resampling
 *       time; every rfsgpu_predict_map(add_birth = 1) while resampleOccured_ is set walks the slots in ascending order and, where
 *       idParent_ != slot, first copies unused_measurements_ and birthGaussians_ from SLOT idParent_ in the state that slot is
 *       in at that moment (a lower slot has already been through this predict's birth step, so its unused list is empty and
 *       its candidates are one check older; a higher slot has not), then runs the slot's own birth step.  nLandmarksInFOV_ is
 *       never copied.  On the device: a snapshot-ordered gather for the children of higher slots, then the birth step level by
 *       level along the chains of lower-slot parents (csrc/birth.h).
 *   RFSGPU_INHERIT_EAGER      rounds 1-2 of this engine ("the evident intent"): a child takes its parent's unused list, FOV count
 *       and candidate list at resampling time, predict copies nothing.  NOT the reference's results in the step after a
 *       resampling.  A handle on which rfsgpu_fastslam_update has run behaves this way whatever the mode: rfs::FastSLAM copies its
       landmark-candidate lists right at resampling time (FastSLAM::resampleWithMapCopy, include/FastSLAM.hpp:747-753).
 *   RFSGPU_INHERIT_EXTERNAL   resampling moves pose + mixture only and predict copies nothing: the host owns the rule (the
 *       multi-GPU hosts, whose parent slot may live on another shard: rfsgpu_get/set_unused_masks).
 * The environment variable RFSGPU_BIRTH_INHERITANCE=eager makes RFSGPU_INHERIT_EAGER the initial mode of new handles (A/B runs
 * of unmodified hosts). */
#define RFSGPU_INHERIT_REFERENCE 0
#define RFSGPU_INHERIT_EAGER 1
#define RFSGPU_INHERIT_EXTERNAL 2
int rfsgpu_set_birth_inheritance(rfsgpu_filter *f, int mode);
int rfsgpu_get_birth_inheritance(const rfsgpu_filter *f);
/* Particle::getId / getParentId of the particle in every slot (either pointer may be NULL); set: for a host that keeps the ids
 * itself (the sharded hosts: ids are GLOBAL slot numbers there). */
int rfsgpu_get_particle_ids(rfsgpu_filter *f, int *id, int *parent_id);
int rfsgpu_set_particle_ids(rfsgpu_filter *f, const int *id, const int *parent_id);
/* RBPHDFilter::resampleOccured_ as the engine tracks it (1 / 0). */
int rfsgpu_resample_occured(const rfsgpu_filter *f);
/* unused_measurements_ of all slots at once, one 64-bit mask per slot (bit z = measurement z of the last update is unused). */
/* [multi] One level of the level-ordered birth step, for a host that carries out the reference's slot-ordered copy of the per-slot
 * birth lists itself because parent slots live on other shards (ids are global there): the birth step runs for the slots whose
 * level_of_slot[] == level; the static step of every Gaussian in the call with do_static != 0 (births of later calls get their
 * + Q where they are created).  Between the calls the host moves lists with rfsgpu_get/set_unused_masks and
 * rfsgpu_export/import_birth_candidates (rfs-slam_amd/sharded.py, rfsgpu_group_predict_map). */
int rfsgpu_predict_map_level(rfsgpu_filter *f, int add_birth, const int *level_of_slot, int level, int do_static);
int rfsgpu_get_unused_masks(rfsgpu_filter *f, unsigned long long *masks);   /* a read: does NOT acknowledge the rule to an EXTERNAL-mode handle */
int rfsgpu_set_unused_masks(rfsgpu_filter *f, const unsigned long long *masks);
/* [multi] 1 if this handle has ever held birth-candidate lists (birthGaussians_, include/RBPHDFilter.hpp:1014-1083: a configuration
 * that keeps them has run a birth predict, or lists were imported with rfsgpu_import_birth_candidates), else 0; -1 for a null
 * handle.  The multi-GPU hosts take the closed form of the inheritance rule over the unused masks only while this is 0 on
 * EVERY shard (rfsgpu_group_predict_map and rfs-slam_amd/sharded.py use this one predicate). */
int rfsgpu_has_birth_candidates(const rfsgpu_filter *f);
/* ParticleFilter::resample(n) with n < nParticles_ (:417-483; FastSLAM::resampleWithMapCopy): the first n_out slots
 * receive src_slot[0..n_out), the particle count becomes n_out.  A source below n_out must keep itself; sources at or
 * beyond n_out are dropped after the copy. */
int rfsgpu_resample_apply_n(rfsgpu_filter *f, const int *src_slot, int n_out);

/* ---- [multi] cross-shard migration for GLOBAL resampling over several GPUs ---------------------------------------
 * The reference resamples over the whole particle set (ParticleFilter::resample, include/ParticleFilter.hpp:399-492) and a
 * child is a deep copy of its parent (Particle::copy, include/Particle.hpp:218-223: pose + mixture).  When parent and child
 * live on different GPUs the parent travels as one packed ROW of rfsgpu_slab_row_bytes() bytes -- pose (+ covariance) and
 * mixture; with RFSGPU_INHERIT_EAGER (and on FastSLAM handles) also the unused-measurement list, FOV count and birth candidates --
 * between DEVICE buffers: export on the source handle, transport by the caller (RCCL send/recv between processes,
 * hipMemcpyPeerAsync inside one process: rfsgpu_group_*), import on the destination handle.  Both calls are stream-ordered
 * on the handle's stream and never synchronise the host; `slots` is a host array that is free again on return.
 * Rows are only meaningful between handles of the same model and gm_capacity. */
size_t rfsgpu_slab_row_bytes(const rfsgpu_filter *f);
int rfsgpu_export_slab_rows(rfsgpu_filter *f, const int *slots, int n, void *dev_rows);        /* rows[k] <- particle slots[k] */
int rfsgpu_import_slab_rows(rfsgpu_filter *f, const int *slots, int n, const void *dev_rows);  /* particle slots[k] <- rows[k] */
/* Device pointer of the N particle weights (for an all-gather that stays on the GPUs); valid for the handle's lifetime. */
void *rfsgpu_weights_device_ptr(rfsgpu_filter *f);

/* ---- [multi] one filter over several GPUs from ONE host thread (SURVEY 8(b) "device_ids[], n_dev"; 8(e)) ------------------------
 * The particle set of rfs::RBPHDFilter is cut into contiguous blocks, one shard (an rfsgpu_filter) per listed device.  What a
 * C++ host standing where rfs::RBPHDFilter stands needs to use more than one GPU:
 *   predict / update / normalise / resample over the whole set, configuration broadcast to all shards, map access by global
 *   particle index.  Shards run their fused steps concurrently; they meet in the weight normalisation (an RCCL all-reduce of
 *   {sum w, sum w^2} on the shards' streams -- single-process communicators from ncclCommInitAll, librccl loaded with dlopen -- or,
 *   with repeated device ids, per-shard sums added on the host in shard order) and in resampling, which stays GLOBAL (ParticleFilter::resample, include/ParticleFilter.hpp:399-492):
 *   cross-device children travel as packed rows with hipMemcpyPeerAsync (rfsgpu_export/import_slab_rows).
 * device_ids may repeat a device (several shards on one GPU: how the single-GPU tests drive this code). */
typedef struct rfsgpu_group rfsgpu_group;
int rfsgpu_group_create(rfsgpu_group **out, int model, int n_particles, const int *device_ids, int n_dev, int gm_capacity);
void rfsgpu_group_destroy(rfsgpu_group *g);
const char *rfsgpu_group_last_error(const rfsgpu_group *g);
int rfsgpu_group_n_shards(const rfsgpu_group *g);
int rfsgpu_group_n_particles(const rfsgpu_group *g);
rfsgpu_filter *rfsgpu_group_shard(rfsgpu_group *g, int k);                    /* the shard's own handle (map import/export, timing) */
int rfsgpu_group_locate(const rfsgpu_group *g, int particle, int *shard, int *slot);
int rfsgpu_group_set_filter_config(rfsgpu_group *g, const rfsgpu_filter_config *c);
int rfsgpu_group_set_model_rngbrg(rfsgpu_group *g, const rfsgpu_rngbrg_config *c);
int rfsgpu_group_set_kf_config(rfsgpu_group *g, const rfsgpu_kf_config *c);
int rfsgpu_group_set_lmk_process_noise(rfsgpu_group *g, const double *Q);
int rfsgpu_group_set_model_victoriapark(rfsgpu_group *g, const rfsgpu_vp_config *c);   /* configs[3] sharded: the 3-D model on every shard */
int rfsgpu_group_set_laser_scan(rfsgpu_group *g, const double *scan, int n);
int rfsgpu_group_set_phase_timing(rfsgpu_group *g, int on);
/* RBPHDFilter::TimingInfo of the group: *_wall = the largest of the shards' (they run side by side), *_cpu = their sum. */
int rfsgpu_group_get_timing(rfsgpu_group *g, rfsgpu_timing *t);
/* How the shards' weight sums meet: "rccl" (all-reduce over RCCL / xGMI on the shards' streams, totals stay on the devices) or
 * "host: <why>" (repeated device ids, RFSGPU_GROUP_RCCL=0, librccl not loadable ...: pairs added on the host in shard order). */
const char *rfsgpu_group_collective(const rfsgpu_group *g);
int rfsgpu_group_set_poses(rfsgpu_group *g, const double *x, const double *cov, int cov_stride);   /* N poses, as rfsgpu_set_poses */
int rfsgpu_group_get_poses(rfsgpu_group *g, double *x);
int rfsgpu_group_set_weights(rfsgpu_group *g, const double *w);
int rfsgpu_group_get_weights(rfsgpu_group *g, double *w);
/* RBPHDFilter::predict, map part (:415-442), incl. the reference's birth-state inheritance after a resampling over GLOBAL slots:
 * a gather over the unused masks for immediate-birth configurations (birthGaussianMeasurementCountThreshold == 1), the
 * level-ordered walk with the lists moved between shards for configurations that keep candidate lists. */
int rfsgpu_group_predict_map(rfsgpu_group *g, int add_birth);
int rfsgpu_group_set_birth_inheritance(rfsgpu_group *g, int mode);          /* RFSGPU_INHERIT_REFERENCE (default) | RFSGPU_INHERIT_EAGER */
int rfsgpu_group_get_particle_ids(rfsgpu_group *g, int *id, int *parent_id); /* Particle::getId / getParentId by global slot */
/* RBPHDFilter::update body (:444-523) on every shard; weights stay un-normalised; sums_out (may be null) = {sum w, sum w^2}. */
int rfsgpu_group_update(rfsgpu_group *g, const double *z, int n_z, double *sums_out);
/* rfsgpu_group_update whose weight normalisation trails by one step (RCCL path: the all-reduce of {sum w, sum w^2} on a side stream
 * per shard beside the next step's kernel; each shard's post kernel divides by the previous call's total, rfsgpu_step_async_deferred).
 * Nothing waits for the GPUs; the next rfsgpu_group_* call that reads or replaces the weights applies the pending total first.
 * For steps after which the host does not need N_eff (the resample test is not due). */
int rfsgpu_group_update_deferred(rfsgpu_group *g, const double *z, int n_z);
/* The group form of rfsgpu_update_io: global poses (+ covariances, NULL = unchanged) and weights (NULL = unchanged) in, the update on every
 * shard (all chains enqueued before the first wait), the updated un-normalised weights of all particles out (NULL = not wanted);
 * device-side errors of any shard are reported by this call. */
int rfsgpu_group_update_io(rfsgpu_group *g, const double *x, const double *x_cov, int cov_stride, const double *w_in, const double *z, int n_z, double *w_out);
int rfsgpu_group_normalize(rfsgpu_group *g, double *sums_out);               /* normalizeWeights over all N (:352-363) */
/* ParticleFilter::resample (:399-492) with the caller's uniform draw (the reference's one drand48()); *fired tells whether the
 * N_eff test let it happen; plan_out (may be null, N ints): global source slot of every slot, for per-particle host data. */
int rfsgpu_group_resample(rfsgpu_group *g, double eff_n_threshold, double u01, int *fired, int *plan_out);
int rfsgpu_group_apply_plan(rfsgpu_group *g, const int *src);                /* a resampling plan computed elsewhere */
int rfsgpu_group_migration_stats(const rfsgpu_group *g, long long *rows, long long *bytes);
int rfsgpu_group_gm_size(rfsgpu_group *g, int particle);                     /* getGMSize, global index */
int rfsgpu_group_get_landmark(rfsgpu_group *g, int particle, int m, double *mean, double *cov, double *w);
int rfsgpu_group_synchronize(rfsgpu_group *g);

/* ---- [core] timing, [bench] / misc ---------------------------------------------------------------------------- */

/* getTimingInfo (:1219-1232).  The device buckets (mapUpdate_wall, ...) come from HIP events, and on the fused one-launch step those
 * events ride on every 8th step only (three marker packets per step cost a configs[1] step 8 us): a SAMPLED step is booked once for
 * itself and once for every un-sampled step since the previous sample, and un-sampled steps behind the last sample are booked at
 * that sample's duration at the next synchronising call -- so the bucket covers every step, as an estimate, not a per-step
 * measurement like the reference's timer_mapUpdate_.  rfsgpu_set_step_timing_stride(f, 1) (bench API) or rfsgpu_set_phase_timing
 * (per-phase launches with their own event pairs) give measured per-step values. */
int rfsgpu_get_timing(rfsgpu_filter *f, rfsgpu_timing *t);
int rfsgpu_reset_timing(rfsgpu_filter *f);
/* Block until all queued device work of this handle is complete; reports a pending device-side error of async steps. */
int rfsgpu_synchronize(rfsgpu_filter *f);
/* Native HIP stream of this handle (hipStream_t as void*), for callers that enqueue around it. */
void *rfsgpu_stream(rfsgpu_filter *f);
/* Run this handle's kernels on a caller-owned stream (e.g. the host framework's current stream) so
 * that collectives / events of the caller order against the engine without host round trips.
 * NULL restores the engine's own stream. */
int rfsgpu_set_stream(rfsgpu_filter *f, void *hip_stream);
/* Let rfsgpu_weight_sums_async write {sum w, sum w^2} into a caller-owned device buffer (2 doubles),
 * e.g. a tensor the multi-GPU host all-reduces in place over RCCL.  NULL restores the internal one. */
int rfsgpu_bind_weight_sums_buffer(rfsgpu_filter *f, void *dev_ptr);
#ifdef RFSGPU_ENABLE_BENCH_API   /* [bench] / [test]: exported by the library, declared only for callers that ask (bench.py, the tests) */
/* Device-side snapshot / restore of the MAP state of every particle (mixtures, sizes, particle weights, unused-measurement
 * lists, FOV counts): one snapshot slot per handle, allocated on first use.  A benchmarking / testing helper -- it is how a
 * timed step is re-seeded -- NOT a checkpoint: poses, the birth-candidate lists and the staged measurement set are not part of
 * it (callers that use candidate lists re-import them).  The reference has no equivalent (it has no checkpointing at all). */
int rfsgpu_save_state(rfsgpu_filter *f);
int rfsgpu_restore_state(rfsgpu_filter *f);
/* A ring of n_slots pre-seeded copies of the saved state, so that the input of every timed step is resident in HBM BEFORE the timed
 * region instead of being copied inside it (restore_state is a 45 MB copy per step at configs[1]).  _create allocates and fills the
 * slots from the snapshot (rfsgpu_save_state first; 0 frees the ring); _next makes the next slot the handle's current state by
 * swapping device pointers -- host work only, nothing is launched; a slot is consumed by the step that runs on it, _next past the
 * last slot is an error, _seed fills all slots again.  A swap moves the handle's weight array: a pointer obtained from
 * rfsgpu_weights_device_ptr before it is stale afterwards.  Results are those of rfsgpu_restore_state + the same step
 * (tests/test_gpu_parity.py::test_state_ring_steps_equal_restore_and_step). */
int rfsgpu_state_ring_create(rfsgpu_filter *f, int n_slots);
int rfsgpu_state_ring_seed(rfsgpu_filter *f);
int rfsgpu_state_ring_next(rfsgpu_filter *f);
/* Duration in ns of the most recent launch of each hot-path kernel, from HIP events on the
 * engine's stream: [0]=phd_update_map [1]=phd_weight_multifeature [2]=gm_merge [3]=gm_prune. */
int rfsgpu_last_kernel_ns(rfsgpu_filter *f, long long *ns4);
/* [test] Which instantiation the last stream-ordered step (rfsgpu_update / _update_async / _step_async) launched: out4 = {waves per
 * particle of the fused step kernel, phase priorities on (1 / 0), log2 of the merge grid's side (5 | 6), fused (1) or three
 * kernels (0)} -- so that the full-size parity tests can say which kernel they checked (phd_step_fused_kernel<2, true, 5> is the
 * one bench.py times at configs[1]). */
int rfsgpu_last_step_variant(const rfsgpu_filter *f, int *out4);
#endif /* RFSGPU_ENABLE_BENCH_API */

/* MatPerm::calc (src/MatrixPermanent.cpp:41-112), batched: `batch` row-major n x n matrices in A
 * (host), permanents to out (host).  n <= 24.  Standalone (no filter handle needed). */
int rfsgpu_mat_perm(const double *A, int n, int batch, double *out, int device_id);
#ifdef RFSGPU_ENABLE_BENCH_API
/* [bench] Device time (ms, HIP events) of the kernel of the last rfsgpu_mat_perm call of this process. */
double rfsgpu_mat_perm_last_kernel_ms(void);
#endif

/* ---- [fastslam] FastSLAM 1.0 on the same handle (reference include/FastSLAM.hpp) --------------------------------
 * The handle's mixtures double as FastSLAM's per-particle landmark maps: a Gaussian's weight is the landmark's
 * log-odds of existence (FastSLAM.hpp:598-617), the birth-candidate lists are the landmark candidates
 * (landmarkCandidates_, :84-88).  Both measurement models.  The map part of FastSLAM::predict (:376-383, staticStep on
 * every landmark) is rfsgpu_predict_map(f, 0). */
void rfsgpu_default_fastslam_config(rfsgpu_fastslam_config *cfg);                       /* constructor defaults :243-257 */
int rfsgpu_set_fastslam_config(rfsgpu_filter *f, const rfsgpu_fastslam_config *cfg);    /* public member `config`        */
int rfsgpu_get_fastslam_config(const rfsgpu_filter *f, rfsgpu_fastslam_config *cfg);
/* FastSLAM::updateMap for every particle (:387-418, 424-706): in-range landmarks, log-likelihood table, CostMatrix::reduce +
 * best data association, Kalman correction of the associated landmarks, existence log-odds, pruning, new landmarks /
 * candidates from the unassociated measurements, particle weight *= exp(sum of the associated log-likelihoods).
 * n_z == 0 returns OK without touching anything (:401-402).  resampleWithMapCopy (:708-735) stays with the caller
 * (rfsgpu_weight_sums / rfsgpu_normalize_weights / rfsgpu_resample_apply[_n]).
 * Multi-hypothesis FastSLAM (config.maxNDataAssocHypotheses in 2..16): every particle keeps Murty's k best associations
 * within maxDataAssocLogLikelihoodDiff of the best and is copied once per extra hypothesis (:506-556,
 * ParticleFilter::copyParticle): the particle set GROWS (rfsgpu_n_particles; room from rfsgpu_create_ex, else
 * RFSGPU_ERR_CAPACITY).  The copies are appended in particle order; rfsgpu_particle_parents tells the host which slot each
 * particle was copied from (itself for the originals) so that it can duplicate its poses.  Limit of this path: landmarks in
 * range and measurements <= 64 each (Murty runs on the dense reduced table like the reference). */
int rfsgpu_fastslam_update(rfsgpu_filter *f, const double *z, int n_z);
/* FastSLAM::resampleOccured_ (whether the previous update ended in a resampling): decides if the copies of a multiplied
 * particle inherit its landmark candidates (:551-553).  The host mirror sets it after its resampleWithMapCopy. */
int rfsgpu_fastslam_set_resample_occured(rfsgpu_filter *f, int flag);
/* parent[k] = the slot particle k was copied from by the last rfsgpu_fastslam_update (k itself when it was not a copy). */
int rfsgpu_particle_parents(rfsgpu_filter *f, int *parent, int max_n);

#ifdef __cplusplus
}
#endif
#endif /* RFSGPU_H */
