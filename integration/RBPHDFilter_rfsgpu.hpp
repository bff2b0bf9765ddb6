/*
 * RBPHDFilter_rfsgpu.hpp -- the reference-side binding of librfsgpu.so, written out in full.
 *
 * WHAT THIS IS.  A replacement for the reference's `include/RBPHDFilter.hpp`: install it AS `RBPHDFilter.hpp` in a
 * directory that precedes `/root/reference/include` on the include path (or `#include` it from a patched copy of that
 * header).  It declares `rfs::RBPHDFilter<RobotProcessModel, LmkProcessModel, MeasurementModel, KalmanFilter>` with the
 * same template signature, base class, public members and public `config` / `timingInfo_` objects as the reference
 * (include/RBPHDFilter.hpp:72-251), so that `src/rbphdslam2dSim.cpp` (:446-491 setup, :588 predict, :592 setParticlePose,
 * :607 update, :613-620 particle log, :628-633 getGMSize/getLandmark, :656-690 timing) and
 * `src/rbphdslam_VictoriaPark.cpp` (:344, :365-398, :513-543, :555-583, :637-657) compile against it unchanged.  The map of
 * every particle lives on the GPU behind the C ABI (include/rfsgpu.h); poses, weights, the process model, the random
 * numbers and the resampling DECISION stay on the host exactly where the reference has them.
 *
 * STATUS.  Compiled and linked by tests/test_reference_binding.py: the UNMODIFIED `src/rbphdslam2dSim.cpp` and
 * `src/rbphdslam_VictoriaPark.cpp` (+ the reference's own model / process / timer sources) against this file installed as
 * `RBPHDFilter.hpp` (integration/include/) and librfsgpu.so.  The build image has neither Eigen3 nor Boost, so that test
 * compiles over small stand-in headers (tests/support/stubs/, test support only); with the real libraries nothing else
 * changes.  The same logic, with plain-array stand-ins for Pose2d / Landmark2d / Measurement2d, is what
 * `rfs-slam_amd/host/rbphd_filter.hpp` compiles and what the GPU tests run.
 *
 * Supported template tuples (anything else is refused at compile time -- there is no CPU fallback in the product):
 *   <MotionModel_Odometry2d, StaticProcessModel<Landmark2d>, MeasurementModel_RngBrg,       KalmanFilter_RngBrg>
 *   <MotionModel_Ackerman2d, StaticProcessModel<Landmark3d>, MeasurementModel_VictoriaPark, KalmanFilter_VictoriaPark>
 * Victoria Park: the filter needs two values the driver hands to the measurement model and the model keeps private
 * (src/rbphdslam_VictoriaPark.cpp:371 `setNoise(R, Slb)`, :582 `setLaserScan(scan)`;
 * include/MeasurementModel_VictoriaPark.hpp:149,151).  No reference file is touched for that: for this tuple
 * `getMeasurementModel()` returns a handle (`rfsgpu_vp_model_handle`, below) that forwards every public member of the
 * model -- `config` included -- and keeps a copy of those two arguments on the way through.
 */
#ifndef RBPHDFILTER_HPP   /* the reference's own include guard: this file stands in for that header */
#define RBPHDFILTER_HPP

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <stdexcept>
#include <string>
#include <vector>

#include <boost/timer/timer.hpp>
#include <Eigen/Core>

#include "GaussianMixture.hpp"
#include "KalmanFilter_RngBrg.hpp"
#include "KalmanFilter_VictoriaPark.hpp"
#include "MeasurementModel_RngBrg.hpp"
#include "MeasurementModel_VictoriaPark.hpp"
#include "ParticleFilter.hpp"
#include "ProcessModel_Ackerman2D.hpp"
#include "ProcessModel_Odometry2D.hpp"
#include "Timer.hpp"

#include "rfsgpu.h"

namespace rfs {

/* ---- which engine model a template tuple maps to -------------------------------------------------------------------- */
template <class MeasurementModel> struct rfsgpu_model_of;  /* undefined: unsupported tuples do not compile */
template <> struct rfsgpu_model_of<MeasurementModel_RngBrg> { enum { value = RFSGPU_MODEL_RNGBRG_2D, dim = 2 }; };
template <> struct rfsgpu_model_of<MeasurementModel_VictoriaPark> { enum { value = RFSGPU_MODEL_VICTORIAPARK_3D, dim = 3 }; };

/* ---- what getMeasurementModel() hands to the driver ------------------------------------------------------------------ */
/* MeasurementModel_VictoriaPark keeps Slb_ and laserscan_ private with no getters (include/MeasurementModel_VictoriaPark.hpp
 * :149,151) and the device needs both.  The handle forwards the model's whole public interface (:29-134) and records them. */
class rfsgpu_vp_model_handle {
 public:
  explicit rfsgpu_vp_model_handle(MeasurementModel_VictoriaPark *m) : config(m->config), m_(m), Slb_(0), haveSlb_(false), scanSerial_(0) {}
  MeasurementModel_VictoriaPark::Config &config;                                     /* `->config.x = ...` (:373-379 of the driver) */
  void setNoise(Measurement3d::Mat &R, double Slb) { m_->setNoise(R, Slb); Slb_ = Slb; haveSlb_ = true; }
  void setNoise(Measurement3d::Mat &R) { m_->MeasurementModel<Pose2d, Landmark3d, Measurement3d>::setNoise(R); }
  void getNoise(Measurement3d::Mat &R) { m_->getNoise(R); }
  void setLaserScan(const std::vector<double> &scan) { m_->setLaserScan(scan); scan_ = scan; scanSerial_++; }
  bool measure(const Pose2d &pose, const Landmark3d &lm, Measurement3d &z, Eigen::Matrix3d *jl = NULL, Eigen::Matrix3d *jp = NULL) { return m_->measure(pose, lm, z, jl, jp); }
  void inverseMeasure(const Pose2d &pose, const Measurement3d &z, Landmark3d &lm) { m_->inverseMeasure(pose, z, lm); }
  double probabilityOfDetection(const Pose2d &pose, const Landmark3d &lm, bool &lim) { return m_->probabilityOfDetection(pose, lm, lim); }
  double probabilityOfDetection2(const Pose2d &pose, const Landmark3d &lm, bool &lim) { return m_->probabilityOfDetection2(pose, lm, lim); }
  double clutterIntensity(Measurement3d &z, int nZ) { return m_->clutterIntensity(z, nZ); }
  double clutterIntensityIntegral(int nZ = 0) { return m_->clutterIntensityIntegral(nZ); }
  operator MeasurementModel_VictoriaPark *() const { return m_; }
  MeasurementModel_VictoriaPark *model() const { return m_; }
  /* for the filter */
  bool haveSlb() const { return haveSlb_; }
  double Slb() const { return Slb_; }
  const std::vector<double> &laserScan() const { return scan_; }
  unsigned long laserScanSerial() const { return scanSerial_; }
 private:
  MeasurementModel_VictoriaPark *m_;
  double Slb_;
  bool haveSlb_;
  std::vector<double> scan_;
  unsigned long scanSerial_;
};
template <class MeasurementModel> struct rfsgpu_model_access {           /* 2-D model: everything the device needs is public */
  typedef MeasurementModel *pointer;
  struct holder {
    explicit holder(MeasurementModel *m) : m_(m) {}
    pointer get() { return m_; }
    MeasurementModel *m_;
  };
};
template <> struct rfsgpu_model_access<MeasurementModel_VictoriaPark> {
  typedef rfsgpu_vp_model_handle *pointer;
  struct holder {
    explicit holder(MeasurementModel_VictoriaPark *m) : h_(m) {}
    pointer get() { return &h_; }
    rfsgpu_vp_model_handle h_;
  };
};

/* ---- one GPU or several ------------------------------------------------------------------------------------------------- */
/* The reference constructor takes a particle count and nothing else (include/RBPHDFilter.hpp:350), so where the maps live comes
 * from the environment: RFSGPU_DEVICE=<ordinal> (default 0) -> ONE handle on that GPU; RFSGPU_DEVICES=0,1,2,... -> the particle set
 * cut into contiguous blocks over the listed GPUs (rfsgpu_group_*: the shards step side by side, the weight sums meet in an RCCL
 * all-reduce, resampling stays global with cross-device children moved over xGMI; a device may be listed more than once).
 * Every call the filter below makes goes through this facade, 1 : 1 onto rfsgpu_* or rfsgpu_group_*. */
class rfsgpu_engine_facade {
 public:
  rfsgpu_engine_facade() : f_(NULL), g_(NULL) {}
  ~rfsgpu_engine_facade() { if (g_) rfsgpu_group_destroy(g_); else rfsgpu_destroy(f_); }
  int create(int model, int n, int capacity) {
    const char *devs = std::getenv("RFSGPU_DEVICES"), *dev = std::getenv("RFSGPU_DEVICE");
    if (devs && *devs) {
      std::vector<int> ids;
      for (const char *p = devs; *p;) {
        char *e;
        const long v = std::strtol(p, &e, 10);
        if (e == p) break;
        ids.push_back((int)v);
        p = (*e == ',') ? e + 1 : e;
      }
      if (ids.empty()) return RFSGPU_ERR_INVALID;
      return rfsgpu_group_create(&g_, model, n, ids.data(), (int)ids.size(), capacity);
    }
    return rfsgpu_create(&f_, model, n, dev ? std::atoi(dev) : 0, capacity);
  }
  const char *last_error() const { return g_ ? rfsgpu_group_last_error(g_) : rfsgpu_last_error(f_); }
  int set_filter_config(const rfsgpu_filter_config *c) { return g_ ? rfsgpu_group_set_filter_config(g_, c) : rfsgpu_set_filter_config(f_, c); }
  int set_kf_config(const rfsgpu_kf_config *c) { return g_ ? rfsgpu_group_set_kf_config(g_, c) : rfsgpu_set_kf_config(f_, c); }
  int set_lmk_process_noise(const double *q) { return g_ ? rfsgpu_group_set_lmk_process_noise(g_, q) : rfsgpu_set_lmk_process_noise(f_, q); }
  int set_model_rngbrg(const rfsgpu_rngbrg_config *c) { return g_ ? rfsgpu_group_set_model_rngbrg(g_, c) : rfsgpu_set_model_rngbrg(f_, c); }
  int set_model_victoriapark(const rfsgpu_vp_config *c) { return g_ ? rfsgpu_group_set_model_victoriapark(g_, c) : rfsgpu_set_model_victoriapark(f_, c); }
  int set_laser_scan(const double *scan, int n) { return g_ ? rfsgpu_group_set_laser_scan(g_, scan, n) : rfsgpu_set_laser_scan(f_, scan, n); }
  int set_poses(const double *x, const double *cov, int stride) { return g_ ? rfsgpu_group_set_poses(g_, x, cov, stride) : rfsgpu_set_poses(f_, x, cov, stride); }
  int set_weights(const double *w) { return g_ ? rfsgpu_group_set_weights(g_, w) : rfsgpu_set_weights(f_, w); }
  int get_weights(double *w) { return g_ ? rfsgpu_group_get_weights(g_, w) : rfsgpu_get_weights(f_, w); }
  int predict_map(int add_birth) { return g_ ? rfsgpu_group_predict_map(g_, add_birth) : rfsgpu_predict_map(f_, add_birth); }
  int update(const double *z, int n_z) { return g_ ? rfsgpu_group_update(g_, z, n_z, NULL) : rfsgpu_update(f_, z, n_z); }
  /* RBPHDFilter::update's device part with its inputs and outputs in ONE call and one wait (poses + covariances + weights in, weights out) */
  int update_io(int predict, const double *x, const double *cov, int stride, const double *w_in, const double *z, int n_z, double *w_out) {
    return g_ ? rfsgpu_group_update_io(g_, x, cov, stride, w_in, z, n_z, w_out)     /* (a group never has a predict pending: single() below) */
              : rfsgpu_update_io(f_, predict, x, cov, stride, w_in, z, n_z, w_out);
  }
  bool single() const { return g_ == NULL; }      /* one handle: rfsgpu_update_io can take the pending predict's map part with it */
  int resample_apply(const int *src) { return g_ ? rfsgpu_group_apply_plan(g_, src) : rfsgpu_resample_apply(f_, src); }
  int gm_size(int i) { return g_ ? rfsgpu_group_gm_size(g_, i) : rfsgpu_gm_size(f_, i); }
  int get_landmark(int i, int m, double *mean, double *cov, double *w) { return g_ ? rfsgpu_group_get_landmark(g_, i, m, mean, cov, w) : rfsgpu_get_landmark(f_, i, m, mean, cov, w); }
  int get_timing(rfsgpu_timing *t) { return g_ ? rfsgpu_group_get_timing(g_, t) : rfsgpu_get_timing(f_, t); }
  int set_phase_timing(int on) { return g_ ? rfsgpu_group_set_phase_timing(g_, on) : rfsgpu_set_phase_timing(f_, on); }
 private:
  rfsgpu_filter *f_;
  rfsgpu_group *g_;
  rfsgpu_engine_facade(const rfsgpu_engine_facade &);
  rfsgpu_engine_facade &operator=(const rfsgpu_engine_facade &);
};

template <class RobotProcessModel, class LmkProcessModel, class MeasurementModel, class KalmanFilter>
class RBPHDFilter : public ParticleFilter<RobotProcessModel, MeasurementModel, GaussianMixture<typename MeasurementModel::TLandmark> > {
 public:
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW;

  typedef typename RobotProcessModel::TState TPose;
  typedef typename RobotProcessModel::TInput TInput;
  typedef typename MeasurementModel::TLandmark TLandmark;
  typedef typename MeasurementModel::TMeasurement TMeasurement;
  typedef GaussianMixture<TLandmark> TGM;
  typedef typename TGM::Gaussian TGaussian;

  /* include/RBPHDFilter.hpp:90-146, member for member */
  struct Config {
    double birthGaussianWeight_;
    uint birthGaussianMeasurementCountThreshold_;
    uint birthGaussianMeasurementCheckThreshold_;
    double birthGaussianMeasurementSupportDist_;
    uint birthGaussianCurrentMeasurementCountThreshold_;
    double newGaussianCreateInnovMDThreshold_;
    int importanceWeightingEvalPointCount_;
    double importanceWeightingEvalPointGuassianWeight_;
    double importanceWeightingMeasurementLikelihoodMDThreshold_;
    double gaussianMergingThreshold_;
    double gaussianMergingCovarianceInflationFactor_;
    double gaussianPruningThreshold_;
    int minUpdatesBeforeResample_;
    int minMeasurementsBeforeResample_;
    bool useClusterProcess_;
  } config;

  /* include/RBPHDFilter.hpp:152-167, member for member (nanoseconds, as boost::timer reports them) */
  struct TimingInfo {
    long long predict_wall;
    long long predict_cpu;
    long long mapUpdate_wall;
    long long mapUpdate_cpu;
    long long mapUpdate_kf_wall;
    long long mapUpdate_kf_cpu;
    long long particleWeighting_wall;
    long long particleWeighting_cpu;
    long long mapMerge_wall;
    long long mapMerge_cpu;
    long long mapPrune_wall;
    long long mapPrune_cpu;
    long long particleResample_wall;
    long long particleResample_cpu;
  } timingInfo_;

  /* :370-382 -- the constructor's defaults.  (The reference leaves importanceWeightingEvalPointGuassianWeight_ and
   * useClusterProcess_ uninitialised; both drivers assign them, :485,490.  They start at 0 / false here.) */
  explicit RBPHDFilter(int n)
      : ParticleFilter<RobotProcessModel, MeasurementModel, GaussianMixture<TLandmark> >(n), model_(this->pMeasurementModel_) {
    lmkModelPtr_ = new LmkProcessModel;
    kf_ = new KalmanFilter(lmkModelPtr_, this->pMeasurementModel_);
    /* :362-364.  Particle::copy (include/Particle.hpp:218-223) dereferences data_, so every particle owns a mixture object as in
     * the reference; it stays EMPTY here -- the particle's Gaussians live on the device (getGMSize / getLandmark read them). */
    for (int i = 0; i < n; i++) this->particleSet_[i]->setData(boost::shared_ptr<TGM>(new TGM()));
    config.birthGaussianWeight_ = 0.25;
    config.birthGaussianMeasurementCountThreshold_ = 1;
    config.birthGaussianMeasurementCheckThreshold_ = 1;
    config.birthGaussianMeasurementSupportDist_ = 1;
    config.birthGaussianCurrentMeasurementCountThreshold_ = 1;
    config.gaussianMergingThreshold_ = 0.5;
    config.gaussianMergingCovarianceInflationFactor_ = 1.5;
    config.gaussianPruningThreshold_ = 0.2;
    config.importanceWeightingEvalPointCount_ = 8;
    config.importanceWeightingEvalPointGuassianWeight_ = 0;
    config.importanceWeightingMeasurementLikelihoodMDThreshold_ = 3.0;
    config.newGaussianCreateInnovMDThreshold_ = 0.2;
    config.minUpdatesBeforeResample_ = 1;
    config.minMeasurementsBeforeResample_ = 1;
    config.useClusterProcess_ = false;
    nUpdatesSinceResample_ = 0;
    nMeasurementsSinceResample_ = 0;
    resampleOccured_ = false;
    /* The engine: device(s) and per-particle capacity come from the environment (the reference constructor has no such
     * arguments): RFSGPU_DEVICE (default 0) or RFSGPU_DEVICES=0,1,... (rfsgpu_engine_facade above), RFSGPU_GM_CAPACITY (default 512:
     * room for nM_max + 4 nZ Gaussians). */
    const char *cap = std::getenv("RFSGPU_GM_CAPACITY");
    const int rc = engine_.create(rfsgpu_model_of<MeasurementModel>::value, n, cap ? std::atoi(cap) : 512);
    if (rc != RFSGPU_OK) throw std::runtime_error("rfsgpu_create failed (no gfx950 device?): status " + std::to_string(rc));
    /* rfsgpu_update runs the 2-D step as one fused launch and books it under TimingInfo::mapUpdate_*; a driver whose timing
     * printout (src/rbphdslam2dSim.cpp:664-679) should keep mapUpdate / particleWeighting / mapMerge apart sets
     * RFSGPU_PHASE_TIMING=1 and gets separate launches -- the same results bit for bit, about a third more device time. */
    if (const char *pt = std::getenv("RFSGPU_PHASE_TIMING")) engine_.set_phase_timing(std::atoi(pt));
  }

  ~RBPHDFilter() {
    if (prof_ && profN_ > 0)
      std::fprintf(stderr, "rfsgpu binding predict() breakdown over %lld calls, us per call: flush + configuration %.2f | map part (lazy: pose check + record; eager: "
                           "pose push + predict launch + wait) %.2f | ParticleFilter::propagate (reference host code) %.2f | lazy=%d\n",
                   profN_, 1e-3 * profNs_[0] / profN_, 1e-3 * profNs_[1] / profN_, 1e-3 * profNs_[2] / profN_, lazyPredict_ ? 1 : 0);
    if (prof_ && profTailN_ > 0)
      std::fprintf(stderr, "rfsgpu binding resample tail of update(): %lld updates, %.2f us per update in all; rfsgpu_resample_apply fired %lld times, %.2f us per call\n",
                   profTailN_, 1e-3 * profTailNs_ / profTailN_, profResN_, profResN_ ? 1e-3 * profResNs_ / profResN_ : 0.0);
    delete kf_;
    delete lmkModelPtr_;
  }

  LmkProcessModel *getLmkProcessModel() { return lmkModelPtr_; }   /* :402-404 */
  KalmanFilter *getKalmanFilter() { return kf_; }                   /* :1189-1191 (one object: no per-thread copies to broadcast) */

  /* include/ParticleFilter.hpp:112 -- hidden here so that, for the Victoria Park tuple, `setNoise(R, Slb)` and `setLaserScan`
   * pass through the handle above (the 2-D tuple gets the model pointer itself, as in the reference) */
  typename rfsgpu_model_access<MeasurementModel>::pointer getMeasurementModel() { return model_.get(); }

  /* :415-442.  Births use the pose BEFORE propagation (:423-426).
   * Round 6: the map part of the predict (addBirthGaussians :1000-1084 + staticStep :433-439) is LAZY where that changes nothing: when
   * the device still holds exactly the poses the particles have now -- the previous update() pushed them and nothing has moved them
   * since, checked against a host copy -- the births can happen at those device poses at any later time, so predict() only records
   * (birthGaussianCheck) and the next update() hands it to rfsgpu_update_io, whose step kernel runs the births + static step at its
   * head: no pose push, no predict launch, no wait for it (the same Gaussians bit for bit,
   * test_fused_predict_update_cycle_is_the_call_by_call_cycle).  Anything that reads the maps in between (getGMSize, getLandmark), a
   * second predict() before the update, a configuration change between predict() and update(), the Victoria Park model and a filter
   * over several GPUs take the eager path -- the call-by-call sequence of rounds 3-5.  RFSGPU_LAZY_PREDICT=0 turns the lazy path off. */
  void predict(TInput u, TimeStamp const &dT, bool useModelNoise = true, bool useInputNoise = false, bool birthGaussianCheck = true) {
    timer_predict_.resume();
    const long long t0 = prof_now();
    flushPredict();                                          /* (a predict still pending from a step without an update) */
    pushConfiguration();
    const long long t1 = prof_now();
    if (lazyPredict_ && rfsgpu_model_of<MeasurementModel>::dim == 2 && engine_.single() && devicePosesAreCurrent()) {
      predictPending_ = birthGaussianCheck ? 1 : 0;
    } else {
      pushPoses();
      check(engine_.predict_map(birthGaussianCheck ? 1 : 0), "predict_map");
    }
    const long long t2 = prof_now();
    this->propagate(u, dT, useModelNoise, useInputNoise, true);                      /* :429, host RNG, keeps the trajectory */
    if (prof_) { const long long t3 = prof_now(); profNs_[0] += t1 - t0; profNs_[1] += t2 - t1; profNs_[2] += t3 - t2; profN_++; }
    timer_predict_.stop();
  }

  /* :444-541 */
  void update(std::vector<TMeasurement> &Z) {
    nUpdatesSinceResample_++;
    this->setMeasurements(Z);                              /* Z is consumed (include/ParticleFilter.hpp:316-320) */
    if (this->measurements_.size() == 0) return;           /* :451-452 */
    nMeasurementsSinceResample_ += this->measurements_.size();
    const int nZ = (int)this->measurements_.size();
    const int D = rfsgpu_model_of<MeasurementModel>::dim;
    std::vector<double> z((size_t)nZ * D);
    for (int k = 0; k < nZ; k++) {
      typename TMeasurement::Vec v = this->measurements_[k].get();
      for (int d = 0; d < D; d++) z[(size_t)D * k + d] = v[d];
    }
    pushConfiguration();                                   /* (flushes a pending predict first if the configuration has changed under it) */
    /* poses (+ covariances) and weights in, [the pending predict's births + static step,] updateMap + importanceWeighting + merge +
     * prune (:469-520), weights out: one engine call, one wait for the device */
    const int n = this->nParticles_;
    xHost_.resize((size_t)3 * n); PHost_.resize((size_t)9 * n); wHost_.resize((size_t)n);
    gatherPoses(xHost_, PHost_);
    for (int i = 0; i < n; i++) wHost_[i] = this->particleSet_[i]->getWeight();
    timer_mapUpdate_.resume();
    const int pred = predictPending_ >= 0 ? predictPending_ : RFSGPU_CYCLE_NO_PREDICT;
    predictPending_ = -1;
    xOnDeviceValid_ = false;
    check(engine_.update_io(pred, xHost_.data(), PHost_.data(), 9, wHost_.data(), z.data(), nZ, wHost_.data()), "update");
    xOnDeviceValid_ = true;                                 /* xHost_ is what the device holds now (a resampling copies both sides alike) */
    timer_mapUpdate_.stop();
    for (int i = 0; i < n; i++) this->particleSet_[i]->setWeight(wHost_[i]);

    timer_particleResample_.resume();                       /* :524-539, unchanged */
    const long long tt0 = prof_now();
    resampleOccured_ = false;
    if (nUpdatesSinceResample_ >= config.minUpdatesBeforeResample_ && nMeasurementsSinceResample_ >= config.minMeasurementsBeforeResample_) {
      resampleOccured_ = resampleWithDeviceMaps();
    }
    if (resampleOccured_) {
      nUpdatesSinceResample_ = 0;
      nMeasurementsSinceResample_ = 0;
    } else {
      this->normalizeWeights();
    }
    if (prof_) { profTailNs_ += prof_now() - tt0; profTailN_++; }
    timer_particleResample_.stop();
  }

  int getGMSize(int i) {                                     /* :1152-1158 */
    flushPredict();                                          /* (the maps are read: a recorded predict happens now) */
    if (i >= 0 && i < this->nParticles_) return engine_.gm_size(i);
    return -1;
  }

  bool getLandmark(const int i, const int m, typename TLandmark::Vec &u, typename TLandmark::Mat &S, double &w) {   /* :1160-1178 */
    const int sz = getGMSize(i);
    if (sz == -1 || m < 0 || m >= sz) return false;
    const int D = rfsgpu_model_of<MeasurementModel>::dim;
    double mean[3], cov[9];
    if (engine_.get_landmark(i, m, mean, cov, &w) != RFSGPU_OK) return false;
    for (int r = 0; r < D; r++) {
      u[r] = mean[r];
      for (int c = 0; c < D; c++) S(r, c) = cov[D * r + c];
    }
    return true;
  }

  void setParticlePose(int i, TPose &p) { *(this->particleSet_[i]) = p; }   /* :1180-1186 (seen by devicePosesAreCurrent() at the next predict) */

  TimingInfo *getTimingInfo() {                              /* :1219-1232: host timers for what stays on the host, the */
    rfsgpu_timing t;                                         /* engine's HIP-event buckets for what runs on the device  */
    engine_.get_timing(&t);
    timer_predict_.elapsed(timingInfo_.predict_wall, timingInfo_.predict_cpu);
    timingInfo_.mapUpdate_wall = t.mapUpdate_wall;                   timingInfo_.mapUpdate_cpu = t.mapUpdate_cpu;
    timingInfo_.mapUpdate_kf_wall = t.mapUpdate_kf_wall;             timingInfo_.mapUpdate_kf_cpu = t.mapUpdate_kf_cpu;
    timingInfo_.particleWeighting_wall = t.particleWeighting_wall;   timingInfo_.particleWeighting_cpu = t.particleWeighting_cpu;
    timingInfo_.mapMerge_wall = t.mapMerge_wall;                     timingInfo_.mapMerge_cpu = t.mapMerge_cpu;
    timingInfo_.mapPrune_wall = t.mapPrune_wall;                     timingInfo_.mapPrune_cpu = t.mapPrune_cpu;
    timer_particleResample_.elapsed(timingInfo_.particleResample_wall, timingInfo_.particleResample_cpu);
    return &timingInfo_;
  }

 private:
  rfsgpu_engine_facade engine_;
  LmkProcessModel *lmkModelPtr_;
  KalmanFilter *kf_;
  typename rfsgpu_model_access<MeasurementModel>::holder model_;
  unsigned long scanPushed_ = 0;
  int nUpdatesSinceResample_, nMeasurementsSinceResample_;
  bool resampleOccured_;
  Timer timer_predict_, timer_mapUpdate_, timer_particleResample_;
  /* lazy predict (round 6) */
  int predictPending_ = -1;                                 /* -1 none | 0 static step only | 1 births + static step: recorded, not yet run */
  bool lazyPredict_ = !(std::getenv("RFSGPU_LAZY_PREDICT") && std::atoi(std::getenv("RFSGPU_LAZY_PREDICT")) == 0);
  bool xOnDeviceValid_ = false;                             /* xHost_ holds the pose means the last update() pushed */
  std::vector<double> xHost_, PHost_, wHost_;
  std::vector<int> srcSlot_;
  struct PushedConfig { rfsgpu_filter_config c; rfsgpu_kf_config k; double q[9]; rfsgpu_rngbrg_config m; bool valid; } pushed_ = {};
  /* RFSGPU_BINDING_PROFILE=1: where predict()'s host time goes, printed by the destructor (profiles/r06*_binding_predict_breakdown.txt) */
  bool prof_ = std::getenv("RFSGPU_BINDING_PROFILE") && std::atoi(std::getenv("RFSGPU_BINDING_PROFILE")) != 0;
  long long profNs_[3] = {0, 0, 0}, profN_ = 0, profResNs_ = 0, profResN_ = 0, profTailNs_ = 0, profTailN_ = 0;
  long long prof_now() const {
    if (!prof_) return 0;
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (long long)ts.tv_sec * 1000000000LL + ts.tv_nsec;
  }

  /* A recorded predict happens NOW, at the poses the device holds (the pre-propagation ones: nothing has pushed poses since). */
  void flushPredict() {
    if (predictPending_ < 0) return;
    const int p = predictPending_;
    predictPending_ = -1;
    check(engine_.predict_map(p), "predict_map");
  }
  /* Are the device's poses the particles' poses?  True when the last update() pushed them (xHost_) and every particle's mean still
   * equals that copy -- a resampling permutes both sides alike (resampleWithDeviceMaps keeps xHost_ in step), setParticlePose and
   * writes through getParticleSet() show up as a difference.  Only the means matter: inverseMeasure (births) reads no covariance. */
  bool devicePosesAreCurrent() {
    if (!xOnDeviceValid_) return false;
    const int n = this->nParticles_;
    if (xHost_.size() != (size_t)3 * n) return false;
    for (int i = 0; i < n; i++) {
      typename TPose::Vec v;
      this->particleSet_[i]->get(v);
      if (v[0] != xHost_[(size_t)3 * i] || v[1] != xHost_[(size_t)3 * i + 1] || v[2] != xHost_[(size_t)3 * i + 2]) return false;
    }
    return true;
  }

  /* include/ParticleFilter.hpp:140 (pure virtual) / include/RBPHDFilter.hpp:306 (private there too).  The reference calls it
   * from inside update() only (:490); here the weighting of ALL particles is a phase of the device step that update() launches,
   * so there is nothing a per-particle call could do on its own: it refuses instead of computing something else. */
  void importanceWeighting(const uint idx) {
    (void)idx;
    throw std::logic_error("rfs::RBPHDFilter (rfsgpu): importanceWeighting(idx) runs on the device inside update(); it cannot be called on its own");
  }

  void check(int rc, const char *what) {
    if (rc != RFSGPU_OK) throw std::runtime_error(std::string("rfsgpu ") + what + ": " + engine_.last_error());
  }

  /* The three config structs of the reference objects -> the engine, before every predict / update (they are public members
   * the driver may change at any time; src/rbphdslam_VictoriaPark.cpp:510,538 changes the landmark noise every step). */
  void pushConfiguration() {
    rfsgpu_filter_config c;
    std::memset(&c, 0, sizeof(c));                          /* (padding bytes: the structs are compared below) */
    c.birthGaussianWeight = config.birthGaussianWeight_;
    c.birthGaussianMeasurementCountThreshold = config.birthGaussianMeasurementCountThreshold_;
    c.birthGaussianMeasurementCheckThreshold = config.birthGaussianMeasurementCheckThreshold_;
    c.birthGaussianMeasurementSupportDist = config.birthGaussianMeasurementSupportDist_;
    c.birthGaussianCurrentMeasurementCountThreshold = config.birthGaussianCurrentMeasurementCountThreshold_;
    c.newGaussianCreateInnovMDThreshold = config.newGaussianCreateInnovMDThreshold_;
    c.importanceWeightingEvalPointCount = config.importanceWeightingEvalPointCount_;
    c.importanceWeightingEvalPointGuassianWeight = config.importanceWeightingEvalPointGuassianWeight_;
    c.importanceWeightingMeasurementLikelihoodMDThreshold = config.importanceWeightingMeasurementLikelihoodMDThreshold_;
    c.gaussianMergingThreshold = config.gaussianMergingThreshold_;
    c.gaussianMergingCovarianceInflationFactor = config.gaussianMergingCovarianceInflationFactor_;
    c.gaussianPruningThreshold = config.gaussianPruningThreshold_;
    c.minUpdatesBeforeResample = config.minUpdatesBeforeResample_;
    c.minMeasurementsBeforeResample = config.minMeasurementsBeforeResample_;
    c.useClusterProcess = config.useClusterProcess_ ? 1 : 0;

    rfsgpu_kf_config k;
    std::memset(&k, 0, sizeof(k));
    k.rangeInnovationThreshold = kf_->config.rangeInnovationThreshold_;
    k.bearingInnovationThreshold = kf_->config.bearingInnovationThreshold_;

    typename TLandmark::Mat Q;
    lmkModelPtr_->getNoise(Q);                                    /* include/ProcessModel.hpp:92-95 */
    const int D = rfsgpu_model_of<MeasurementModel>::dim;
    double q[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = 0; r < D; r++)
      for (int cc = 0; cc < D; cc++) q[D * r + cc] = Q(r, cc);

    /* A recorded predict must run under the configuration predict() saw (birth weight, R, Q, ...): if the driver has changed
     * anything since, it happens now, before the new values go to the engine. */
    const bool same = pushed_.valid && std::memcmp(&pushed_.c, &c, sizeof(c)) == 0 && std::memcmp(&pushed_.k, &k, sizeof(k)) == 0 &&
                      std::memcmp(pushed_.q, q, sizeof(q)) == 0;
    if (!same) {
      flushPredict();
      check(engine_.set_filter_config(&c), "set_filter_config");
      check(engine_.set_kf_config(&k), "set_kf_config");
      check(engine_.set_lmk_process_noise(q), "set_lmk_process_noise");
      pushed_.c = c; pushed_.k = k; std::memcpy(pushed_.q, q, sizeof(q));
    }
    pushModel(model_.get());
    pushed_.valid = true;
  }
  void pushModel(MeasurementModel_RngBrg *m) {                     /* include/MeasurementModel_RngBrg.hpp:65-71 */
    rfsgpu_rngbrg_config c;
    std::memset(&c, 0, sizeof(c));
    MeasurementModel_RngBrg::TMeasurement::Mat R;
    m->getNoise(R);
    c.R[0] = R(0, 0); c.R[1] = R(0, 1); c.R[2] = R(1, 0); c.R[3] = R(1, 1);
    c.probabilityOfDetection = m->config.probabilityOfDetection_;
    c.uniformClutterIntensity = m->config.uniformClutterIntensity_;
    c.rangeLimMax = m->config.rangeLimMax_;
    c.rangeLimMin = m->config.rangeLimMin_;
    c.rangeLimBuffer = m->config.rangeLimBuffer_;
    if (pushed_.valid && std::memcmp(&pushed_.m, &c, sizeof(c)) == 0) return;    /* unchanged since the last push */
    flushPredict();                                                   /* (R enters the birth covariance) */
    check(engine_.set_model_rngbrg(&c), "set_model_rngbrg");
    pushed_.m = c;
  }
  void pushModel(rfsgpu_vp_model_handle *m) {                      /* include/MeasurementModel_VictoriaPark.hpp:136-152 */
    rfsgpu_vp_config c;
    Measurement3d::Mat R;
    m->getNoise(R);
    for (int r = 0; r < 3; r++)
      for (int cc = 0; cc < 3; cc++) c.R[3 * r + cc] = R(r, cc);
    if (!m->haveSlb()) throw std::runtime_error("rfsgpu: MeasurementModel_VictoriaPark::setNoise(R, Slb) has not been called through getMeasurementModel()");
    c.Slb = m->Slb();                                                /* the Slb handed to setNoise(R, Slb) (:59) */
    c.nPd = (int)m->config.probabilityOfDetection_.size();
    if (c.nPd > RFSGPU_VP_MAX_PD) throw std::runtime_error("rfsgpu: Pd table longer than RFSGPU_VP_MAX_PD");
    for (int k = 0; k < c.nPd; k++) c.PdTable[k] = m->config.probabilityOfDetection_[k];
    c.expectedClutterNumber = m->config.expectedClutterNumber_;
    c.rangeLimMax = m->config.rangeLimMax_;
    c.rangeLimMin = m->config.rangeLimMin_;
    c.bearingLimitMax = m->config.bearingLimitMax_;
    c.bearingLimitMin = m->config.bearingLimitMin_;
    c.bufferZonePd = m->config.bufferZonePd_;
    check(engine_.set_model_victoriapark(&c), "set_model_victoriapark");
    if (m->laserScanSerial() != scanPushed_) {                        /* a new scan since the last push (driver :582) */
      check(engine_.set_laser_scan(m->laserScan().data(), (int)m->laserScan().size()), "set_laser_scan");
      scanPushed_ = m->laserScanSerial();
    }
  }

  /* pose mean + covariance of every particle (the covariance enters S in the 2-D model, src/MeasurementModel_RngBrg.cpp:102) */
  void gatherPoses(std::vector<double> &x, std::vector<double> &P) {
    const int n = this->nParticles_;
    for (int i = 0; i < n; i++) {
      typename TPose::Vec v;
      typename TPose::Mat S;
      this->particleSet_[i]->get(v, S);
      for (int r = 0; r < 3; r++) {
        x[(size_t)3 * i + r] = v[r];
        for (int c = 0; c < 3; c++) P[(size_t)9 * i + 3 * r + c] = S(r, c);
      }
    }
  }
  void pushPoses() {
    std::vector<double> x((size_t)3 * this->nParticles_), P((size_t)9 * this->nParticles_);
    gatherPoses(x, P);
    check(engine_.set_poses(x.data(), P.data(), 9), "set_poses");
  }
  /* ParticleFilter::resample() (include/ParticleFilter.hpp:399-492) with n = nParticles_, restated only because the maps are
   * not in Particle::data_: the decision, the draw and the slot assignment are the reference's, statement for statement;
   * `particleSet_[next] = particleSet_[idx]->copy()` (:473) copies the host part (pose, id, the empty mixture object) and sets
   * src_slot[next] = idx; ONE rfsgpu_resample_apply(src_slot) then deep-copies the mixtures on the device.  The engine keeps the
   * same id_ / idParent_ bookkeeping per slot as the Particle objects here and performs addBirthGaussians' lazy, slot-ordered
   * copy of unused_measurements_ / birthGaussians_ (:1005-1011) inside the next rfsgpu_predict_map(1), exactly as written
   * (RFSGPU_INHERIT_REFERENCE, the engine's default). */
  bool resampleWithDeviceMaps() {
    this->normalizeWeights();                                                   /* :402 */
    const int n = this->nParticles_;
    double sum_of_weight_squared = 0;                                            /* :405-415 */
    for (int i = 0; i < n; i++) {
      const double w_i = this->particleSet_[i]->getWeight();
      sum_of_weight_squared += (w_i * w_i);
    }
    const double nEffParticles = 1.0 / sum_of_weight_squared;
    if (nEffParticles > this->effNParticles_t_ && nEffParticles / n > this->effNParticles_t_percent_) return false;

    const double randomNum_0_to_1 = drand48();                                  /* :421 */
    unsigned int idx = 0;
    const double sample_interval = 1.0 / double(n);
    double sample_point = sample_interval * randomNum_0_to_1;
    double cumulative_weight = this->particleSet_[idx]->getWeight();
    std::vector<char> flag_particle_sampled(n, 0);
    std::vector<unsigned int> sampled_idx(n, 0);
    for (int i = 0; i < n; i++) {                                               /* :433-444 */
      while (sample_point > cumulative_weight) {
        idx++;
        if ((int)idx >= n) { idx = n - 1; break; }   /* (the reference reads past the end of particleSet_ when round-off leaves the
                                                        last sample point above the final cumulative weight; clamped here) */
        cumulative_weight += this->particleSet_[idx]->getWeight();
      }
      sampled_idx[i] = idx;
      flag_particle_sampled[idx] = 1;
      sample_point += sample_interval;
    }
    std::vector<int> src_slot(n);
    for (int i = 0; i < n; i++) src_slot[i] = i;
    unsigned int idx_prev = 0, next_unsampled_idx = 0;
    for (int i = 0; i < n; i++) {                                               /* :446-479 */
      bool firstTime = true;
      idx = sampled_idx[i];
      if (i > 0 && idx == idx_prev) firstTime = false;
      idx_prev = idx;
      if (firstTime) {                                                           /* case 1 (idx < n always holds here) */
        this->particleSet_[idx]->setParentId(this->particleSet_[idx]->getId());
      } else {                                                                   /* case 2 */
        while (flag_particle_sampled[next_unsampled_idx] == 1) next_unsampled_idx++;
        this->particleSet_[next_unsampled_idx] = this->particleSet_[idx]->copy();   /* pose, id, (empty) mixture object */
        this->particleSet_[next_unsampled_idx]->setParentId(this->particleSet_[idx]->getId());
        src_slot[next_unsampled_idx] = (int)idx;
        next_unsampled_idx++;
      }
    }
    const long long tr0 = prof_now();
    check(engine_.resample_apply(src_slot.data()), "resample_apply");   /* also resets the device weights to 1 */
    if (prof_) { profResNs_ += prof_now() - tr0; profResN_++; }
    if (xOnDeviceValid_ && xHost_.size() == (size_t)3 * n)                /* the device's pose array travelled with the particles: so does its host copy */
      for (int i = 0; i < n; i++)
        if (src_slot[i] != i)
          for (int r = 0; r < 3; r++) xHost_[(size_t)3 * i + r] = xHost_[(size_t)3 * src_slot[i] + r];
    for (int i = 0; i < n; i++) this->particleSet_[i]->setWeight(1);            /* :486-489 */
    return true;
  }
};

}  // namespace rfs

#endif
