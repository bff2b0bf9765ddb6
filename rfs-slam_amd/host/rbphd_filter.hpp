// rbphd_filter.hpp -- C++ host mirror of rfs::RBPHDFilter for the device path (header-only, over the C ABI in
// include/rfsgpu.h).  Same public surface as the reference class template instantiated by rbphdslam2dSim
//   RBPHDFilter<MotionModel_Odometry2d, StaticProcessModel<Landmark2d>, MeasurementModel_RngBrg, KalmanFilter_RngBrg>
// (reference include/RBPHDFilter.hpp:72-251, src/rbphdslam2dSim.cpp:446-491): predict / update / getGMSize /
// getLandmark / setParticlePose / getKalmanFilter / getLmkProcessModel / getMeasurementModel / getProcessModel /
// getTimingInfo / public `config`, same argument meaning and error behaviour (void predict/update, getGMSize -> -1,
// getLandmark -> false on bad indices).  What stays on the host, exactly as in the reference: pose propagation with
// host RNG (ParticleFilter::propagate, include/ParticleFilter.hpp:322-341), the resampling decision and the
// systematic draw with drand48() (ParticleFilter::resample :399-492), the resample counters (:526-539).
// Eigen/Boost are not available in this image, so Pose2d / Landmark2d / Measurement2d are plain structs here;
// INTEGRATION.md shows the Eigen-typed binding for the reference tree.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <random>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "rfsgpu.h"

namespace rfs_amd {

struct Pose2d {
  double x[3] = {0, 0, 0};
  double P[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // row-major covariance (enters S, MeasurementModel_RngBrg.cpp:102)
};
struct Odometry2d {
  double u[3] = {0, 0, 0};
  double t = 0;
};
struct Measurement2d {
  double z[2] = {0, 0};
  double t = 0;
};

// MotionModel_Odometry2d::step (reference src/ProcessModel_Odometry2D.cpp:40-88) + ProcessModel::sample
// (include/ProcessModel.hpp:126-150): host-side, RNG-bound, 3 doubles per particle.
class MotionModel_Odometry2d {
 public:
  void setNoise(const double Q[9]) { std::memcpy(Q_, Q, sizeof(Q_)); chol(); }
  void getNoise(double Q[9]) const { std::memcpy(Q, Q_, sizeof(Q_)); }
  static void step(Pose2d &s_k, const Pose2d &s_km, const Odometry2d &in) {
    const double ct = std::cos(s_km.x[2]), st = std::sin(s_km.x[2]);
    // C_km = [ct st; -st ct]; p_k = p_km + C_km^T * dp
    const double px = s_km.x[0] + (ct * in.u[0] + (-st) * in.u[1]);
    const double py = s_km.x[1] + (st * in.u[0] + ct * in.u[1]);
    const double cd = std::cos(in.u[2]), sd = std::sin(in.u[2]);
    // C_k = C_d * C_km ; theta = atan2(C_k(0,1), C_k(0,0))
    const double c00 = cd * ct + sd * (-st), c01 = cd * st + sd * ct;
    s_k.x[0] = px;
    s_k.x[1] = py;
    s_k.x[2] = std::atan2(c01, c00);
  }
  template <class RNG>
  void sample(Pose2d &s_k, const Pose2d &s_km, const Odometry2d &in, bool useModelNoise, RNG &rng) const {
    Pose2d out = s_km;
    step(out, s_km, in);
    bool zeroQ = true;
    for (double q : Q_) zeroQ = zeroQ && (q == 0);
    if (useModelNoise && !zeroQ) {
      std::memcpy(out.P, Q_, sizeof(Q_));  // the sampled pose keeps covariance Q (ProcessModel.hpp:145-149)
      std::normal_distribution<double> N01(0.0, 1.0);
      const double n0 = N01(rng), n1 = N01(rng), n2 = N01(rng);
      out.x[0] += L_[0] * n0;
      out.x[1] += L_[3] * n0 + L_[4] * n1;
      out.x[2] += L_[6] * n0 + L_[7] * n1 + L_[8] * n2;
    }
    s_k = out;
  }

 private:
  double Q_[9] = {0}, L_[9] = {0};
  void chol() {  // lower Cholesky factor of Q (RandomVec::sample uses Eigen::LLT)
    std::memset(L_, 0, sizeof(L_));
    for (int i = 0; i < 3; i++)
      for (int j = 0; j <= i; j++) {
        double s = Q_[3 * i + j];
        for (int k = 0; k < j; k++) s -= L_[3 * i + k] * L_[3 * j + k];
        L_[3 * i + j] = (i == j) ? (s > 0 ? std::sqrt(s) : 0.0) : (L_[3 * j + j] != 0 ? s / L_[3 * j + j] : 0.0);
      }
  }
};

class RBPHDFilter2d {
 public:
  // --- nested "model" handles so that driver code reads like the reference's (getXxx()->config.yyy_ = ...) ---
  struct MeasurementModel {
    struct Config {
      double probabilityOfDetection_ = 0.95, uniformClutterIntensity_ = 0.1, rangeLimMax_ = 5, rangeLimMin_ = 0.3, rangeLimBuffer_ = 0.25;
    } config;
    double R[4] = {0, 0, 0, 0};
    void setNoise(const double Rin[4]) { std::memcpy(R, Rin, sizeof(R)); }
  };
  struct KalmanFilter {
    struct Config {
      double rangeInnovationThreshold_ = -1, bearingInnovationThreshold_ = -1;
    } config;
  };
  struct LmkProcessModel {
    double Q[4] = {0, 0, 0, 0};
    void setNoise(const double Qin[4]) { std::memcpy(Q, Qin, sizeof(Q)); }
  };
  struct Config {  // RBPHDFilter::Config (RBPHDFilter.hpp:90-146), reference member names
    double birthGaussianWeight_ = 0.25;
    unsigned birthGaussianMeasurementCountThreshold_ = 1, birthGaussianMeasurementCheckThreshold_ = 1;
    double birthGaussianMeasurementSupportDist_ = 1;
    unsigned birthGaussianCurrentMeasurementCountThreshold_ = 1;
    double newGaussianCreateInnovMDThreshold_ = 0.2;
    int importanceWeightingEvalPointCount_ = 8;
    double importanceWeightingEvalPointGuassianWeight_ = 0;
    double importanceWeightingMeasurementLikelihoodMDThreshold_ = 3.0;
    double gaussianMergingThreshold_ = 0.5, gaussianMergingCovarianceInflationFactor_ = 1.5, gaussianPruningThreshold_ = 0.2;
    int minUpdatesBeforeResample_ = 1, minMeasurementsBeforeResample_ = 1;
    bool useClusterProcess_ = false;
  } config;
  typedef rfsgpu_timing TimingInfo;  // same 14 fields as RBPHDFilter::TimingInfo (:152-167)

  // max_particles > n reserves room for particle sets that grow (MH-FastSLAM); 0 = n.
  explicit RBPHDFilter2d(int n, int device_id = 0, int gm_capacity = 512, int max_particles = 0, int model = RFSGPU_MODEL_RNGBRG_2D)
      : n_(n), nInit_(n), poses_(n), weights_(n, 1.0), rng_(std::rand()) {
    int rc = rfsgpu_create_ex(&h_, model, n, device_id, gm_capacity, max_particles > n ? max_particles : n);
    if (rc != RFSGPU_OK) throw std::runtime_error("rfsgpu_create_ex failed with status " + std::to_string(rc) + " (no gfx950 device? there is no CPU fallback)");
    effNParticles_t_ = double(n) / 4.0;  // ParticleFilter.hpp:232
    effNParticles_t_percent_ = effNParticles_t_ / n;
    // (RFSGPU_PHASE_TIMING=1: update() as separate launches per phase, for a TimingInfo printout with its buckets apart)
    if (const char *pt = std::getenv("RFSGPU_PHASE_TIMING")) rfsgpu_set_phase_timing(h_, std::atoi(pt));
  }
  // The same filter over SEVERAL GPUs from this one host thread (rfsgpu_group_*: contiguous particle blocks, one shard per
  // device id, global resampling with peer-copy migration).  Everything below dispatches on g_; the public interface is
  // unchanged.  (2-D model; device ids may repeat, which is how a single-GPU box exercises it.)
  RBPHDFilter2d(int n, const std::vector<int> &device_ids, int gm_capacity = 512)
      : n_(n), nInit_(n), poses_(n), weights_(n, 1.0), rng_(std::rand()) {
    int rc = rfsgpu_group_create(&g_, RFSGPU_MODEL_RNGBRG_2D, n, device_ids.data(), (int)device_ids.size(), gm_capacity);
    if (rc != RFSGPU_OK) throw std::runtime_error("rfsgpu_group_create failed with status " + std::to_string(rc));
    h_ = rfsgpu_group_shard(g_, 0);   // (timing buckets are read from shard 0: all shards run the same launches)
    effNParticles_t_ = double(n) / 4.0;
    effNParticles_t_percent_ = effNParticles_t_ / n;
  }
  ~RBPHDFilter2d() { if (g_) rfsgpu_group_destroy(g_); else rfsgpu_destroy(h_); }
  RBPHDFilter2d(const RBPHDFilter2d &) = delete;
  RBPHDFilter2d &operator=(const RBPHDFilter2d &) = delete;

  MotionModel_Odometry2d *getProcessModel() { return &motion_; }
  LmkProcessModel *getLmkProcessModel() { return &lmk_; }
  MeasurementModel *getMeasurementModel() { return &meas_; }
  KalmanFilter *getKalmanFilter() { return &kf_; }
  int getParticleCount() const { return n_; }
  void setEffectiveParticleCountThreshold(double t) {  // ParticleFilter.hpp:386-391
    effNParticles_t_ = t;
    effNParticles_t_percent_ = t / n_;
  }
  double getEffectiveParticleCountThreshold() const { return effNParticles_t_; }
  const Pose2d &getParticlePose(int i) const {
    pullPoses();
    return poses_[i];
  }
  double getParticleWeight(int i) {
    pullWeights();
    return weights_[i];
  }
  bool resampleOccured() const { return resampleOccured_; }

  // RBPHDFilter::setParticlePose (:1181-1186)
  void setParticlePose(int i, const Pose2d &p) {
    pullPoses();
    poses_[i] = p;
    posesDirty_ = true;
    xbufFresh_ = false;
  }

  // RBPHDFilter::predict (:415-442): birth Gaussians at the pre-propagation pose, host propagation, Sigma += Q.
  void predict(const Odometry2d &u, double /*dT*/, bool useModelNoise = true, bool /*useInputNoise*/ = false, bool birthGaussianCheck = true) {
    pushConfig();
    pushPoses();
    check(g_ ? rfsgpu_group_predict_map(g_, birthGaussianCheck ? 1 : 0) : rfsgpu_predict_map(h_, birthGaussianCheck ? 1 : 0), "predict_map");
    for (int i = 0; i < n_; i++) {  // ParticleFilter::propagate
      Pose2d xk;
      motion_.sample(xk, poses_[i], u, useModelNoise, rng_);
      poses_[i] = xk;
    }
    posesDirty_ = true;
  }

  // RBPHDFilter::update (:444-541).  Z is consumed (swapped into the filter and cleared, ParticleFilter.hpp:316-320).
  void update(std::vector<Measurement2d> &Z) {
    nUpdatesSinceResample_++;
    std::vector<Measurement2d> meas;
    meas.swap(Z);
    Z.clear();
    if (meas.empty()) return;  // :450-452
    nMeasurementsSinceResample_ += (unsigned)meas.size();
    pushConfig();
    pushPoses();
    std::vector<double> z(2 * meas.size());
    for (size_t k = 0; k < meas.size(); k++) { z[2 * k] = meas[k].z[0]; z[2 * k + 1] = meas[k].z[1]; }
    check(g_ ? rfsgpu_group_update(g_, z.data(), (int)meas.size(), nullptr) : rfsgpu_update(h_, z.data(), (int)meas.size()), "update");
    weightsStale_ = true;
    resampleOccured_ = false;
    if (nUpdatesSinceResample_ >= (unsigned)config.minUpdatesBeforeResample_ &&
        nMeasurementsSinceResample_ >= (unsigned)config.minMeasurementsBeforeResample_)
      resampleOccured_ = resample();
    if (resampleOccured_) {
      nUpdatesSinceResample_ = 0;
      nMeasurementsSinceResample_ = 0;
    } else {
      normalizeWeights();
    }
  }

  // RBPHDFilter::getGMSize / getLandmark (:1152-1178)
  int getGMSize(int i) { return g_ ? rfsgpu_group_gm_size(g_, i) : rfsgpu_gm_size(h_, i); }
  bool getLandmark(int i, int m, double u[2], double S[4], double &w) {
    return (g_ ? rfsgpu_group_get_landmark(g_, i, m, u, S, &w) : rfsgpu_get_landmark(h_, i, m, u, S, &w)) == RFSGPU_OK;
  }

  TimingInfo *getTimingInfo() {
    rfsgpu_get_timing(h_, &timing_);
    return &timing_;
  }
  rfsgpu_filter *handle() { return h_; }

 protected:
  rfsgpu_filter *h_ = nullptr;
  rfsgpu_group *g_ = nullptr;   // set: the particle set is sharded over several devices
  int n_, nInit_;
  MotionModel_Odometry2d motion_;
  LmkProcessModel lmk_;
  MeasurementModel meas_;
  KalmanFilter kf_;
  mutable std::vector<Pose2d> poses_;
  // device-side propagation (RBPHDFilterVP::setDeviceMotion): the device owns the poses, the host copy is fetched on demand
  bool devicePoses_ = false;
  mutable bool hostPosesStale_ = false;
  // propagations queued for the device (RBPHDFilterVP with setDeviceMotion): a run of odometry messages goes out as ONE launch
  // (rfsgpu_propagate_ackerman_run_async) before the next call that reads the poses
  mutable std::vector<double> pendU_, pendVar_, pendDt_;
  double pendGeom_[4] = {0, 0, 0, 0};
  unsigned long long motionSeed_ = 0;
  mutable unsigned long long motionCall_ = 0;
  void flushMotion() const {
    if (pendDt_.empty()) return;
    const int n = (int)pendDt_.size();
    if (rfsgpu_propagate_ackerman_run_async(h_, n, pendU_.data(), pendVar_.data(), pendDt_.data(), pendGeom_, motionSeed_, motionCall_) != RFSGPU_OK)
      throw std::runtime_error(std::string("propagate_ackerman_run: ") + rfsgpu_last_error(h_));
    motionCall_ += (unsigned long long)n;
    pendU_.clear(); pendVar_.clear(); pendDt_.clear();
  }
  void pullPoses() const {
    flushMotion();
    if (!hostPosesStale_) return;
    std::vector<double> x(3 * (size_t)n_);
    if (rfsgpu_get_poses(h_, x.data()) != RFSGPU_OK) throw std::runtime_error(std::string("get_poses: ") + rfsgpu_last_error(h_));
    for (int i = 0; i < n_; i++) std::memcpy(poses_[i].x, &x[3 * i], 3 * sizeof(double));
    hostPosesStale_ = false;
  }
  std::vector<double> weights_;
  std::mt19937 rng_;
  double effNParticles_t_, effNParticles_t_percent_;
  unsigned nUpdatesSinceResample_ = 0, nMeasurementsSinceResample_ = 0;
  bool resampleOccured_ = false, posesDirty_ = true, weightsStale_ = false;
  bool xbufFresh_ = false;   // (RBPHDFilterVP) its packed pose buffer already holds the current poses
  TimingInfo timing_{};

  void check(int rc, const char *what) {
    if (rc != RFSGPU_OK) throw std::runtime_error(std::string(what) + ": " + (g_ ? rfsgpu_group_last_error(g_) : rfsgpu_last_error(h_)));
  }
  void pushConfig() {
    pushFilterConfig();
    rfsgpu_rngbrg_config m;
    std::memcpy(m.R, meas_.R, sizeof(m.R));
    m.probabilityOfDetection = meas_.config.probabilityOfDetection_;
    m.uniformClutterIntensity = meas_.config.uniformClutterIntensity_;
    m.rangeLimMax = meas_.config.rangeLimMax_;
    m.rangeLimMin = meas_.config.rangeLimMin_;
    m.rangeLimBuffer = meas_.config.rangeLimBuffer_;
    check(g_ ? rfsgpu_group_set_model_rngbrg(g_, &m) : rfsgpu_set_model_rngbrg(h_, &m), "set_model_rngbrg");
    check(g_ ? rfsgpu_group_set_lmk_process_noise(g_, lmk_.Q) : rfsgpu_set_lmk_process_noise(h_, lmk_.Q), "set_lmk_process_noise");
  }
  // RBPHDFilter::Config + the Kalman filter's gates: the model-independent part
  void pushFilterConfig() {
    rfsgpu_filter_config c;
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
    check(g_ ? rfsgpu_group_set_filter_config(g_, &c) : rfsgpu_set_filter_config(h_, &c), "set_filter_config");
    rfsgpu_kf_config k{kf_.config.rangeInnovationThreshold_, kf_.config.bearingInnovationThreshold_};
    check(g_ ? rfsgpu_group_set_kf_config(g_, &k) : rfsgpu_set_kf_config(h_, &k), "set_kf_config");
  }
  void pushPoses() {
    if (!posesDirty_) return;
    std::vector<double> x(3 * (size_t)n_), P(9 * (size_t)n_);
    bool anyCov = false;
    for (int i = 0; i < n_; i++) {
      std::memcpy(&x[3 * i], poses_[i].x, 3 * sizeof(double));
      std::memcpy(&P[9 * i], poses_[i].P, 9 * sizeof(double));
      for (int t = 0; t < 9; t++) anyCov = anyCov || (poses_[i].P[t] != 0.0);
    }
    // (all-zero covariances -- setParticlePose'd poses, the Ackerman model -- cross as "no covariance": 24 B per particle)
    check(g_ ? rfsgpu_group_set_poses(g_, x.data(), anyCov ? P.data() : nullptr, anyCov ? 9 : 0)
             : rfsgpu_set_poses(h_, x.data(), anyCov ? P.data() : nullptr, anyCov ? 9 : 0), "set_poses");
    posesDirty_ = false;
  }
  void pullWeights() {
    if (!weightsStale_) return;
    check(g_ ? rfsgpu_group_get_weights(g_, weights_.data()) : rfsgpu_get_weights(h_, weights_.data()), "get_weights");
    weightsStale_ = false;
  }
  // ParticleFilter::normalizeWeights (ParticleFilter.hpp:352-363): sum on the device, divide on the device.
  void normalizeWeights() {
    double s[2];
    if (g_) {
      check(rfsgpu_group_normalize(g_, s), "group_normalize");
    } else {
      check(rfsgpu_weight_sums(h_, s), "weight_sums");
      check(rfsgpu_normalize_weights(h_, s[0], nullptr), "normalize_weights");
    }
    weightsStale_ = true;
  }
  // ParticleFilter::resample(n, forceResample) (ParticleFilter.hpp:399-492).  nOut == 0 or > nParticles_ keeps the count; a
  // smaller nOut draws nOut samples from all particles and shrinks the set (FastSLAM::resampleWithMapCopy).
  bool resample(unsigned nOut = 0, bool force = false, bool alreadyNormalized = false) {
    flushMotion();
    if (!alreadyNormalized) normalizeWeights();
    pullWeights();
    const int N = n_;
    if (!force) {
      double s2 = 0;
      for (int i = 0; i < N; i++) s2 += weights_[i] * weights_[i];
      const double nEff = 1.0 / s2;
      if (nEff > effNParticles_t_ && nEff / N > effNParticles_t_percent_) return false;
    }
    const int n = (nOut == 0 || (int)nOut > N) ? N : (int)nOut;
    const double u01 = drand48();
    unsigned idx = 0;
    const double interval = 1.0 / double(n);
    double sample_point = interval * u01;
    double cumulative = weights_[0];
    std::vector<char> sampled(N, 0);
    std::vector<unsigned> sampled_idx(n, 0);
    for (int i = 0; i < n; i++) {
      while (sample_point > cumulative && (int)idx < N - 1) {
        idx++;
        cumulative += weights_[idx];
      }
      sampled_idx[i] = idx;
      sampled[idx] = 1;
      sample_point += interval;
    }
    std::vector<int> src(n);
    for (int i = 0; i < n; i++) src[i] = i;
    unsigned idx_prev = 0, next_unsampled = 0;
    for (int i = 0; i < n; i++) {
      idx = sampled_idx[i];
      const bool firstTime = !(i > 0 && idx == idx_prev);
      idx_prev = idx;
      if ((int)idx < n && firstTime) continue;  // case 1: keeps its slot
      while (next_unsampled < (unsigned)N && sampled[next_unsampled] == 1) next_unsampled++;
      src[next_unsampled] = (int)idx;           // cases 2-4: a copy (or a move from beyond the new count) into a free slot
      poses_[next_unsampled] = poses_[idx];     // Particle::copy copies the pose too
      next_unsampled++;
    }
    if (g_) {
      if (n != N) throw std::runtime_error("resample(n < nParticles) is not offered by the multi-GPU form");
      check(rfsgpu_group_apply_plan(g_, src.data()), "group_apply_plan");   // local gathers + peer-copy migration
    } else if (n == N) check(rfsgpu_resample_apply(h_, src.data()), "resample_apply");
    else check(rfsgpu_resample_apply_n(h_, src.data(), n), "resample_apply_n");
    n_ = n;
    poses_.resize(n);
    weights_.assign(n, 1.0);
    if (devicePoses_ && hostPosesStale_) {   // the device gathered its own (current) poses with the maps; the host copy stays stale
      posesDirty_ = false;
    } else {
      posesDirty_ = true;
      xbufFresh_ = false;
    }
    weightsStale_ = false;
    return true;
  }
};


// rfs::FastSLAM<MotionModel_Odometry2d, StaticProcessModel<Landmark2d>, MeasurementModel_RngBrg, KalmanFilter_RngBrg>
// (reference include/FastSLAM.hpp) over the same C ABI: the handle's mixtures are the landmark maps (weights = log-odds of
// existence), rfsgpu_fastslam_update is updateMap for every particle.  config.maxNDataAssocHypotheses_ > 1 (MH-FastSLAM)
// multiplies particles inside an update: construct with max_hypotheses so that the handle has room for
// nParticlesMax_ * max_hypotheses particles (the call fails loudly with RFSGPU_ERR_CAPACITY otherwise).
class FastSLAM2d : public RBPHDFilter2d {
 public:
  struct Config {  // FastSLAM::Config (FastSLAM.hpp:106-132), reference member names and constructor defaults (:243-257)
    int minUpdatesBeforeResample_ = 1, minMeasurementsBeforeResample_ = 1;
    bool reportTimingInfo_ = false;
    double landmarkExistencePrior_ = 0.5, mapExistencePruneThreshold_ = -3.0, minLogMeasurementLikelihood_ = -10.0;
    int nParticlesMax_ = 0;
    unsigned maxNDataAssocHypotheses_ = 1;
    double maxDataAssocLogLikelihoodDiff_ = 5, landmarkCandidateMeasurementSupportDist_ = 1;
    unsigned landmarkCandidateMeasurementCountThreshold_ = 1, landmarkCandidateCurrentMeasurementCountThreshold_ = 1,
             landmarkCandidateMeasurementCheckThreshold_ = 2;
    double landmarkLockWeight_ = 10;
    unsigned pruningMeasurementsThreshold_ = 0;
  } config;

  explicit FastSLAM2d(int n, int device_id = 0, int gm_capacity = 512, unsigned max_hypotheses = 1)
      : RBPHDFilter2d(n, device_id, gm_capacity, max_hypotheses > 1 ? 3 * n * (int)max_hypotheses : n) {
    config.nParticlesMax_ = 3 * n;  // FastSLAM.hpp:250
    config.maxNDataAssocHypotheses_ = max_hypotheses;
  }

  // FastSLAM::predict (:362-385): propagate the particles, staticStep on every landmark (no births in predict)
  void predict(const Odometry2d &u, double dT, bool useModelNoise = true, bool useInputNoise = false) {
    RBPHDFilter2d::predict(u, dT, useModelNoise, useInputNoise, /*birthGaussianCheck=*/false);
  }

  // FastSLAM::update (:387-421) + resampleWithMapCopy (:708-735)
  void update(std::vector<Measurement2d> &Z) {
    nUpdatesSinceResample_++;
    std::vector<Measurement2d> meas;
    meas.swap(Z);
    Z.clear();
    if (meas.empty()) return;  // :401-402
    nMeasurementsSinceResample_ += (unsigned)meas.size();
    pushConfig();
    pushFastSlamConfig();
    pushPoses();
    std::vector<double> z(2 * meas.size());
    for (size_t k = 0; k < meas.size(); k++) { z[2 * k] = meas[k].z[0]; z[2 * k + 1] = meas[k].z[1]; }
    check(rfsgpu_fastslam_update(h_, z.data(), (int)meas.size()), "fastslam_update");
    weightsStale_ = true;
    const int nNow = rfsgpu_n_particles(h_);
    if (nNow != n_) {  // hypotheses beyond the first became new particles (:462-476): their poses are their parents'
      std::vector<int> parent(nNow);
      check(rfsgpu_particle_parents(h_, parent.data(), nNow), "particle_parents");
      poses_.resize(nNow);
      for (int i = n_; i < nNow; i++) poses_[i] = poses_[parent[i]];
      weights_.resize(nNow, 1.0);
      n_ = nNow;
    }
    resampleOccured_ = false;
    if (n_ > config.nParticlesMax_)  // :711-712
      resampleOccured_ = resample((unsigned)nInit_, true);
    else if (nUpdatesSinceResample_ >= (unsigned)config.minUpdatesBeforeResample_ &&
             nMeasurementsSinceResample_ >= (unsigned)config.minMeasurementsBeforeResample_)
      resampleOccured_ = resample((unsigned)nInit_);  // landmark candidates travel with their particle (rfsgpu_resample_apply)
    check(rfsgpu_fastslam_set_resample_occured(h_, resampleOccured_ ? 1 : 0), "fastslam_set_resample_occured");
    if (resampleOccured_) {
      nUpdatesSinceResample_ = 0;
      nMeasurementsSinceResample_ = 0;
    } else {
      normalizeWeights();
    }
  }

 private:
  void pushFastSlamConfig() {
    rfsgpu_fastslam_config c;
    rfsgpu_default_fastslam_config(&c);
    c.minUpdatesBeforeResample = config.minUpdatesBeforeResample_;
    c.minMeasurementsBeforeResample = config.minMeasurementsBeforeResample_;
    c.landmarkExistencePrior = config.landmarkExistencePrior_;
    c.mapExistencePruneThreshold = config.mapExistencePruneThreshold_;
    c.minLogMeasurementLikelihood = config.minLogMeasurementLikelihood_;
    c.nParticlesMax = config.nParticlesMax_;
    c.maxNDataAssocHypotheses = config.maxNDataAssocHypotheses_;
    c.maxDataAssocLogLikelihoodDiff = config.maxDataAssocLogLikelihoodDiff_;
    c.landmarkCandidateMeasurementSupportDist = config.landmarkCandidateMeasurementSupportDist_;
    c.landmarkCandidateMeasurementCountThreshold = config.landmarkCandidateMeasurementCountThreshold_;
    c.landmarkCandidateCurrentMeasurementCountThreshold = config.landmarkCandidateCurrentMeasurementCountThreshold_;
    c.landmarkCandidateMeasurementCheckThreshold = config.landmarkCandidateMeasurementCheckThreshold_;
    c.landmarkLockWeight = config.landmarkLockWeight_;
    c.pruningMeasurementsThreshold = config.pruningMeasurementsThreshold_;
    check(rfsgpu_set_fastslam_config(h_, &c), "set_fastslam_config");
  }
};

// ---- Victoria Park (reference src/rbphdslam_VictoriaPark.cpp) ----------------------------------------------------------
struct Measurement3d {      // MeasurementModel_VictoriaPark::TMeasurement: range, bearing, trunk diameter
  double z[3] = {0, 0, 0};
  double t = 0;
};
struct AckermanInput {      // MotionModel_Ackerman2d::TInput: speed, steering angle (+ their variances for predict's input noise)
  double u[2] = {0, 0};
  double var[2] = {0, 0};
};

// MotionModel_Ackerman2d::step (reference src/ProcessModel_Ackerman2D.cpp:47-78): host-side, 3 doubles per particle.
class MotionModel_Ackerman2d {
 public:
  void setAckermanParams(double h, double l, double dx, double dy) { h_ = h; l_ = l; dx_ = dx; dy_ = dy; }
  void getAckermanParams(double g[4]) const { g[0] = h_; g[1] = l_; g[2] = dx_; g[3] = dy_; }
  void step(Pose2d &s_k, const Pose2d &s_km, double u_v, double u_r, double dt) const {
    const double r = s_km.x[2];
    const double c = std::cos(r), s = std::sin(r), t = std::tan(u_r);
    const double v = u_v / (1 - t * h_ / l_);
    s_k = s_km;
    s_k.x[0] = s_km.x[0] + dt * (v * c - v / l_ * t * (dx_ * s + dy_ * c));
    s_k.x[1] = s_km.x[1] + dt * (v * s + v / l_ * t * (dx_ * c - dy_ * s));
    double th = s_km.x[2] + dt * v / l_ * t;
    if (th > PI_) th -= 2 * PI_;
    else if (th < -PI_) th += 2 * PI_;
    s_k.x[2] = th;
  }

 private:
  double h_ = 0.76, l_ = 2.83, dx_ = 3.78, dy_ = 0.50;
  const double PI_ = std::acos(-1.0);
};

// rfs::RBPHDFilter<MotionModel_Ackerman2d, StaticProcessModel<Landmark3d>, MeasurementModel_VictoriaPark,
// KalmanFilter_VictoriaPark> over the C ABI (model RFSGPU_MODEL_VICTORIAPARK_3D): the members the reference's Victoria Park
// driver touches (src/rbphdslam_VictoriaPark.cpp:344-398, 497-583), same names and argument meaning.  The pose, its host RNG
// and the resampling logic are the base class's.
class RBPHDFilterVP : public RBPHDFilter2d {
 public:
  struct MeasurementModelVP {
    struct Config {  // MeasurementModel_VictoriaPark::Config (include/MeasurementModel_VictoriaPark.hpp:150-158)
      std::vector<double> probabilityOfDetection_;
      double expectedClutterNumber_ = 0, rangeLimMax_ = 0, rangeLimMin_ = 0, bearingLimitMax_ = 0, bearingLimitMin_ = 0, bufferZonePd_ = 0;
    } config;
    double R[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Slb = 0;
    std::vector<double> scan;
    bool scanDirty = true;
    void setNoise(const double Rin[9], double SlbIn) { std::memcpy(R, Rin, sizeof(R)); Slb = SlbIn; }
    void setLaserScan(const std::vector<double> &s) { scan = s; scanDirty = true; }
  };
  struct LmkProcessModel3d {
    double Q[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    void setNoise(const double Qin[9]) { std::memcpy(Q, Qin, sizeof(Q)); }
  };

  explicit RBPHDFilterVP(int n, int device_id = 0, int gm_capacity = 192) : RBPHDFilter2d(n, device_id, gm_capacity, 0, RFSGPU_MODEL_VICTORIAPARK_3D) {}

  MotionModel_Ackerman2d *getProcessModel() { return &ackerman_; }
  // ParticleFilter::propagate on the device (rfsgpu_propagate_ackerman_async, csrc/motion.h): poses stay there, the host fetches
  // them when asked (getParticlePose).  Off by default: the host loop below is the reference's shape.
  void setDeviceMotion(bool on, unsigned long long seed = 0) { pullPoses(); devicePoses_ = on; motionSeed_ = seed; motionCall_ = 0; }
  LmkProcessModel3d *getLmkProcessModel() { return &lmk3_; }
  MeasurementModelVP *getMeasurementModel() { return &measVP_; }

  // RBPHDFilter::predict(u, dT, useModelNoise = false, useInputNoise, birthGaussianCheck) (:415-442) with
  // ProcessModel::sample's input-noise branch (include/ProcessModel.hpp:126-150): every particle draws its own input.
  double tCfg_ = 0, tIn_ = 0, tPm_ = 0, tProp_ = 0;   // host seconds per section of predict() (printed by the driver with -v)
  void predict(const AckermanInput &u, double dT, bool /*useModelNoise*/, bool useInputNoise, bool birthGaussianCheck) {
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t0 = now();
    pushConfigVP();
    auto t1 = now();
    // The device reads the poses in two places only: the map update, and the births of the first predict after an update --
    // and those are born at the poses the update just used, which are on the device already (update() pushed them, a
    // resampling gathered them there).  So the poses of a run of odometry messages are NOT sent message by message: only
    // when something other than propagation changed them (setParticlePose, resampling on the host side), or by update().
    if (posesDirty_ && !xbufFresh_) pushInputsAsync();
    auto t2 = now();
    // A predict that adds no births is one static step of the landmarks (Sigma += Q) and nothing else on the device: those are
    // counted and applied in ONE launch before the next call that reads the maps (same additions, same order, same bits) --
    // nine of ten Victoria Park messages are such predicts.
    if (birthGaussianCheck) {
      flushStatic();
      flushMotion();   // (the births below are born at the poses reached so far)
      check(rfsgpu_predict_map_async(h_, 1), "predict_map");   // stream-ordered: no host wait
    } else {
      pendingStatic_.insert(pendingStatic_.end(), lmk3_.Q, lmk3_.Q + 9);   // this predict's own Q (the driver scales it with dT)
    }
    auto t3 = now();
    tCfg_ += std::chrono::duration<double>(t1 - t0).count(); tIn_ += std::chrono::duration<double>(t2 - t1).count();
    tPm_ += std::chrono::duration<double>(t3 - t2).count();
    struct Acc { double &a; std::chrono::steady_clock::time_point t; ~Acc() { a += std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); } } acc{tProp_, t3};
    if (devicePoses_) {
      if (posesDirty_) pushInputsAsync();   // (poses set on the host since the last device step)
      const double zeroVar[2] = {0, 0};
      double geom[4];
      ackerman_.getAckermanParams(geom);
      std::memcpy(pendGeom_, geom, sizeof(geom));
      pendU_.insert(pendU_.end(), u.u, u.u + 2);
      pendVar_.insert(pendVar_.end(), useInputNoise ? u.var : zeroVar, (useInputNoise ? u.var : zeroVar) + 2);
      pendDt_.push_back(dT);                       // applied by flushMotion() before the next call that reads the poses
      hostPosesStale_ = true;
      return;
    }
    // ParticleFilter::propagate (include/ParticleFilter.hpp:322-339) is a serial loop over one random stream in the reference;
    // at 5000 particles that host loop (two normal draws + the Ackerman step per particle) costs more than the device work of
    // a lidar message.  Here the particles are taken in fixed chunks of 256, each with its own generator seeded from ONE draw of
    // the filter's master stream per predict: same statistics, results independent of the thread count, chunks in parallel.
    const unsigned long long stepSeed = rng_();
    const double sv = std::sqrt(u.var[0]), sr = std::sqrt(u.var[1]);
    const int nChunks = (n_ + 255) / 256;
    xbuf_.resize(3 * (size_t)n_);
    double *xb = xbuf_.data();
    // (RFS_HOST_THREADS, default min(16, CPUs the container may use) -- an OpenMP team as wide as the host's logical CPU count
    //  would oversubscribe a container with a CPU quota and spin)
    static const int hostThreads = [] {
      if (const char *e = std::getenv("RFS_HOST_THREADS")) { const int v = std::atoi(e); return v < 1 ? 1 : v; }
      int v = (int)std::thread::hardware_concurrency();
      if (FILE *fq = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {   // a container's CPU quota, when there is one
        long long q = 0, per = 0;
        if (std::fscanf(fq, "%lld %lld", &q, &per) == 2 && q > 0 && per > 0) v = std::min<long long>(v, q / per);
        std::fclose(fq);
      }
      return std::max(1, std::min(v, 16));
    }();
#pragma omp parallel for schedule(static) num_threads(hostThreads)
    for (int c = 0; c < nChunks; c++) {
      std::mt19937_64 eng(stepSeed + 0xD1B54A32D192ED03ull * (unsigned long long)(c + 1));
      std::normal_distribution<double> N01(0.0, 1.0);
      const int hi = std::min(n_, (c + 1) * 256);
      for (int i = c * 256; i < hi; i++) {
        double uv = u.u[0], ur = u.u[1];
        if (useInputNoise) { uv += sv * N01(eng); ur += sr * N01(eng); }
        Pose2d xk;
        ackerman_.step(xk, poses_[i], uv, ur, dT);
        poses_[i] = xk;
        xb[3 * i] = xk.x[0]; xb[3 * i + 1] = xk.x[1]; xb[3 * i + 2] = xk.x[2];   // the packed copy the next input push sends
      }
    }
    posesDirty_ = true;
    xbufFresh_ = true;
  }

  // RBPHDFilter::update (:444-541); Z is consumed.  One lidar message = one input call + one step call, both stream-ordered
  // (src/rbphdslam_VictoriaPark.cpp:555-583).  The host waits for the device only when the resampling test is due
  // (minUpdatesBeforeResample_ / minMeasurementsBeforeResample_, :528-531), for the 16 bytes the N_eff test needs.
  void update(std::vector<Measurement3d> &Z) {
    nUpdatesSinceResample_++;
    std::vector<Measurement3d> meas;
    meas.swap(Z);
    Z.clear();
    if (meas.empty()) return;  // :450-452
    nMeasurementsSinceResample_ += (unsigned)meas.size();
    pushConfigVP();
    flushMotion();
    pushInputsAsync();
    flushStatic();
    std::vector<double> z(3 * meas.size());
    for (size_t k = 0; k < meas.size(); k++) std::memcpy(&z[3 * k], meas[k].z, 3 * sizeof(double));
    // the step + normalizeWeights in its post launch: what update() does when no resampling happens (:537-539), and what
    // resample() does first when it is attempted (include/ParticleFilter.hpp:402)
    check(rfsgpu_step_async(h_, z.data(), (int)meas.size(), 1), "step");
    weightsStale_ = true;
    resampleOccured_ = false;
    if (nUpdatesSinceResample_ >= (unsigned)config.minUpdatesBeforeResample_ &&
        nMeasurementsSinceResample_ >= (unsigned)config.minMeasurementsBeforeResample_)
      resampleOccured_ = resampleNormalized();
    if (resampleOccured_) {
      nUpdatesSinceResample_ = 0;
      nMeasurementsSinceResample_ = 0;
    }
  }

  // ParticleFilter::resample (:399-492) on weights that the step has already normalised: N_eff = 1 / sum w^2 from the device
  // reduction; no resampling -> the reference divides by the (now ~1) sum once more (RBPHDFilter.hpp:537-539), so does this.
  bool resampleNormalized() {
    double s[2];
    check(rfsgpu_weight_sums(h_, s), "weight_sums");                 // the one host wait of the message
    check(rfsgpu_synchronize(h_), "synchronize");                    // device-side errors of the asynchronous calls
    const double nEff = 1.0 / s[1];
    if (nEff > effNParticles_t_ && nEff / n_ > effNParticles_t_percent_) {
      check(rfsgpu_normalize_weights(h_, s[0], nullptr), "normalize_weights");
      weightsStale_ = true;
      return false;
    }
    return resample(0, true, /*alreadyNormalized=*/true);
  }

  bool getLandmark(int i, int m, double u[3], double S[9], double &w) {
    flushStatic();
    return rfsgpu_get_landmark(h_, i, m, u, S, &w) == RFSGPU_OK;
  }

 private:
  std::vector<double> pendingStatic_;   // the landmark process noise (3 x 3) of every birth-less predict not yet applied on the device
  void flushStatic() {
    if (pendingStatic_.empty()) return;
    check(rfsgpu_static_steps_async(h_, (int)(pendingStatic_.size() / 9), pendingStatic_.data()), "static_steps");
    pendingStatic_.clear();
  }
  MotionModel_Ackerman2d ackerman_;
  LmkProcessModel3d lmk3_;
  MeasurementModelVP measVP_;

  void pushConfigVP() {
    pushFilterConfig();
    rfsgpu_vp_config m;
    std::memset(&m, 0, sizeof(m));
    std::memcpy(m.R, measVP_.R, sizeof(m.R));
    m.Slb = measVP_.Slb;
    m.nPd = (int)measVP_.config.probabilityOfDetection_.size();
    if (m.nPd > RFSGPU_VP_MAX_PD) throw std::runtime_error("probabilityOfDetection_ table longer than RFSGPU_VP_MAX_PD");
    for (int k = 0; k < m.nPd; k++) m.PdTable[k] = measVP_.config.probabilityOfDetection_[k];
    m.expectedClutterNumber = measVP_.config.expectedClutterNumber_;
    m.rangeLimMax = measVP_.config.rangeLimMax_;
    m.rangeLimMin = measVP_.config.rangeLimMin_;
    m.bearingLimitMax = measVP_.config.bearingLimitMax_;
    m.bearingLimitMin = measVP_.config.bearingLimitMin_;
    m.bufferZonePd = measVP_.config.bufferZonePd_;
    check(rfsgpu_set_model_victoriapark(h_, &m), "set_model_victoriapark");
    check(rfsgpu_set_lmk_process_noise(h_, lmk3_.Q), "set_lmk_process_noise");
  }
  // poses and the laser scan of this message -> the device, one stream-ordered call.  (No pose covariances: the Victoria Park
  // measurement model rebuilds the pose from its mean only, src/MeasurementModel_VictoriaPark.cpp:112-114.)
  std::vector<double> xbuf_;
  void pushInputsAsync() {
    const bool scanNow = measVP_.scanDirty && !measVP_.scan.empty();
    if (!posesDirty_ && !scanNow) return;
    if (posesDirty_ && !xbufFresh_) {   // (poses changed by something other than predict(): setParticlePose, resampling)
      xbuf_.resize(3 * (size_t)n_);
      for (int i = 0; i < n_; i++) { xbuf_[3 * i] = poses_[i].x[0]; xbuf_[3 * i + 1] = poses_[i].x[1]; xbuf_[3 * i + 2] = poses_[i].x[2]; }
    }
    xbufFresh_ = false;
    check(rfsgpu_set_step_inputs_async(h_, posesDirty_ ? xbuf_.data() : nullptr, nullptr, 0, scanNow ? measVP_.scan.data() : nullptr,
                                       scanNow ? (int)measVP_.scan.size() : 0), "set_step_inputs");
    posesDirty_ = false;
    measVP_.scanDirty = false;
  }
};

}  // namespace rfs_amd
