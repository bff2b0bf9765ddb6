// rbphdslam2d_sim.cpp -- minimal 2-D RB-PHD SLAM simulator driving the device path through the C++ host mirror
// (rbphd_filter.hpp).  Reproduces what the reference driver feeds its filter (reference src/rbphdslam2dSim.cpp):
// same XML keys (cfg/rbphdslam2dSim.xml), same data generation procedure (:150-366: drand48-driven trajectory
// segments, landmarks by inverse measurement of random range/bearing along the trajectory, noisy odometry, Pd-thinned
// detections + Poisson clutter), same filter setup (:444-492), same run loop (:540-643: predict, k<=100 pose reset to
// ground truth, gather Z by timestamp, update), same log formats (particlePose.dat `t i x y th w`, landmarkEst.dat
// `t i mx my Sxx Sxy Syy w` :620,635-638) and the same TimingInfo printout (:654-690).
// Gaussian noise comes from std::mt19937 (the reference uses boost::mt19937 + boost::normal_distribution, whose sample
// stream cannot be matched without Boost); drand48() is used where the reference uses it.
//
//   rbphdslam2d_sim [-c cfg.xml] [-t trajSeed] [-s simSeed] [-n nParticlesOverride] [-k kMaxOverride] [-o outDir]
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <map>
#include <random>
#include <sstream>
#include <memory>
#include <string>
#include <vector>

#include "rbphd_filter.hpp"
#include "xml_cfg.hpp"

using namespace rfs_amd;

static const double PI = std::acos(-1.0);

struct Landmark { double x[2]; };

int main(int argc, char **argv) {
  std::string cfgFile, outDir;
  int trajSeed = 1, simSeed = 1, nParticlesOverride = -1, kMaxOverride = -1, device = 0;
  std::vector<int> devices;
  for (int a = 1; a < argc; a++) {
    std::string s = argv[a];
    auto next = [&]() { return (a + 1 < argc) ? std::string(argv[++a]) : std::string(); };
    if (s == "-c") cfgFile = next();
    else if (s == "-t") trajSeed = std::atoi(next().c_str());
    else if (s == "-s") simSeed = std::atoi(next().c_str());
    else if (s == "-n") nParticlesOverride = std::atoi(next().c_str());
    else if (s == "-k") kMaxOverride = std::atoi(next().c_str());
    else if (s == "-o") outDir = next();
    else if (s == "-d") device = std::atoi(next().c_str());
    else if (s == "--devices") {   // e.g. "0,1,2,3" (ids may repeat): one shard of the particle set per entry, rfsgpu_group_*
      std::string list = next();
      size_t pos = 0;
      while (pos <= list.size()) {
        const size_t comma = list.find(',', pos);
        const std::string tok = list.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
        if (!tok.empty()) devices.push_back(std::atoi(tok.c_str()));
        if (comma == std::string::npos) break;
        pos = comma + 1;
      }
    }
  }
  Cfg c;
  if (!cfgFile.empty()) c = read_xml_cfg(cfgFile);
  // defaults = shipped cfg/rbphdslam2dSim.xml values
  int kMax = c.i("config.timesteps", 3000);
  const double dT = c.d("config.sec_per_timestep", 0.1);
  const int nSegments = c.i("config.trajectory.nSegments", 20);
  const double max_dx = c.d("config.trajectory.max_dx_per_sec", 0.30), max_dy = c.d("config.trajectory.max_dy_per_sec", 0.0),
               max_dz = c.d("config.trajectory.max_dz_per_sec", 0.50), min_dx = c.d("config.trajectory.min_dx_per_sec", 0.10);
  const double vardx = c.d("config.trajectory.vardx", 0.002), vardy = c.d("config.trajectory.vardy", 0.002), vardz = c.d("config.trajectory.vardz", 0.002);
  const int nLandmarks = c.i("config.landmarks.nLandmarks", 50);
  const double varlmx = c.d("config.landmarks.varlmx", 0.0002), varlmy = c.d("config.landmarks.varlmy", 0.0002);
  const double rMax = c.d("config.measurements.rangeLimitMax", 2.5), rMin = c.d("config.measurements.rangeLimitMin", 0.5),
               rBuf = c.d("config.measurements.rangeLimitBuffer", 0.05), Pd = c.d("config.measurements.probDetection", 0.99),
               clutter = c.d("config.measurements.clutterIntensity", 1e-4), varzr = c.d("config.measurements.varzr", 5e-4),
               varzb = c.d("config.measurements.varzb", 5e-5);
  int nParticles = c.i("config.filter.nParticles", 200);
  const double pNoiseInfl = c.d("config.filter.predict.processNoiseInflationFactor", 1.5), birthW = c.d("config.filter.predict.birthGaussianWeight", 0.01);
  const double zNoiseInfl = c.d("config.filter.update.measurementNoiseInflationFactor", 10.0);
  const double innovR = c.d("config.filter.update.KalmanFilter.innovationThreshold.range", 1.0),
               innovB = c.d("config.filter.update.KalmanFilter.innovationThreshold.bearing", 0.2);
  const double newGaussMD = c.d("config.filter.update.GaussianCreateInnovMDThreshold", 3.0);
  const int nEvalPt = c.i("config.filter.weighting.nEvalPt", 15);
  const double minWeight = c.d("config.filter.weighting.minWeight", 0.75), weightThr = c.d("config.filter.weighting.threshold", 3.0);
  const bool useCluster = c.i("config.filter.weighting.useClusterProcess", 0) == 1;
  const double effN = c.d("config.filter.resampling.effNParticle", 100.0);
  const int minSteps = c.i("config.filter.resampling.minTimesteps", 2);
  const double mergeThr = c.d("config.filter.merge.threshold", 0.5), mergeInfl = c.d("config.filter.merge.covInflationFactor", 1.5);
  const double pruneThr = c.d("config.filter.prune.threshold", 0.01);
  if (nParticlesOverride > 0) nParticles = nParticlesOverride;
  if (kMaxOverride > 0) kMax = kMaxOverride;

  // ---------------- data generation (:150-366) ----------------
  std::mt19937 gen(simSeed * 7919 + 13);
  std::normal_distribution<double> N01(0.0, 1.0);
  srand48(trajSeed);
  std::vector<Odometry2d> gtDisp(kMax), odom(kMax);
  std::vector<Pose2d> gtPose(kMax);
  {
    int seg = 0;
    Odometry2d in;
    for (int k = 1; k < kMax; k++) {
      if (k <= 50) {
        in = Odometry2d();
      } else if (k >= kMax / nSegments * seg) {
        seg++;
        double dx = drand48() * max_dx * dT;
        while (dx < min_dx * dT) dx = drand48() * max_dx * dT;
        double dy = (drand48() * max_dy * 2 - max_dy) * dT;
        double dz = (drand48() * max_dz * 2 - max_dz) * dT;
        in.u[0] = dx; in.u[1] = dy; in.u[2] = dz;
      }
      in.t = k * dT;
      gtDisp[k] = in;
      MotionModel_Odometry2d::step(gtPose[k], gtPose[k - 1], in);
    }
  }
  srand48(simSeed);
  std::vector<Landmark> gtLm;
  {
    int created = 0;
    for (int k = 1; k < kMax; k++)
      if (k >= kMax / nLandmarks * created) {
        const double r = drand48() * rMax, b = drand48() * 2 * PI;
        Landmark lm;
        lm.x[0] = gtPose[k].x[0] + r * std::cos(gtPose[k].x[2] + b);
        lm.x[1] = gtPose[k].x[1] + r * std::sin(gtPose[k].x[2] + b);
        gtLm.push_back(lm);
        created++;
      }
  }
  for (int k = 1; k < kMax; k++) {  // odometry = displacement + N(0, Q dt^2)
    odom[k] = gtDisp[k];
    odom[k].u[0] += std::sqrt(vardx) * dT * N01(gen);
    odom[k].u[1] += std::sqrt(vardy) * dT * N01(gen);
    odom[k].u[2] += std::sqrt(vardz) * dT * N01(gen);
  }
  std::vector<Measurement2d> measurements;
  std::vector<double> lmkFirstObs(gtLm.size(), -1.0);
  {
    const double meanClutter = clutter * 2 * PI * (rMax - rMin);
    double cmf[100], pmf = std::exp(-meanClutter), mp = 1, fact = 1;
    cmf[0] = pmf;
    for (int i = 1; i < 100; i++) { mp *= meanClutter; fact *= i; cmf[i] = cmf[i - 1] + mp / fact * std::exp(-meanClutter); }
    for (int k = 1; k < kMax; k++) {
      const double t = k * dT;
      for (size_t li = 0; li < gtLm.size(); li++) {
        const Landmark &lm = gtLm[li];
        const double dx = lm.x[0] - gtPose[k].x[0], dy = lm.x[1] - gtPose[k].x[1];
        double r = std::sqrt(dx * dx + dy * dy);
        double b = std::atan2(dy, dx) - gtPose[k].x[2];
        const bool success = !(r > rMax || r < rMin);  // MeasurementModel::sample returns measure()'s flag
        r += std::sqrt(varzr) * N01(gen);
        b += std::sqrt(varzb) * N01(gen);
        while (b > PI) b -= 2 * PI;
        while (b < -PI) b += 2 * PI;
        if (success) {
          if (r <= rMax && r >= rMin && drand48() <= Pd) { Measurement2d z; z.z[0] = r; z.z[1] = b; z.t = t; measurements.push_back(z); }
          if (lmkFirstObs[li] < 0) lmkFirstObs[li] = t;   // lmkFirstObsTime_: first time in sensor range, detected or not (src/rbphdslam2dSim.cpp:337-339)
        }
      }
      const double u = drand48();
      int nC = 0;
      while (nC < 99 && u > cmf[nC]) nC++;
      for (int i = 0; i < nC; i++) {
        double r = drand48() * rMax;
        while (r < rMin) r = drand48() * rMax;
        Measurement2d z; z.z[0] = r; z.z[1] = drand48() * 2 * PI - PI; z.t = t;
        measurements.push_back(z);
      }
    }
  }

  // ---------------- filter setup (:444-492) ----------------
#ifdef USE_FASTSLAM
  // fastslam2dSim (reference src/fastslam2dSim.cpp:444-482): same simulator, the FastSLAM filter class and its keys
  FastSLAM2d filter(nParticles, device, 384, (unsigned)c.i("config.filter.update.maxNDataAssocHypotheses", 1));
  filter.config.minUpdatesBeforeResample_ = c.i("config.filter.resampling.minTimesteps", 1);
  filter.config.minLogMeasurementLikelihood_ = c.d("config.filter.weighting.minLogMeasurementLikelihood", -10.0);
  filter.config.maxNDataAssocHypotheses_ = (unsigned)c.i("config.filter.update.maxNDataAssocHypotheses", 1);
  filter.config.maxDataAssocLogLikelihoodDiff_ = c.d("config.filter.update.maxDataAssocLogLikelihoodDiff", 3.0);
  filter.config.mapExistencePruneThreshold_ = c.d("config.filter.prune.threshold", -5.0);
  filter.config.landmarkExistencePrior_ = 0.5;
  (void)birthW; (void)newGaussMD; (void)nEvalPt; (void)minWeight; (void)weightThr; (void)useCluster; (void)minSteps; (void)mergeThr; (void)mergeInfl; (void)pruneThr;
#else
  std::unique_ptr<RBPHDFilter2d> filterOwner(devices.size() > 1 ? new RBPHDFilter2d(nParticles, devices, 384) : new RBPHDFilter2d(nParticles, device, 384));
  RBPHDFilter2d &filter = *filterOwner;
  if (devices.size() > 1) std::printf("particle set sharded over %zu device entries (rfsgpu_group_*)\n", devices.size());
#endif
  double Q[9] = {vardx, 0, 0, 0, vardy, 0, 0, 0, vardz};
  for (double &q : Q) q *= pNoiseInfl * dT * dT;
  filter.getProcessModel()->setNoise(Q);
  double Qlm[4] = {varlmx * dT * dT, 0, 0, varlmy * dT * dT};
  filter.getLmkProcessModel()->setNoise(Qlm);
  double R[4] = {varzr * zNoiseInfl, 0, 0, varzb * zNoiseInfl};
  filter.getMeasurementModel()->setNoise(R);
  filter.getMeasurementModel()->config.probabilityOfDetection_ = Pd;
  filter.getMeasurementModel()->config.uniformClutterIntensity_ = clutter;
  filter.getMeasurementModel()->config.rangeLimMax_ = rMax;
  filter.getMeasurementModel()->config.rangeLimMin_ = rMin;
  filter.getMeasurementModel()->config.rangeLimBuffer_ = rBuf;
  filter.getKalmanFilter()->config.rangeInnovationThreshold_ = innovR;
  filter.getKalmanFilter()->config.bearingInnovationThreshold_ = innovB;
  filter.setEffectiveParticleCountThreshold(effN);
#ifndef USE_FASTSLAM
  filter.config.birthGaussianWeight_ = birthW;
  filter.config.minUpdatesBeforeResample_ = minSteps;
  filter.config.newGaussianCreateInnovMDThreshold_ = newGaussMD;
  filter.config.importanceWeightingMeasurementLikelihoodMDThreshold_ = weightThr;
  filter.config.importanceWeightingEvalPointCount_ = nEvalPt;
  filter.config.importanceWeightingEvalPointGuassianWeight_ = minWeight;
  filter.config.gaussianMergingThreshold_ = mergeThr;
  filter.config.gaussianMergingCovarianceInflationFactor_ = mergeInfl;
  filter.config.gaussianPruningThreshold_ = pruneThr;
  filter.config.useClusterProcess_ = useCluster;
#endif

  // ---------------- run (:540-643) ----------------
  FILE *fPose = nullptr, *fLm = nullptr;
  if (!outDir.empty()) {
    fPose = std::fopen((outDir + "/particlePose.dat").c_str(), "w");
    fLm = std::fopen((outDir + "/landmarkEst.dat").c_str(), "w");
    // exportSimData (src/rbphdslam2dSim.cpp:380-440): the files analysis2dSim reads (tools/analysis2d_sim.py here)
    if (FILE *fg = std::fopen((outDir + "/gtLandmark.dat").c_str(), "w")) {   // x y (time first observed; -1: never)
      for (size_t li = 0; li < gtLm.size(); li++) std::fprintf(fg, "%f   %f   %f\n", gtLm[li].x[0], gtLm[li].x[1], lmkFirstObs[li]);
      std::fclose(fg);
    }
    if (FILE *fg = std::fopen((outDir + "/gtPose.dat").c_str(), "w")) {
      for (int k = 0; k < kMax; k++) std::fprintf(fg, "%f   %f   %f   %f\n", k * dT, gtPose[k].x[0], gtPose[k].x[1], gtPose[k].x[2]);
      std::fclose(fg);
    }
    if (FILE *fg = std::fopen((outDir + "/odometry.dat").c_str(), "w")) {
      for (int k = 0; k < kMax; k++) std::fprintf(fg, "%f   %f   %f   %f\n", k * dT, odom[k].u[0], odom[k].u[1], odom[k].u[2]);
      std::fclose(fg);
    }
    if (FILE *fg = std::fopen((outDir + "/measurement.dat").c_str(), "w")) {
      for (const Measurement2d &z : measurements) std::fprintf(fg, "%f   %f   %f\n", z.t, z.z[0], z.z[1]);
      std::fclose(fg);
    }
    if (FILE *fg = std::fopen((outDir + "/deadReckoning.dat").c_str(), "w")) {   // the noisy odometry integrated from the origin (:322-330)
      Pose2d dr;
      std::fprintf(fg, "%f   %f   %f   %f\n", 0.0, dr.x[0], dr.x[1], dr.x[2]);
      for (int k = 1; k < kMax; k++) {
        Pose2d nx;
        MotionModel_Odometry2d::step(nx, dr, odom[k]);
        dr = nx;
        std::fprintf(fg, "%f   %f   %f   %f\n", k * dT, dr.x[0], dr.x[1], dr.x[2]);
      }
      std::fclose(fg);
    }
  }
  srand48(simSeed);
  size_t zIdx = 0;
  int nUpdates = 0;
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 1; k < kMax; k++) {
    const double time = k * dT;
    filter.predict(odom[k], dT);
    if (k <= 100)
      for (int i = 0; i < filter.getParticleCount(); i++) { Pose2d p = gtPose[k]; filter.setParticlePose(i, p); }
    std::vector<Measurement2d> Z;
    while (zIdx < measurements.size() && std::fabs(measurements[zIdx].t - time) < 1e-9) Z.push_back(measurements[zIdx++]);
    if (!Z.empty()) nUpdates++;
    filter.update(Z);
    if (fPose || fLm) {
      int best = 0;
      double bw = -1;
      for (int i = 0; i < filter.getParticleCount(); i++) {  // the count varies under MH-FastSLAM
        const double w = filter.getParticleWeight(i);
        const Pose2d &x = filter.getParticlePose(i);
        if (fPose) std::fprintf(fPose, "%f   %d   %f   %f   %f   %f\n", time, i, x.x[0], x.x[1], x.x[2], w);
        if (w > bw) { bw = w; best = i; }
      }
      if (fLm) {
        const int n = filter.getGMSize(best);
        for (int m = 0; m < n; m++) {
          double u[2], S[4], w;
          filter.getLandmark(best, m, u, S, w);
          std::fprintf(fLm, "%f   %d   %f   %f   %f   %f   %f   %f\n", time, best, u[0], u[1], S[0], S[1], S[3], w);
        }
      }
    }
  }
  const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (fPose) std::fclose(fPose);
  if (fLm) std::fclose(fLm);

  // ---------------- result summary: map error of the best particle ----------------
  int best = 0;
  double bw = -1;
  for (int i = 0; i < filter.getParticleCount(); i++) { const double w = filter.getParticleWeight(i); if (w > bw) { bw = w; best = i; } }
  const int n = filter.getGMSize(best);
  int matched = 0, strong = 0;
  double errSum = 0;
  std::vector<char> taken(gtLm.size(), 0);
  for (int m = 0; m < n; m++) {
    double u[2], S[4], w;
    filter.getLandmark(best, m, u, S, w);
    if (w < 0.5) continue;
    strong++;
    double bd = 1e9; int bj = -1;
    for (size_t j = 0; j < gtLm.size(); j++) {
      if (taken[j]) continue;
      const double d = std::hypot(u[0] - gtLm[j].x[0], u[1] - gtLm[j].x[1]);
      if (d < bd) { bd = d; bj = (int)j; }
    }
    if (bj >= 0 && bd < 0.5) { taken[bj] = 1; matched++; errSum += bd; }
  }
  const Pose2d &bp = filter.getParticlePose(best);
  const double poseErr = std::hypot(bp.x[0] - gtPose[kMax - 1].x[0], bp.x[1] - gtPose[kMax - 1].x[1]);
  RBPHDFilter2d::TimingInfo *ti = filter.getTimingInfo();
  std::printf("particles %d  steps %d  updates %d  wall %.3f s\n", nParticles, kMax - 1, nUpdates, wall);
  std::printf("best particle: %d Gaussians (%d with w>=0.5), %d of %zu landmarks matched, mean error %.4f m, final pose error %.4f m\n", n, strong,
              matched, gtLm.size(), matched ? errSum / matched : -1.0, poseErr);
  std::printf("Elapsed Timing Information [nsec]\n");  // format of src/rbphdslam2dSim.cpp:654-690
  std::printf("%-22s%15s%15s\n", "", "wall", "cpu");
  std::printf("%-22s%15lld%15lld\n", "Prediction", ti->predict_wall, ti->predict_cpu);
  std::printf("%-22s%15lld%15lld\n", "Map Update", ti->mapUpdate_wall, ti->mapUpdate_cpu);
  std::printf("%-22s%15lld%15lld\n", "Map Update (KF)", ti->mapUpdate_kf_wall, ti->mapUpdate_kf_cpu);
  std::printf("%-22s%15lld%15lld\n", "Weighting", ti->particleWeighting_wall, ti->particleWeighting_cpu);
  std::printf("%-22s%15lld%15lld\n", "Map Merge", ti->mapMerge_wall, ti->mapMerge_cpu);
  std::printf("%-22s%15lld%15lld\n", "Map Prune", ti->mapPrune_wall, ti->mapPrune_cpu);
  std::printf("%-22s%15lld%15lld\n", "Resampling", ti->particleResample_wall, ti->particleResample_cpu);
  std::printf("RESULT matched=%d landmarks=%zu mean_err=%.6f pose_err=%.6f ms_per_update=%.4f\n", matched, gtLm.size(), matched ? errSum / matched : -1.0,
              poseErr, nUpdates ? 1e-6 * (ti->mapUpdate_wall + ti->particleWeighting_wall + ti->mapMerge_wall + ti->mapPrune_wall) / nUpdates : 0.0);
  return 0;
}
