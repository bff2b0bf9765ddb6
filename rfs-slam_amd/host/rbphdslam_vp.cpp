// rbphdslam_vp -- the Victoria Park host loop around the device engine: what the reference's rbphdslam_VictoriaPark
// driver does (src/rbphdslam_VictoriaPark.cpp: readData :180-340, filter set-up :344-398, run :440-660), written against
// rfs_amd::RBPHDFilterVP (host/rbphd_filter.hpp -> include/rfsgpu.h).  Reads the reference's own files: the XML
// configuration (cfg/rbphdslam_VictoriaPark*.xml keys), Sensors_manager.txt (t, sensor type, 1-based index), inputs.dat
// (t, speed, steering), measurements.dat (t, range, bearing, diameter).  `Input` messages: predict only; `Lidar` messages:
// predict, optional artificial clutter, setLaserScan, update; logs in the reference's formats (particlePose.dat,
// landmarkEst.dat).  LASER.txt is missing from the reference tree (.MISSING_LARGE_BLOBS), so unless a lidar file is found
// the raw scan is the synthetic one SURVEY 8d prescribes: 361 beams at rangeLimitMax.
//
//   rbphdslam_vp -c cfg.xml [-d dataDir] [-n nParticles] [-m nMessages] [-e effNParticle] [-s seed] [-o outDir] [--device k] [--host-motion] [--repeat K]
//                [--no-input-noise] [--no-clutter]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "rbphd_filter.hpp"
#include "xml_cfg.hpp"

using namespace rfs_amd;

struct ManagerMsg { double t; int type, idx; };   // type 1 GPS, 2 Input, 3 Lidar (Sensors_manager.txt)

template <class Row>
static bool read_rows(const std::string &fn, int nCols, std::vector<Row> &rows, void (*put)(Row &, const double *)) {
  std::ifstream in(fn);
  if (!in) return false;
  std::string line;
  std::vector<double> v(nCols);
  while (std::getline(in, line)) {
    std::istringstream ss(line);
    int k = 0;
    while (k < nCols && (ss >> v[k])) k++;
    if (k < nCols) continue;
    Row r;
    put(r, v.data());
    rows.push_back(r);
  }
  return true;
}

int main(int argc, char **argv) {
  std::string cfgFile, dataDir, outDir;
  int nParticlesOverride = -1, nMsgOverride = -1, seed = 1, device = 0;
  double effNOverride = -1;
  bool inputNoise = true, clutter = true, deviceMotion = true;
  int repeat = 1;
  for (int a = 1; a < argc; a++) {
    std::string s = argv[a];
    auto next = [&]() { return (a + 1 < argc) ? std::string(argv[++a]) : std::string(); };
    if (s == "-c") cfgFile = next();
    else if (s == "-d") dataDir = next();
    else if (s == "-n") nParticlesOverride = std::atoi(next().c_str());
    else if (s == "-m") nMsgOverride = std::atoi(next().c_str());
    else if (s == "-s") seed = std::atoi(next().c_str());
    else if (s == "-e") effNOverride = std::atof(next().c_str());
    else if (s == "-o") outDir = next();
    else if (s == "--device") device = std::atoi(next().c_str());
    else if (s == "--no-input-noise") inputNoise = false;
    else if (s == "--no-clutter") clutter = false;
    else if (s == "--repeat") repeat = std::max(1, std::atoi(next().c_str()));   // the whole run K times in this process (fresh filter each): the first pass carries the GPU's clock ramp
    else if (s == "--host-motion") deviceMotion = false;   // ParticleFilter::propagate on the host (the reference's shape) instead of the device kernel
  }
  Cfg c;
  if (!cfgFile.empty()) c = read_xml_cfg(cfgFile);
  if (dataDir.empty()) dataDir = c.s("config.dataset.directory", "data/VictoriaPark/");
  if (!dataDir.empty() && dataDir.back() != '/') dataDir += '/';
  // defaults = shipped cfg/rbphdslam_VictoriaPark_artificialClutter.xml values
  const double varuv = c.d("config.process.varuv", 0.2), varur = c.d("config.process.varur", 0.025), urScale = c.d("config.process.ur_scale", 1.0);
  const double varlm[3] = {c.d("config.landmarks.varlmx", 5e-4), c.d("config.landmarks.varlmy", 5e-4), c.d("config.landmarks.varlmd", 1e-4)};
  const double rMax = c.d("config.measurements.rangeLimitMax", 70), rMin = c.d("config.measurements.rangeLimitMin", 5);
  const double bMaxDeg = c.d("config.measurements.bearingLimitMax", 177.0), bMinDeg = c.d("config.measurements.bearingLimitMin", 6.3025);
  const double clutterExpected = c.d("config.measurements.expectedNClutter", 6), clutterAdded = clutter ? c.d("config.measurements.addedClutter", 3) : 0;
  const double varz[3] = {c.d("config.measurements.varzr", 0.025), c.d("config.measurements.varzb", 2.5e-5), c.d("config.measurements.varzd", 0.002)};
  const double varza = c.d("config.measurements.varza", 1e-5);
  std::vector<double> pdTable = c.dl("config.measurements.Pd.value");
  if (pdTable.empty()) pdTable = {0.0, 0.05, 0.35, 0.76, 0.89, 0.90};
  int nMsg = c.i("config.filter.nMsgToProcess", 0);
  int nParticles = c.i("config.filter.nParticles", 100);
  const double pNoiseInfl = c.d("config.filter.predict.processNoiseInflationFactor", 20.0), zNoiseInfl = c.d("config.filter.update.measurementNoiseInflationFactor", 40.0);
  if (nParticlesOverride > 0) nParticles = nParticlesOverride;
  if (nMsgOverride > 0) nMsg = nMsgOverride;
  const double PI = std::acos(-1.0);

  // ---------------- data (readData, :180-340) ----------------
  std::vector<ManagerMsg> msgs;
  struct In { double t, v, r; };
  struct Det { double t, z[3]; };
  std::vector<In> inputs;
  std::vector<Det> dets;
  const bool ok = read_rows<ManagerMsg>(dataDir + c.s("config.dataset.filename.manager", "Sensors_manager.txt"), 3, msgs,
                                        [](ManagerMsg &m, const double *v) { m.t = v[0]; m.type = (int)v[1]; m.idx = (int)v[2] - 1; }) &&
                  read_rows<In>(dataDir + c.s("config.dataset.filename.input", "inputs.dat"), 3, inputs, [](In &m, const double *v) { m.t = v[0]; m.v = v[1]; m.r = v[2]; }) &&
                  read_rows<Det>(dataDir + c.s("config.dataset.filename.detection", "measurements.dat"), 4, dets,
                                 [](Det &m, const double *v) { m.t = v[0]; m.z[0] = v[1]; m.z[1] = v[2]; m.z[2] = v[3]; });
  if (!ok || msgs.empty()) { std::fprintf(stderr, "cannot read the dataset under %s\n", dataDir.c_str()); return 2; }
  if (nMsg <= 0 || nMsg > (int)msgs.size()) nMsg = (int)msgs.size();
  const std::vector<double> syntheticScan(361, rMax);   // LASER.txt is missing from the reference tree

  for (int pass = 0; pass < repeat; pass++) {
  if (repeat > 1) { std::printf("---- pass %d of %d ----\n", pass + 1, repeat); if (pass > 0) outDir.clear(); }
  // ---------------- filter set-up (:344-398) ----------------
  RBPHDFilterVP filter(nParticles, device, 192);
  filter.getProcessModel()->setAckermanParams(c.d("config.process.AckermanModel.rearWheelOffset", 0.76), c.d("config.process.AckermanModel.frontToRearDist", 2.83),
                                              c.d("config.process.AckermanModel.sensorOffset_x", 3.78), c.d("config.process.AckermanModel.sensorOffset_y", 0.50));
  filter.setDeviceMotion(deviceMotion, 0x9E3779B97F4A7C15ull * (unsigned long long)(seed + 1));
  double R[9] = {0};
  for (int k = 0; k < 3; k++) R[4 * k] = varz[k] * zNoiseInfl;
  filter.getMeasurementModel()->setNoise(R, varza);
  auto &mc = filter.getMeasurementModel()->config;
  mc.probabilityOfDetection_ = pdTable;
  mc.expectedClutterNumber_ = clutterExpected;
  mc.rangeLimMax_ = rMax;
  mc.rangeLimMin_ = rMin;
  mc.bearingLimitMax_ = bMaxDeg * PI / 180;
  mc.bearingLimitMin_ = bMinDeg * PI / 180;
  mc.bufferZonePd_ = c.d("config.measurements.bufferZonePd", 0.4);
  filter.getKalmanFilter()->config.rangeInnovationThreshold_ = c.d("config.filter.update.KalmanFilter.innovationThreshold.range", 7.5);
  filter.getKalmanFilter()->config.bearingInnovationThreshold_ = c.d("config.filter.update.KalmanFilter.innovationThreshold.bearing", 0.2);
  filter.config.birthGaussianWeight_ = c.d("config.filter.predict.birthGaussian.Weight", 0.01);
  filter.config.birthGaussianMeasurementSupportDist_ = c.d("config.filter.predict.birthGaussian.SupportMeasurementDist", 2);
  filter.config.birthGaussianMeasurementCountThreshold_ = (unsigned)c.i("config.filter.predict.birthGaussian.SupportMeasurementThreshold", 5);
  filter.config.birthGaussianMeasurementCheckThreshold_ = (unsigned)c.i("config.filter.predict.birthGaussian.CheckCountThreshold", 10);
  filter.config.birthGaussianCurrentMeasurementCountThreshold_ = (unsigned)c.d("config.filter.predict.birthGaussian.CurrentMeasurementCountThreshold", 2);
  filter.config.newGaussianCreateInnovMDThreshold_ = c.d("config.filter.update.GaussianCreateInnovMDThreshold", 3.0);
  filter.config.importanceWeightingEvalPointCount_ = c.i("config.filter.weighting.nEvalPt", 15);
  filter.config.importanceWeightingEvalPointGuassianWeight_ = c.d("config.filter.weighting.minWeight", 0.75);
  filter.config.importanceWeightingMeasurementLikelihoodMDThreshold_ = c.d("config.filter.weighting.threshold", 3.0);
  filter.config.useClusterProcess_ = c.i("config.filter.weighting.useClusterProcess", 0) == 1;
  filter.config.minUpdatesBeforeResample_ = c.i("config.filter.resampling.minTimesteps", 2);
  filter.config.minMeasurementsBeforeResample_ = c.i("config.filter.resampling.minMeasurements", 15);
  filter.config.gaussianMergingThreshold_ = c.d("config.filter.merge.threshold", 1.0);
  filter.config.gaussianMergingCovarianceInflationFactor_ = c.d("config.filter.merge.covInflationFactor", 1.5);
  filter.config.gaussianPruningThreshold_ = c.d("config.filter.prune.threshold", filter.config.birthGaussianWeight_);
  filter.setEffectiveParticleCountThreshold(effNOverride > 0 ? effNOverride : c.d("config.filter.resampling.effNParticle", (double)nParticles));
  filter.getMeasurementModel()->setLaserScan(syntheticScan);

  FILE *fPose = nullptr, *fLm = nullptr;
  if (!outDir.empty()) {
    if (outDir.back() != '/') outDir += '/';
    fPose = std::fopen((outDir + "particlePose.dat").c_str(), "w");
    fLm = std::fopen((outDir + "landmarkEst.dat").c_str(), "w");
  }

  // ---------------- run (:440-660) ----------------
  std::mt19937 rng((unsigned)seed);
  srand48(seed);
  std::uniform_real_distribution<double> U01(0.0, 1.0);
  std::poisson_distribution<int> nClutter(clutterAdded > 0 ? clutterAdded : 1.0);
  AckermanInput u_km;
  u_km.var[0] = varuv * pNoiseInfl;
  u_km.var[1] = varur * pNoiseInfl;
  double t_km = 0;
  bool stationary = true, birthCheck = true;
  size_t zIdx = 0;
  int nLidar = 0, nResample = 0;
  double tPredict = 0, tUpdate = 0;   // host wall time inside the two filter calls
  auto now = [] { return std::chrono::steady_clock::now(); };
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < nMsg; k++) {
    const ManagerMsg &m = msgs[k];
    if (m.type != 2 && m.type != 3) continue;
    const double dt = m.t - t_km;
    const double Qlm[9] = {varlm[0] * dt * dt, 0, 0, 0, varlm[1] * dt * dt, 0, 0, 0, varlm[2] * dt * dt};
    filter.getLmkProcessModel()->setNoise(Qlm);
    { const auto ta = now(); filter.predict(u_km, dt, false, !stationary && inputNoise, birthCheck);   // (stationary: all particles sit still)
      tPredict += std::chrono::duration<double>(now() - ta).count(); }
    birthCheck = false;
    if (m.type == 2) {
      if (m.idx >= 0 && m.idx < (int)inputs.size()) { u_km.u[0] = inputs[m.idx].v; u_km.u[1] = inputs[m.idx].r * urScale; }
      if (u_km.u[0] != 0) stationary = false;
    } else {
      std::vector<Measurement3d> Z;
      while (zIdx < dets.size() && std::fabs(dets[zIdx].t - m.t) < 1e-9) {
        Measurement3d z;
        z.t = m.t;
        z.z[0] = dets[zIdx].z[0]; z.z[1] = dets[zIdx].z[1]; z.z[2] = dets[zIdx].z[2];
        Z.push_back(z);
        zIdx++;
      }
      if (clutterAdded > 0) {
        const int nc = nClutter(rng);
        for (int q = 0; q < nc; q++) {
          Measurement3d z;
          z.t = m.t;
          z.z[0] = U01(rng) * (rMax - rMin) + rMin;
          z.z[1] = U01(rng) * ((bMaxDeg - bMinDeg) + bMinDeg) * PI / 180;  // (sic, :563)
          z.z[2] = 1.0;
          Z.push_back(z);
        }
      }
      if (Z.size() > RFSGPU_MAX_Z) Z.resize(RFSGPU_MAX_Z);
      filter.getMeasurementModel()->setLaserScan(syntheticScan);
      const bool any = !Z.empty();
      { const auto ta = now(); filter.update(Z); tUpdate += std::chrono::duration<double>(now() - ta).count(); }
      if (any) { nLidar++; nResample += filter.resampleOccured() ? 1 : 0; }
      birthCheck = true;
      if (fPose || fLm) {
        int best = 0;
        double bw = 0;
        for (int i = 0; i < filter.getParticleCount(); i++) {
          const double w = filter.getParticleWeight(i);
          const Pose2d &x = filter.getParticlePose(i);
          if (fPose) std::fprintf(fPose, "%10.3f%5d%10.3f%10.3f%10.3f%10.3f\n", m.t, i, x.x[0], x.x[1], x.x[2], w);
          if (w > bw) { bw = w; best = i; }
        }
        if (fLm)
          for (int g = 0; g < filter.getGMSize(best); g++) {
            double mu[3], S[9], w;
            filter.getLandmark(best, g, mu, S, w);
            std::fprintf(fLm, "%10.3f%5d%10.3f%10.3f%10.3f%10.3f%10.3f%10.3f\n", m.t, best, mu[0], mu[1], S[0], S[1], S[4], w);
          }
      }
    }
    t_km = m.t;
  }
  const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (fPose) std::fclose(fPose);
  if (fLm) std::fclose(fLm);

  int best = 0;
  double bw = -1;
  for (int i = 0; i < filter.getParticleCount(); i++) { const double w = filter.getParticleWeight(i); if (w > bw) { bw = w; best = i; } }
  const int nMap = filter.getGMSize(best);
  if (!outDir.empty()) {   // the best particle's final map at full precision (for comparisons between host loops)
    FILE *fm = std::fopen((outDir + "finalMap.dat").c_str(), "w");
    if (fm) {
      const Pose2d &x = filter.getParticlePose(best);
      std::fprintf(fm, "# pose %.17g %.17g %.17g\n", x.x[0], x.x[1], x.x[2]);
      for (int g = 0; g < nMap; g++) {
        double mu[3], S[9], w;
        filter.getLandmark(best, g, mu, S, w);
        std::fprintf(fm, "%.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", w, mu[0], mu[1], mu[2], S[0], S[1], S[2], S[4], S[5], S[8]);
      }
      std::fclose(fm);
    }
  }
  int strong = 0;
  for (int g = 0; g < nMap; g++) { double mu[3], S[9], w; filter.getLandmark(best, g, mu, S, w); strong += (w >= 0.5) ? 1 : 0; }
  const Pose2d &bp = filter.getParticlePose(best);
  std::printf("particles %d  messages %d  lidar updates %d  resamplings %d  wall %.3f s  (inside predict() %.3f s, inside update() %.3f s)  propagation on the %s\n", nParticles, nMsg, nLidar,
              nResample, wall, tPredict, tUpdate, deviceMotion ? "device" : "host");
  std::printf("predict() host sections [s]: config %.4f  inputs %.4f  predict_map launch %.4f  propagate %.4f\n", filter.tCfg_, filter.tIn_, filter.tPm_, filter.tProp_);
  RBPHDFilter2d::TimingInfo *ti = filter.getTimingInfo();
  std::printf("Elapsed Timing Information [nsec]\n");  // format of the reference drivers' timing printout
  std::printf("%-22s%15s%15s\n", "", "wall", "cpu");
  std::printf("%-22s%15lld%15lld\n", "Prediction", ti->predict_wall, ti->predict_cpu);
  std::printf("%-22s%15lld%15lld\n", "Map Update", ti->mapUpdate_wall, ti->mapUpdate_cpu);
  std::printf("%-22s%15lld%15lld\n", "Weighting", ti->particleWeighting_wall, ti->particleWeighting_cpu);
  std::printf("%-22s%15lld%15lld\n", "Map Merge", ti->mapMerge_wall, ti->mapMerge_cpu);
  std::printf("%-22s%15lld%15lld\n", "Map Prune", ti->mapPrune_wall, ti->mapPrune_cpu);
  std::printf("%-22s%15lld%15lld\n", "Resampling", ti->particleResample_wall, ti->particleResample_cpu);
  std::printf("RESULT lidar=%d resamples=%d best=%d map=%d strong=%d pose=%.6f,%.6f,%.6f ms_per_update=%.4f\n", nLidar, nResample, best, nMap, strong, bp.x[0], bp.x[1],
              bp.x[2], nLidar ? wall * 1e3 / nLidar : 0.0);
  }
  return 0;
}
