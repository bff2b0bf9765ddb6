// group.h -- rfsgpu_group_*: ONE filter over SEVERAL GPUs from a single host thread ("rfsgpu_create(model, n,
// device_ids[], n_dev, ...)", 8(e)).  The particle set of rfs::RBPHDFilter (include/RBPHDFilter.hpp:72-251) is cut into
// contiguous blocks, one rfsgpu_filter shard per listed device; every phase of update() before resampling is independent per
// particle (:469-520) and runs on all shards concurrently (stream-ordered fused steps); the shards meet
//   * in normalizeWeights / N_eff (include/ParticleFilter.hpp:352-363, 405-415): each shard's post kernel leaves {sum w, sum w^2};
//     over distinct devices the pairs are ALL-REDUCED over RCCL / xGMI (one communicator per shard from ncclCommInitAll, the
//     collective ordered on each shard's stream) and every shard divides by the device-resident total -- no host round trip, the
//     16 bytes come to the host only when the caller asks for them (the N_eff test); with repeated device ids (several shards on
//     one GPU: the single-GPU tests) or without librccl the pairs are added on the host in shard order instead, and
//   * in resample (:399-492), which keeps the reference's GLOBAL systematic resampling: one plan over all N weights, local
//     children by a device gather, cross-device children as packed rows (rfsgpu_export_slab_rows) moved with
//     hipMemcpyPeerAsync over xGMI and unpacked on the destination (rfsgpu_import_slab_rows).
// (The multi-process form of the same logic, over RCCL, is rfs-slam_amd/sharded.py.)  Included at the end of rfsgpu_engine.hip.
#pragma once

struct rfsgpu_group {
  std::vector<rfsgpu_filter *> shard;
  std::vector<int> first;      // first global particle index of every shard (+ total at the end)
  int N = 0;
  std::vector<unsigned char *> sendBuf, recvBuf;   // per shard, device memory on its device, grown on demand
  std::vector<size_t> sendCap, recvCap;
  std::vector<hipEvent_t> evExport;
  std::vector<int> lastPlan;
  long long rowsMigrated = 0, bytesMigrated = 0;
  // birth-state inheritance after a resampling (include/RBPHDFilter.hpp:1005-1011): ids are GLOBAL slot numbers, so the group owns
  // the rule and its shards run in RFSGPU_INHERIT_EXTERNAL (RFSGPU_INHERIT_EAGER: the shards copy eagerly, the group does nothing)
  int inheritMode = RFSGPU_INHERIT_REFERENCE;
  std::vector<int> pid, ppid;
  bool resampleOccured = false;
  // RCCL (single process, one communicator per shard): loaded lazily with dlopen, so that the library itself links libamdhip64 only
  void *rcclLib = nullptr;
  std::vector<void *> comm;            // ncclComm_t per shard (empty: host path)
  std::vector<double *> dTot;          // per shard, on its device: the all-reduced {sum w, sum w^2}
  std::string rcclNote;                // why the host path is in use, if it is
  // trailing normalisation (rfsgpu_group_update_deferred): the all-reduce of step k on a side stream per shard, beside step k + 1's kernel
  std::vector<hipStream_t> side;
  int handover = 0;            // 0 not probed yet | 1 device sequence numbers | 2 stream events (rfsgpu_group_update_deferred)
  std::vector<hipEvent_t> evPost, evTot;
  bool pendingTotal = false;           // dTot holds a total the weights have not been divided by yet
  std::string err;
};

// ---- RCCL through dlopen: the six entry points the group needs (rccl.h: ncclCommInitAll, ncclCommDestroy, ncclAllReduce,
// ncclGroupStart / End, ncclGetErrorString); ncclDouble = 8, ncclSum = 0, ncclSuccess = 0 ---------------------------------------
#include <dlfcn.h>
struct RcclApi {
  int (*CommInitAll)(void **comm, int ndev, const int *devlist) = nullptr;
  int (*CommDestroy)(void *comm) = nullptr;
  int (*AllReduce)(const void *sendbuff, void *recvbuff, size_t count, int datatype, int op, void *comm, hipStream_t stream) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
static RcclApi g_rccl;
static void *rccl_load(std::string &why) {
  // RFSGPU_RCCL_LIB names the library file instead of the three default names (an installation elsewhere; the test of this very fall-back)
  const char *named = getenv("RFSGPU_RCCL_LIB");
  void *h = (named && *named) ? dlopen(named, RTLD_NOW | RTLD_LOCAL) : dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h && !(named && *named)) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!h && !(named && *named)) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) { const char *e = dlerror(); why = std::string("librccl not loadable: ") + (e ? e : "?"); return nullptr; }   // (dlerror() clears itself: one call)
  g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))dlsym(h, "ncclCommInitAll");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
  g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
  g_rccl.GroupStart = (decltype(g_rccl.GroupStart))dlsym(h, "ncclGroupStart");
  g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))dlsym(h, "ncclGroupEnd");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!g_rccl.CommInitAll || !g_rccl.CommDestroy || !g_rccl.AllReduce || !g_rccl.GroupStart || !g_rccl.GroupEnd) { why = "librccl lacks an expected symbol"; dlclose(h); return nullptr; }
  return h;
}

static int gfail(rfsgpu_group *g, int code, const std::string &msg) {
  g->err = msg;
  return code;
}
#define GCHK(call)                                                                      \
  do {                                                                                  \
    hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess) return gfail(g, RFSGPU_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)
#define GFWD(k, call)                                                                   \
  do {                                                                                  \
    int rc_ = (call);                                                                   \
    if (rc_ != RFSGPU_OK) return gfail(g, rc_, std::string("shard ") + std::to_string(k) + ": " + rfsgpu_last_error(g->shard[k])); \
  } while (0)

// ParticleFilter::resample's systematic sampling + slot assignment (include/ParticleFilter.hpp:419-479), given the uniform draw.
static void systematic_plan(const std::vector<double> &w, double u01, std::vector<int> &src) {
  const int n = (int)w.size();
  std::vector<int> sampledIdx(n);
  std::vector<char> sampled(n, 0);
  const double interval = 1.0 / double(n);
  double samplePoint = interval * u01, cumulative = w[0];
  int idx = 0;
  for (int i = 0; i < n; i++) {
    while (samplePoint > cumulative && idx < n - 1) { idx++; cumulative += w[idx]; }
    sampledIdx[i] = idx;
    sampled[idx] = 1;
    samplePoint += interval;
  }
  src.resize(n);
  for (int i = 0; i < n; i++) src[i] = i;
  int next = 0, prev = -1;
  for (int i = 0; i < n; i++) {
    const int s = sampledIdx[i];
    const bool first = !(i > 0 && s == prev);
    prev = s;
    if (first) continue;                       // the particle keeps its own slot (:459-463)
    while (next < n && sampled[next]) next++;  // copies go into the un-sampled slots, in order (:464-477)
    src[next] = s;
    next++;
  }
}

extern "C" {

int rfsgpu_group_create(rfsgpu_group **out, int model, int n_particles, const int *device_ids, int n_dev, int gm_capacity) {
  if (!out || !device_ids || n_dev < 1 || n_particles < n_dev) return RFSGPU_ERR_INVALID;
  *out = nullptr;
  rfsgpu_group *g = new rfsgpu_group();
  g->N = n_particles;
  g->first.resize(n_dev + 1);
  for (int k = 0; k <= n_dev; k++) g->first[k] = (int)((long long)n_particles * k / n_dev);
  for (int k = 0; k < n_dev; k++) {
    rfsgpu_filter *f = nullptr;
    const int rc = rfsgpu_create(&f, model, g->first[k + 1] - g->first[k], device_ids[k], gm_capacity);
    if (rc != RFSGPU_OK) { rfsgpu_group_destroy(g); return rc; }
    if (f->inheritMode == RFSGPU_INHERIT_EAGER) g->inheritMode = RFSGPU_INHERIT_EAGER;   // (RFSGPU_BIRTH_INHERITANCE=eager in the environment)
    else rfsgpu_set_birth_inheritance(f, RFSGPU_INHERIT_EXTERNAL);
    g->shard.push_back(f);
  }
  g->pid.resize(n_particles); g->ppid.resize(n_particles);
  for (int p = 0; p < n_particles; p++) g->pid[p] = g->ppid[p] = p;
  g->sendBuf.assign(n_dev, nullptr); g->recvBuf.assign(n_dev, nullptr);
  g->sendCap.assign(n_dev, 0); g->recvCap.assign(n_dev, 0);
  g->evExport.assign(n_dev, nullptr);
  for (int k = 0; k < n_dev; k++) {
    hipSetDevice(device_ids[k]);
    if (hipEventCreateWithFlags(&g->evExport[k], hipEventDisableTiming) != hipSuccess) { rfsgpu_group_destroy(g); return RFSGPU_ERR_HIP; }
    for (int j = 0; j < n_dev; j++)   // xGMI peer access for the row transport (already-enabled / same-device answers are fine)
      if (device_ids[j] != device_ids[k]) { int can = 0; if (hipDeviceCanAccessPeer(&can, device_ids[k], device_ids[j]) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(device_ids[j], 0); }
    (void)hipGetLastError();
  }
  // RCCL communicators over the group's devices -- only when they are pairwise distinct (ncclCommInitAll refuses duplicates) and
  // RFSGPU_GROUP_RCCL is not 0.  Any failure leaves the host path in use and says why (rfsgpu_group_collective).
  {
    bool distinct = true;
    for (int k = 0; k < n_dev; k++) for (int j = 0; j < k; j++) distinct &= device_ids[j] != device_ids[k];
    const char *e = getenv("RFSGPU_GROUP_RCCL");
    if (e && atoi(e) == 0) g->rcclNote = "RFSGPU_GROUP_RCCL=0";
    else if (!distinct) g->rcclNote = "repeated device ids (several shards on one GPU)";
    else if ((g->rcclLib = rccl_load(g->rcclNote)) != nullptr) {
      g->comm.assign(n_dev, nullptr);
      const int rc = g_rccl.CommInitAll(g->comm.data(), n_dev, device_ids);
      if (rc != 0) {
        g->rcclNote = std::string("ncclCommInitAll: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error");
        g->comm.clear();
      } else {
        g->dTot.assign(n_dev, nullptr);
        for (int k = 0; k < n_dev && !g->comm.empty(); k++) {
          hipSetDevice(device_ids[k]);
          if (hipMalloc(&g->dTot[k], 2 * sizeof(double)) != hipSuccess) {
            g->rcclNote = "hipMalloc of the all-reduce buffer failed";
            for (void *c : g->comm) if (c) g_rccl.CommDestroy(c);
            g->comm.clear();
          }
        }
      }
      if (!g->comm.empty()) {           // side streams + events of the trailing normalisation
        g->side.assign(n_dev, nullptr); g->evPost.assign(n_dev, nullptr); g->evTot.assign(n_dev, nullptr);
        for (int k = 0; k < n_dev; k++) {
          hipSetDevice(device_ids[k]);
          if (hipStreamCreateWithFlags(&g->side[k], hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&g->evPost[k], hipEventDisableTiming) != hipSuccess ||
              hipEventCreateWithFlags(&g->evTot[k], hipEventDisableTiming) != hipSuccess) { g->side.clear(); break; }
        }
      }
      (void)hipGetLastError();
    }
  }
  *out = g;
  return RFSGPU_OK;
}
void rfsgpu_group_destroy(rfsgpu_group *g) {
  if (!g) return;
  for (size_t k = 0; k < g->shard.size(); k++) {
    if (g->shard[k]) hipSetDevice(g->shard[k]->device);
    if (k < g->sendBuf.size() && g->sendBuf[k]) hipFree(g->sendBuf[k]);
    if (k < g->recvBuf.size() && g->recvBuf[k]) hipFree(g->recvBuf[k]);
    if (k < g->evExport.size() && g->evExport[k]) hipEventDestroy(g->evExport[k]);
    if (k < g->dTot.size() && g->dTot[k]) hipFree(g->dTot[k]);
    if (k < g->side.size() && g->side[k]) hipStreamDestroy(g->side[k]);
    if (k < g->evPost.size() && g->evPost[k]) hipEventDestroy(g->evPost[k]);
    if (k < g->evTot.size() && g->evTot[k]) hipEventDestroy(g->evTot[k]);
    rfsgpu_destroy(g->shard[k]);
  }
  for (void *c : g->comm) if (c) g_rccl.CommDestroy(c);
  delete g;
}
// "rccl" when the weight sums are all-reduced over RCCL, else "host: <reason>"
const char *rfsgpu_group_collective(const rfsgpu_group *g) {
  static thread_local std::string s;
  if (!g) return "null group";
  s = g->comm.empty() ? ("host: " + g->rcclNote) : std::string("rccl");
  // (once rfsgpu_group_update_deferred has probed the shards' stream pairs: which hand-over the trailing normalisation uses)
  if (!g->comm.empty() && g->handover) s += g->handover == 1 ? "; hand-over: sequence numbers" : "; hand-over: stream events";
  return s.c_str();
}
const char *rfsgpu_group_last_error(const rfsgpu_group *g) { return g ? g->err.c_str() : "null group"; }
int rfsgpu_group_n_shards(const rfsgpu_group *g) { return g ? (int)g->shard.size() : -1; }
int rfsgpu_group_n_particles(const rfsgpu_group *g) { return g ? g->N : -1; }
rfsgpu_filter *rfsgpu_group_shard(rfsgpu_group *g, int k) { return (g && k >= 0 && k < (int)g->shard.size()) ? g->shard[k] : nullptr; }
int rfsgpu_group_locate(const rfsgpu_group *g, int particle, int *shard, int *slot) {
  if (!g || particle < 0 || particle >= g->N || !shard || !slot) return RFSGPU_ERR_INVALID;
  int k = 0;
  while (particle >= g->first[k + 1]) k++;
  *shard = k;
  *slot = particle - g->first[k];
  return RFSGPU_OK;
}

// configuration: the same structs on every shard (RBPHDFilter keeps one copy per OpenMP thread and broadcasts them at update, :460-464)
int rfsgpu_group_set_filter_config(rfsgpu_group *g, const rfsgpu_filter_config *c) {
  if (!g) return RFSGPU_ERR_INVALID;
  for (size_t k = 0; k < g->shard.size(); k++) GFWD(k, rfsgpu_set_filter_config(g->shard[k], c));
  return RFSGPU_OK;
}
int rfsgpu_group_set_model_rngbrg(rfsgpu_group *g, const rfsgpu_rngbrg_config *c) {
  if (!g) return RFSGPU_ERR_INVALID;
  for (size_t k = 0; k < g->shard.size(); k++) GFWD(k, rfsgpu_set_model_rngbrg(g->shard[k], c));
  return RFSGPU_OK;
}
int rfsgpu_group_set_kf_config(rfsgpu_group *g, const rfsgpu_kf_config *c) {
  if (!g) return RFSGPU_ERR_INVALID;
  for (size_t k = 0; k < g->shard.size(); k++) GFWD(k, rfsgpu_set_kf_config(g->shard[k], c));
  return RFSGPU_OK;
}
int rfsgpu_group_set_lmk_process_noise(rfsgpu_group *g, const double *Q) {
  if (!g) return RFSGPU_ERR_INVALID;
  for (size_t k = 0; k < g->shard.size(); k++) GFWD(k, rfsgpu_set_lmk_process_noise(g->shard[k], Q));
  return RFSGPU_OK;
}
// Victoria Park model on every shard (VERDICT r3 missing 5): configs[3] can be sharded like the 2-D configurations
int rfsgpu_group_set_model_victoriapark(rfsgpu_group *g, const rfsgpu_vp_config *c) {
  if (!g) return RFSGPU_ERR_INVALID;
  for (size_t k = 0; k < g->shard.size(); k++) GFWD(k, rfsgpu_set_model_victoriapark(g->shard[k], c));
  return RFSGPU_OK;
}
int rfsgpu_group_set_laser_scan(rfsgpu_group *g, const double *scan, int n) {
  if (!g) return RFSGPU_ERR_INVALID;
  for (size_t k = 0; k < g->shard.size(); k++) GFWD(k, rfsgpu_set_laser_scan(g->shard[k], scan, n));
  return RFSGPU_OK;
}
int rfsgpu_group_set_phase_timing(rfsgpu_group *g, int on) {
  if (!g) return RFSGPU_ERR_INVALID;
  for (size_t k = 0; k < g->shard.size(); k++) GFWD(k, rfsgpu_set_phase_timing(g->shard[k], on));
  return RFSGPU_OK;
}
// RBPHDFilter::TimingInfo of the group: the shards run side by side, so a *_wall bucket is the LARGEST of the shards' (what the
// step waited for), a *_cpu bucket (host time inside the calls, made one after the other by this thread) their SUM.
int rfsgpu_group_get_timing(rfsgpu_group *g, rfsgpu_timing *t) {
  if (!g || !t) return RFSGPU_ERR_INVALID;
  memset(t, 0, sizeof(*t));
  long long *acc = reinterpret_cast<long long *>(t);
  for (size_t k = 0; k < g->shard.size(); k++) {
    rfsgpu_timing s1;
    GFWD(k, rfsgpu_get_timing(g->shard[k], &s1));
    const long long *v = reinterpret_cast<const long long *>(&s1);
    for (int q = 0; q < 14; q++) acc[q] = (q & 1) ? acc[q] + v[q] : std::max(acc[q], v[q]);   // even: *_wall, odd: *_cpu
  }
  return RFSGPU_OK;
}
int rfsgpu_group_set_poses(rfsgpu_group *g, const double *x, const double *cov, int cov_stride) {
  if (!g || !x) return RFSGPU_ERR_INVALID;
  for (size_t k = 0; k < g->shard.size(); k++)
    GFWD(k, rfsgpu_set_poses(g->shard[k], x + 3 * (size_t)g->first[k], cov ? cov + (size_t)cov_stride * g->first[k] : nullptr, cov_stride));
  return RFSGPU_OK;
}
int rfsgpu_group_get_poses(rfsgpu_group *g, double *x) {
  if (!g || !x) return RFSGPU_ERR_INVALID;
  for (size_t k = 0; k < g->shard.size(); k++) GFWD(k, rfsgpu_get_poses(g->shard[k], x + 3 * (size_t)g->first[k]));
  return RFSGPU_OK;
}
// A total left pending by rfsgpu_group_update_deferred: every shard's stream waits for its collective, then divides (device-side
// divisor).  Called by whatever reads or replaces the weights.
static int group_flush(rfsgpu_group *g) {
  if (!g->pendingTotal) return RFSGPU_OK;
  g->pendingTotal = false;
  for (size_t k = 0; k < g->shard.size(); k++) {
    rfsgpu_filter *f = g->shard[k];
    hipSetDevice(f->device);
    GCHK(hipStreamWaitEvent(f->stream, g->evTot[k], 0));
    GFWD(k, rfsgpu_normalize_weights(f, 0.0, g->dTot[k]));
  }
  return RFSGPU_OK;
}
#define GFLUSH(g) do { const int rcf_ = group_flush(g); if (rcf_ != RFSGPU_OK) return rcf_; } while (0)
int rfsgpu_group_set_weights(rfsgpu_group *g, const double *w) {
  if (!g || !w) return RFSGPU_ERR_INVALID;
  GFLUSH(g);
  for (size_t k = 0; k < g->shard.size(); k++) GFWD(k, rfsgpu_set_weights(g->shard[k], w + g->first[k]));
  return RFSGPU_OK;
}
int rfsgpu_group_get_weights(rfsgpu_group *g, double *w) {
  if (!g || !w) return RFSGPU_ERR_INVALID;
  GFLUSH(g);
  for (size_t k = 0; k < g->shard.size(); k++) GFWD(k, rfsgpu_get_weights(g->shard[k], w + g->first[k]));
  return RFSGPU_OK;
}
// RFSGPU_INHERIT_REFERENCE (default) or RFSGPU_INHERIT_EAGER for the whole group (see rfsgpu_set_birth_inheritance).
int rfsgpu_group_set_birth_inheritance(rfsgpu_group *g, int mode) {
  if (!g || (mode != RFSGPU_INHERIT_REFERENCE && mode != RFSGPU_INHERIT_EAGER)) return RFSGPU_ERR_INVALID;
  g->inheritMode = mode;
  for (size_t k = 0; k < g->shard.size(); k++)
    GFWD(k, rfsgpu_set_birth_inheritance(g->shard[k], mode == RFSGPU_INHERIT_EAGER ? RFSGPU_INHERIT_EAGER : RFSGPU_INHERIT_EXTERNAL));
  return RFSGPU_OK;
}
int rfsgpu_group_get_particle_ids(rfsgpu_group *g, int *id, int *parent_id) {
  if (!g) return RFSGPU_ERR_INVALID;
  for (int p = 0; p < g->N; p++) { if (id) id[p] = g->pid[p]; if (parent_id) parent_id[p] = g->ppid[p]; }
  return RFSGPU_OK;
}
// RBPHDFilter::predict, map part (:415-442).  In the predicts that follow a resampling the reference first copies, slot by slot in
// ascending order, unused_measurements_ / birthGaussians_ from SLOT idParent_ in its current state (:1005-1011).  With immediate
// births (birthGaussianMeasurementCountThreshold == 1, the 2-D simulator: no candidate list ever exists) that walk has a closed
// form over the lists as they are before the predict -- own list if idParent_ == slot, the parent slot's list if it is a HIGHER
// slot (not yet visited), nothing if it is a LOWER one (already consumed) -- which needs only the 8-byte masks of all shards.
// Configurations that keep candidate lists take the walk in full, level by level over the global slots (group_predict_levels).
struct GroupBirthList {
  unsigned long long mask = 0;
  int n = 0;
  std::vector<double> mean, cov;
  std::vector<int> sup, chk;
};
// Level 0 = slots that keep their lists or copy from a HIGHER slot (its lists as they are before this predict); level L = slots
// whose parent id names a LOWER slot of level L - 1 (its lists after its own birth step) -- csrc/birth.h has the single-GPU form.
// Per level: read the needed source lists (host-staged: this happens in the predicts that follow a resampling only), install
// them in the destination slots, run the birth step of that level on every shard.
static int group_predict_levels(rfsgpu_group *g, int add_birth) {
  const int N = g->N, S = (int)g->shard.size();
  const int D = g->shard[0]->D;
  auto shard_of = [&](int p) { int k = 0; while (p >= g->first[k + 1]) k++; return k; };
  std::vector<int> level((size_t)N, 0);
  int maxL = 0;
  for (int i = 0; i < N; i++) {
    int p = g->ppid[i];
    if (p < 0 || p >= N) p = i;
    level[i] = (p >= i) ? 0 : level[p] + 1;
    if (level[i] > maxL) maxL = level[i];
  }
  std::vector<unsigned long long> masks((size_t)N);
  for (int L = 0; L <= maxL; L++) {
    for (int k = 0; k < S; k++) GFWD(k, rfsgpu_get_unused_masks(g->shard[k], masks.data() + g->first[k]));
    std::vector<int> dst;
    for (int i = 0; i < N; i++) if (level[i] == L && g->ppid[i] != i && g->ppid[i] >= 0 && g->ppid[i] < N) dst.push_back(i);
    std::vector<GroupBirthList> src(dst.size());
    for (size_t t = 0; t < dst.size(); t++) {          // every source is read before any destination is written
      const int q = g->ppid[dst[t]], kq = shard_of(q);
      GroupBirthList &b = src[t];
      b.mask = masks[q];
      b.mean.resize((size_t)RFSGPU_MAX_CANDIDATES * D); b.cov.resize((size_t)RFSGPU_MAX_CANDIDATES * D * D);
      b.sup.resize(RFSGPU_MAX_CANDIDATES); b.chk.resize(RFSGPU_MAX_CANDIDATES);
      GFWD(kq, rfsgpu_export_birth_candidates(g->shard[kq], q - g->first[kq], RFSGPU_MAX_CANDIDATES, &b.n, b.mean.data(), b.cov.data(), b.sup.data(), b.chk.data()));
      if (b.n > RFSGPU_MAX_CANDIDATES) b.n = RFSGPU_MAX_CANDIDATES;
    }
    for (size_t t = 0; t < dst.size(); t++) {
      const int i = dst[t], ki = shard_of(i);
      masks[i] = src[t].mask;
      GFWD(ki, rfsgpu_import_birth_candidates(g->shard[ki], i - g->first[ki], src[t].n, src[t].mean.data(), src[t].cov.data(), src[t].sup.data(), src[t].chk.data()));
    }
    if (!dst.empty()) for (int k = 0; k < S; k++) GFWD(k, rfsgpu_set_unused_masks(g->shard[k], masks.data() + g->first[k]));
    for (int k = 0; k < S; k++) GFWD(k, rfsgpu_predict_map_level(g->shard[k], add_birth, level.data() + g->first[k], L, L == 0 ? 1 : 0));
  }
  return RFSGPU_OK;
}
int rfsgpu_group_predict_map(rfsgpu_group *g, int add_birth) {
  if (!g) return RFSGPU_ERR_INVALID;
  if (add_birth && g->resampleOccured && g->inheritMode == RFSGPU_INHERIT_REFERENCE) {
    const rfsgpu_filter *f0 = g->shard[0];
    bool anyCopy = false;
    for (int p = 0; p < g->N; p++) anyCopy |= g->ppid[p] != p;
    bool candUsed = false;            // any shard that has held candidate lists (imported ones included) sends the whole group through the walk
    for (size_t k = 0; k < g->shard.size(); k++) candUsed |= rfsgpu_has_birth_candidates(g->shard[k]) == 1;
    if (anyCopy && (f0->D != 2 || f0->cfg.birthGaussianMeasurementCountThreshold != 1u || candUsed)) return group_predict_levels(g, add_birth);
    if (anyCopy) {
      std::vector<unsigned long long> m((size_t)g->N), mn((size_t)g->N);
      for (size_t k = 0; k < g->shard.size(); k++) GFWD(k, rfsgpu_get_unused_masks(g->shard[k], m.data() + g->first[k]));
      bool any = false;
      for (int p = 0; p < g->N; p++) {
        const int q = g->ppid[p];
        mn[p] = (q == p || q < 0 || q >= g->N) ? m[p] : (q > p ? m[q] : 0ull);
        any |= mn[p] != m[p];
      }
      if (any) for (size_t k = 0; k < g->shard.size(); k++) GFWD(k, rfsgpu_set_unused_masks(g->shard[k], mn.data() + g->first[k]));
    }
  }
  for (size_t k = 0; k < g->shard.size(); k++) {
    g->shard[k]->externalAck = true;   // the group has applied the rule above (or no slot has a foreign parent)
    GFWD(k, rfsgpu_predict_map(g->shard[k], add_birth));
  }
  return RFSGPU_OK;
}
// (every shard is synchronised even when one reports an error: the others' streams and error words must not stay unharvested)
int rfsgpu_group_synchronize(rfsgpu_group *g) {
  if (!g) return RFSGPU_ERR_INVALID;
  GFLUSH(g);
  int first = RFSGPU_OK;
  for (size_t k = 0; k < g->shard.size(); k++) {
    const int rc = rfsgpu_synchronize(g->shard[k]);
    if (rc != RFSGPU_OK && first == RFSGPU_OK) first = gfail(g, rc, std::string("shard ") + std::to_string(k) + ": " + rfsgpu_last_error(g->shard[k]));
  }
  return first;
}

// {sum w, sum w^2} over all shards (each shard's pair was left by its post kernel / weight_sums kernel), added in shard order.
// HOST path: one 16-byte copy per shard and a synchronisation of every shard.
static int group_totals(rfsgpu_group *g, bool launch_sums, double tot[2]) {
  const int S = (int)g->shard.size();
  for (int k = 0; k < S; k++) {
    rfsgpu_filter *f = g->shard[k];
    hipSetDevice(f->device);
    if (launch_sums) GFWD(k, rfsgpu_weight_sums_async(f));
    GCHK(hipMemcpyAsync(f->hSums, f->dSums, 2 * sizeof(double), hipMemcpyDeviceToHost, f->stream));
  }
  tot[0] = tot[1] = 0.0;
  const int rcs = rfsgpu_group_synchronize(g);    // all shards; also surfaces device-side errors of the async steps
  if (rcs != RFSGPU_OK) return rcs;
  for (int k = 0; k < S; k++) {
    tot[0] += g->shard[k]->hSums[0];
    tot[1] += g->shard[k]->hSums[1];
  }
  return RFSGPU_OK;
}
// RCCL path: all-reduce (sum) of the shards' pairs, each shard's collective ordered on its own stream after the kernel that wrote
// its pair; the totals stay on the devices (dTot[k]).  tot != nullptr: they are also brought to the host (from shard 0; that
// stream is synchronised, the others are not).
static int group_allreduce(rfsgpu_group *g, bool launch_sums, double *tot) {
  const int S = (int)g->shard.size();
  if (launch_sums) for (int k = 0; k < S; k++) GFWD(k, rfsgpu_weight_sums_async(g->shard[k]));
  int rc = g_rccl.GroupStart();
  for (int k = 0; k < S && rc == 0; k++) {
    rfsgpu_filter *f = g->shard[k];
    hipSetDevice(f->device);
    rc = g_rccl.AllReduce(f->dSums, g->dTot[k], 2, /*ncclDouble*/ 8, /*ncclSum*/ 0, g->comm[k], f->stream);
  }
  const int rce = g_rccl.GroupEnd();
  if (rc == 0) rc = rce;
  if (rc != 0) return gfail(g, RFSGPU_ERR_HIP, std::string("ncclAllReduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error"));
  if (tot) {
    rfsgpu_filter *f0 = g->shard[0];
    hipSetDevice(f0->device);
    GCHK(hipMemcpyAsync(f0->hSums, g->dTot[0], 2 * sizeof(double), hipMemcpyDeviceToHost, f0->stream));
    GCHK(hipStreamSynchronize(f0->stream));
    tot[0] = f0->hSums[0];
    tot[1] = f0->hSums[1];
  }
  return RFSGPU_OK;
}

// RBPHDFilter::update body on every shard (:444-523): one fused step + post kernel per shard, all shards in flight together.
// Leaves the weights un-normalised; sums_out (may be null) receives {sum w, sum w^2} over all shards.  RCCL path with sums_out ==
// nullptr: nothing waits for the GPUs (device-side errors surface at the next synchronising call).
int rfsgpu_group_update(rfsgpu_group *g, const double *z, int n_z, double *sums_out) {
  if (!g) return RFSGPU_ERR_INVALID;
  GFLUSH(g);
  for (size_t k = 0; k < g->shard.size(); k++) GFWD(k, rfsgpu_step_async(g->shard[k], z, n_z, 0));
  if (n_z > 0) g->resampleOccured = false;       // RBPHDFilter.hpp:526
  if (n_z == 0) { if (sums_out) { sums_out[0] = sums_out[1] = 0.0; } return RFSGPU_OK; }   // (:451-452: nothing ran, no sums were written)
  if (!g->comm.empty()) {
    if (!sums_out) return RFSGPU_OK;             // (the pairs are reduced by the normalisation that follows)
    return group_allreduce(g, false, sums_out);
  }
  double tot[2];
  const int rc = group_totals(g, false, tot);
  if (rc != RFSGPU_OK) return rc;
  if (sums_out) { sums_out[0] = tot[0]; sums_out[1] = tot[1]; }
  return RFSGPU_OK;
}
// rfsgpu_group_update with the weight normalisation trailing by one step (round 5): every shard's post kernel divides
// its weights by the all-reduced total of the PREVIOUS call (rfsgpu_step_async_deferred) and leaves this step's pair; the pairs
// are all-reduced on a side stream per shard, beside the next step's kernel.  Nothing waits for the GPUs.  The next call that
// reads or replaces the weights (normalize, get / set_weights, resample, apply_plan, update, update_io, synchronize) applies the
// pending total first.  Host path (repeated device ids, no librccl): rfsgpu_group_update + rfsgpu_group_normalize, as before.
int rfsgpu_group_update_deferred(rfsgpu_group *g, const double *z, int n_z) {
  if (!g) return RFSGPU_ERR_INVALID;
  if (g->comm.empty() || g->side.empty()) {
    const int rc = rfsgpu_group_update(g, z, n_z, nullptr);
    return rc != RFSGPU_OK ? rc : rfsgpu_group_normalize(g, nullptr);
  }
  const int S = (int)g->shard.size();
  // hand-over between a shard's stream and its side stream: device sequence numbers (rfsgpu_step_async_trailing: nothing but the two
  // kernels on the step's stream) where the two streams of EVERY shard make progress side by side -- probed once, at the first
  // deferred step (rfsgpu_collective_probe) --, otherwise an event record and an event wait per step (rfsgpu_step_async_deferred, the
  // fall-back: +9 us per step with one rank).  RFSGPU_GROUP_EVENTS=1 forces the event form (A/B).
  if (g->handover == 0) {
    const char *e = getenv("RFSGPU_GROUP_EVENTS");
    bool ev = e && e[0] == '1';
    for (int k = 0; k < S && !ev; k++) {
      int ok = 0;
      GFWD(k, rfsgpu_collective_probe(g->shard[k], g->side[k], &ok));
      if (!ok) ev = true;
    }
    g->handover = ev ? 2 : 1;
  }
  const bool useEvents = g->handover == 2;
  for (int k = 0; k < S; k++) {
    rfsgpu_filter *f = g->shard[k];
    if (useEvents) {
      GFWD(k, rfsgpu_step_async_deferred(f, z, n_z, g->pendingTotal ? g->dTot[k] : nullptr, g->pendingTotal ? (void *)g->evTot[k] : nullptr));
      hipSetDevice(f->device);
      GCHK(hipEventRecord(g->evPost[k], f->stream));
      GCHK(hipStreamWaitEvent(g->side[k], g->evPost[k], 0));
    } else {
      GFWD(k, rfsgpu_step_async_trailing(f, z, n_z, g->dTot[k], g->pendingTotal ? 1 : 0));
      GFWD(k, rfsgpu_collective_gate(f, g->side[k]));
    }
  }
  if (n_z > 0) g->resampleOccured = false;       // RBPHDFilter.hpp:526
  int rc = g_rccl.GroupStart();
  for (int k = 0; k < S && rc == 0; k++) {
    rfsgpu_filter *f = g->shard[k];
    hipSetDevice(f->device);
    rc = g_rccl.AllReduce(f->dSums, g->dTot[k], 2, /*ncclDouble*/ 8, /*ncclSum*/ 0, g->comm[k], g->side[k]);
  }
  const int rce = g_rccl.GroupEnd();
  if (rc == 0) rc = rce;
  if (rc != 0) return gfail(g, RFSGPU_ERR_HIP, std::string("ncclAllReduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error"));
  for (int k = 0; k < S; k++) {
    hipSetDevice(g->shard[k]->device);
    if (!useEvents) GFWD(k, rfsgpu_collective_publish(g->shard[k], g->side[k]));
    GCHK(hipEventRecord(g->evTot[k], g->side[k]));      // (waited for only when the pending total is applied: group_flush)
  }
  g->pendingTotal = true;
  return RFSGPU_OK;
}
// RBPHDFilter::update (:444-541) over the group with its inputs and outputs in one call (the group form of rfsgpu_update_io; what
// integration/RBPHDFilter_rfsgpu.hpp's update() calls when RFSGPU_DEVICES lists several GPUs): global poses (+ covariances) and
// weights in, every shard's chain enqueued before the first wait, the updated weights of all particles out.  The shards' device
// error words are read by THIS call (rfsgpu_group_update(..., NULL) on the RCCL path left them for a later call).
int rfsgpu_group_update_io(rfsgpu_group *g, const double *x, const double *x_cov, int cov_stride, const double *w_in, const double *z, int n_z, double *w_out) {
  if (!g) return RFSGPU_ERR_INVALID;
  GFLUSH(g);
  if (x_cov && cov_stride != 0 && cov_stride != 9) return gfail(g, RFSGPU_ERR_INVALID, "group_update_io: cov_stride must be 0 or 9");
  const int S = (int)g->shard.size();
  for (int k = 0; k < S; k++) {
    const size_t o = (size_t)g->first[k];
    GFWD(k, update_io_begin(g->shard[k], RFSGPU_CYCLE_NO_PREDICT, x ? x + 3 * o : nullptr, x_cov ? x_cov + (cov_stride == 9 ? 9 * o : 0) : nullptr, cov_stride,
                            w_in ? w_in + o : nullptr, z, n_z, w_out != nullptr));
  }
  if (n_z > 0) g->resampleOccured = false;       // RBPHDFilter.hpp:526
  int first = RFSGPU_OK;
  for (int k = 0; k < S; k++) {
    const int rc = update_io_end(g->shard[k], w_out ? w_out + g->first[k] : nullptr);
    if (rc != RFSGPU_OK && first == RFSGPU_OK) first = gfail(g, rc, std::string("shard ") + std::to_string(k) + ": " + rfsgpu_last_error(g->shard[k]));
  }
  return first;
}
// ParticleFilter::normalizeWeights over the whole particle set (:352-363).
int rfsgpu_group_normalize(rfsgpu_group *g, double *sums_out) {
  if (!g) return RFSGPU_ERR_INVALID;
  GFLUSH(g);
  if (!g->comm.empty()) {
    const int rc = group_allreduce(g, true, sums_out);
    if (rc != RFSGPU_OK) return rc;
    for (size_t k = 0; k < g->shard.size(); k++) GFWD(k, rfsgpu_normalize_weights(g->shard[k], 0.0, g->dTot[k]));   // divisor read on the device
    return RFSGPU_OK;
  }
  double tot[2];
  int rc = group_totals(g, true, tot);
  if (rc != RFSGPU_OK) return rc;
  for (size_t k = 0; k < g->shard.size(); k++) GFWD(k, rfsgpu_normalize_weights(g->shard[k], tot[0], nullptr));
  if (sums_out) { sums_out[0] = tot[0]; sums_out[1] = tot[1]; }
  return RFSGPU_OK;
}

static int grow(rfsgpu_group *g, unsigned char *&buf, size_t &cap, size_t need) {
  if (need <= cap) return RFSGPU_OK;
  if (buf) GCHK(hipFree(buf));
  buf = nullptr; cap = 0;
  GCHK(hipMalloc(&buf, need));
  cap = need;
  return RFSGPU_OK;
}

// Carry out a global plan (src[global slot] = global source slot; a source keeps itself).
int rfsgpu_group_apply_plan(rfsgpu_group *g, const int *src) {
  if (!g || !src) return RFSGPU_ERR_INVALID;
  GFLUSH(g);
  const int S = (int)g->shard.size();
  auto shard_of = [&](int p) { int k = 0; while (p >= g->first[k + 1]) k++; return k; };
  for (int p = 0; p < g->N; p++)
    if (src[p] < 0 || src[p] >= g->N || src[src[p]] != src[p]) return gfail(g, RFSGPU_ERR_INVALID, "apply_plan: a source must be a slot that keeps itself");
  // children that cross a shard boundary, grouped (source shard -> destination shard), ascending destination slot
  std::vector<std::vector<std::vector<int>>> cross(S, std::vector<std::vector<int>>(S));
  std::vector<std::vector<int>> localSrc(S);
  for (int k = 0; k < S; k++) { localSrc[k].resize(g->first[k + 1] - g->first[k]); for (size_t q = 0; q < localSrc[k].size(); q++) localSrc[k][q] = (int)q; }
  for (int p = 0; p < g->N; p++) {
    if (src[p] == p) continue;
    const int kd = shard_of(p), ks = shard_of(src[p]);
    if (kd == ks) localSrc[kd][p - g->first[kd]] = src[p] - g->first[kd];
    else cross[ks][kd].push_back(p);
  }
  const size_t R = rfsgpu_slab_row_bytes(g->shard[0]);
  for (int k = 1; k < S; k++)   // source and destination must agree on the row layout (it depends on per-shard state in eager mode)
    if (rfsgpu_slab_row_bytes(g->shard[k]) != R) return gfail(g, RFSGPU_ERR_INVALID, "apply_plan: shards disagree on the migration row size (candidate lists imported on some shards only?)");
  // 1. export on every source shard (its stream), all destinations back to back in one buffer
  std::vector<std::vector<size_t>> sendOff(S, std::vector<size_t>(S, 0)), recvOff(S, std::vector<size_t>(S, 0));
  for (int ks = 0; ks < S; ks++) {
    std::vector<int> slots;
    for (int kd = 0; kd < S; kd++) { sendOff[ks][kd] = slots.size(); for (int p : cross[ks][kd]) slots.push_back(src[p] - g->first[ks]); }
    if (slots.empty()) continue;
    rfsgpu_filter *f = g->shard[ks];
    hipSetDevice(f->device);
    int rc = grow(g, g->sendBuf[ks], g->sendCap[ks], slots.size() * R);
    if (rc != RFSGPU_OK) return rc;
    GFWD(ks, rfsgpu_export_slab_rows(f, slots.data(), (int)slots.size(), g->sendBuf[ks]));
    GCHK(hipEventRecord(g->evExport[ks], f->stream));
    g->rowsMigrated += (long long)slots.size();
    g->bytesMigrated += (long long)(slots.size() * R);
  }
  // 2. per destination shard: wait for the exporters, pull the rows device-to-device, gather the local children, unpack
  for (int kd = 0; kd < S; kd++) {
    rfsgpu_filter *f = g->shard[kd];
    hipSetDevice(f->device);
    std::vector<int> slots;
    for (int ks = 0; ks < S; ks++) { recvOff[kd][ks] = slots.size(); for (int p : cross[ks][kd]) slots.push_back(p - g->first[kd]); }
    if (!slots.empty()) {
      int rc = grow(g, g->recvBuf[kd], g->recvCap[kd], slots.size() * R);
      if (rc != RFSGPU_OK) return rc;
      for (int ks = 0; ks < S; ks++) {
        const size_t n = cross[ks][kd].size();
        if (!n) continue;
        GCHK(hipStreamWaitEvent(f->stream, g->evExport[ks], 0));
        const rfsgpu_filter *fs = g->shard[ks];
        if (fs->device == f->device)
          GCHK(hipMemcpyAsync(g->recvBuf[kd] + recvOff[kd][ks] * R, g->sendBuf[ks] + sendOff[ks][kd] * R, n * R, hipMemcpyDeviceToDevice, f->stream));
        else
          GCHK(hipMemcpyPeerAsync(g->recvBuf[kd] + recvOff[kd][ks] * R, f->device, g->sendBuf[ks] + sendOff[ks][kd] * R, fs->device, n * R, f->stream));
      }
    }
    GFWD(kd, rfsgpu_resample_apply(f, localSrc[kd].data()));   // also resets every weight to 1 (:486-489); stream-ordered (round 6)
    if (!slots.empty()) GFWD(kd, rfsgpu_import_slab_rows(f, slots.data(), (int)slots.size(), g->recvBuf[kd]));
  }
  for (int k = 0; k < S; k++) GFWD(k, rfsgpu_synchronize(g->shard[k]));   // (send buffers are reused by the next resampling)
  g->lastPlan.assign(src, src + g->N);
  // ids as ParticleFilter::resample leaves them (:446-479; a copy keeps its source's id, Particle::copy), over GLOBAL slots
  for (int p = 0; p < g->N; p++) {
    if (src[p] != p) { g->pid[p] = g->pid[src[p]]; g->ppid[p] = g->pid[src[p]]; }
    else g->ppid[p] = g->pid[p];
  }
  g->resampleOccured = true;
  return RFSGPU_OK;
}

// ParticleFilter::resample (:399-492) over the whole particle set: normalise; N_eff = 1 / sum w^2; resample only if
// N_eff <= eff_n_threshold or N_eff / N <= eff_n_threshold / N (:412); systematic sampling with the caller's uniform draw.
// plan_out (may be null, N ints) receives the global plan when *fired.
int rfsgpu_group_resample(rfsgpu_group *g, double eff_n_threshold, double u01, int *fired, int *plan_out) {
  if (!g || !fired) return RFSGPU_ERR_INVALID;
  GFLUSH(g);
  *fired = 0;
  double tot[2];
  int rc = rfsgpu_group_normalize(g, tot);
  if (rc != RFSGPU_OK) return rc;
  const double nEff = tot[0] * tot[0] / tot[1];   // = 1 / sum (w / S)^2
  if (nEff > eff_n_threshold && nEff / g->N > eff_n_threshold / g->N) return RFSGPU_OK;
  std::vector<double> w(g->N);
  rc = rfsgpu_group_get_weights(g, w.data());
  if (rc != RFSGPU_OK) return rc;
  std::vector<int> plan;
  systematic_plan(w, u01, plan);
  rc = rfsgpu_group_apply_plan(g, plan.data());
  if (rc != RFSGPU_OK) return rc;
  if (plan_out) memcpy(plan_out, plan.data(), (size_t)g->N * sizeof(int));
  *fired = 1;
  return RFSGPU_OK;
}
int rfsgpu_group_migration_stats(const rfsgpu_group *g, long long *rows, long long *bytes) {
  if (!g) return RFSGPU_ERR_INVALID;
  if (rows) *rows = g->rowsMigrated;
  if (bytes) *bytes = g->bytesMigrated;
  return RFSGPU_OK;
}
// map access by GLOBAL particle index (getGMSize / getLandmark, include/RBPHDFilter.hpp:1152-1178)
int rfsgpu_group_gm_size(rfsgpu_group *g, int particle) {
  int k, s;
  if (rfsgpu_group_locate(g, particle, &k, &s) != RFSGPU_OK) return -1;
  return rfsgpu_gm_size(g->shard[k], s);
}
int rfsgpu_group_get_landmark(rfsgpu_group *g, int particle, int m, double *mean, double *cov, double *w) {
  int k, s;
  if (rfsgpu_group_locate(g, particle, &k, &s) != RFSGPU_OK) return RFSGPU_ERR_INVALID;
  return rfsgpu_get_landmark(g->shard[k], s, m, mean, cov, w);
}

}  // extern "C"
