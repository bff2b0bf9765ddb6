// common.h -- device-side layout, parameter block and wave-level helpers (gfx950 / wave64 only).
//
// HBM layout of the per-particle Gaussian mixtures (the "map" of each particle):
//   slab[particle][plane][cap] doubles, planes = { W, WP, MX, MY, SXX, SXY, SYY }
// i.e. structure-of-arrays inside one contiguous slab per particle, so that lane l reading landmark
// (pass*64 + l) of a plane is one coalesced 512-byte wave access, and a resample copy of one particle is
// one contiguous block.  Sigma is stored packed-symmetric (the reference keeps full 2x2 matrices; its
// off-diagonals can differ by <= 1 ulp only for freshly born Gaussians -- see DESIGN.md).
// Two slabs (ping-pong): kernels that reorder (sort / compact) read one and write the other.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RFS_WAVE 64
// LDS head of every hot-path kernel: the measurement set as doubles [2*MAX_Z] | an fp32 copy for the innovation-gate
// prefilter, ranges [MAX_Z] then bearings [MAX_Z] | {max |range|, max |bearing|} as floats (+ pad)
#define RFS_Z_LDS_BYTES (2 * RFSGPU_MAX_Z * 8 + 2 * RFSGPU_MAX_Z * 4 + 16)
#define RFS_PI 3.14159265358979323846 /* == acos(-1) in fp64 (reference include/RandomVec.hpp:55) */
#define RFS_DENORM_MIN 4.9406564584124654e-324

enum Plane { PL_W = 0, PL_WP = 1, PL_MX = 2, PL_MY = 3, PL_SXX = 4, PL_SXY = 5, PL_SYY = 6, PL_COUNT = 7 };

enum ErrBits { ERRBIT_CAPACITY = 1, ERRBIT_MURTY = 2, ERRBIT_EVALPTS = 4, ERRBIT_BIRTHLIST = 8, ERRBIT_COLLECTIVE = 16 };

// Everything a kernel needs besides the buffers; passed by value (lives in SGPRs / kernarg segment).
struct Params {
  // MeasurementModel_RngBrg
  double R[4];
  double Pd, clutter, rmax, rmin, rbuf;
  double rmaxIn, rmaxOut, rminIn, rminOut;  // rmax - rbuf, rmax + rbuf, rmin + rbuf, rmin - rbuf (host-computed: kernel arguments stay in SGPRs)
  // KalmanFilter_RngBrg
  double kfRange, kfBearing;
  // RBPHDFilter::Config
  double birthW;
  double newGaussMd2;      // newGaussianCreateInnovMDThreshold^2
  double evalMinW;         // importanceWeightingEvalPointGuassianWeight
  double weightingMd2;     // importanceWeightingMeasurementLikelihoodMDThreshold^2
  double mergeT2;          // gaussianMergingThreshold^2
  double mergeInfl;
  double pruneT;
  double Qlm[3];           // packed xx, xy, yy
  int evalCount;           // importanceWeightingEvalPointCount
  int useCluster;
  unsigned birthCountThr, birthCurThr, birthCheckThr;
  double birthSupportD2;   // birthGaussianMeasurementSupportDist^2
  int poseCovStride;       // 0 shared, 9 per particle
  int exactPartitions;     // rfsgpu_set_partition_mode: partitions with nR + nC > 8 by the exact subset recurrence instead of Murty-200
  int denseIntensity;      // RFSGPU_DENSE_INTENSITY=1: the intensity sums over EVERY (evaluation point, Gaussian) pair, the reference's term list (deviation 9 off)
  // MeasurementModel_VictoriaPark (model 1); R[] above then holds its 2x2 range-bearing block
  double R9[9];
  double Slb;
  double PdTable[16];
  int nPd;
  double vpClutter;        // expectedClutterNumber / FoV area of the current scan (setLaserScan)
  double vpExpClutter;     // clutterIntensityIntegral
  double bmax, bmin, bufferPd;
  double Qlm6[6];          // packed xx, xy, xd, yy, yd, dd
  double twoPiPowD;        // pow(2*pi, d_z) as the host libm rounds it (RandomVec.hpp:419)
};

// The measurement set of a step, by value in the kernel-argument block (<= 1.5 KB): no staging buffer, no copy-engine hop.
struct ZArg {
  double v[RFSGPU_MAX_Z * 3];
};

struct Buffers {
  double *slab[2];
  int *count;
  double *pose;      // [N][3]
  double *poseCov;   // [9] or [N][9]
  double *weight;    // [N]
  unsigned long long *unusedMask;  // [N]
  int *nInFov;       // [N]
  int *err;          // [1] OR-ed ErrBits
  double *Z;         // [RFSGPU_MAX_Z][d_z]
  int N, cap;
  int npl;           // planes per particle: 7 (2-D) or 11 (3-D)
  double *scan;      // [RFSGPU_VP_MAX_SCAN] laser scan (Victoria Park Pd)
  int nScan;
  // birth-Gaussian candidates (RBPHDFilter::birthGaussians_), list order == array order
  double *candMean;  // [N][RFSGPU_MAX_CANDIDATES][3]
  double *candCov;   // [N][RFSGPU_MAX_CANDIDATES][6] packed symmetric
  int *candSup, *candChk;  // [N][RFSGPU_MAX_CANDIDATES]
  int *candCount;    // [N]
  long long *dbg;    // section timestamps (only written by -DRFS_PROFILE builds; NULL otherwise)
};

// Which particles a birth launch works on.  Normally all of them.  In the first predicts after a resampling the reference's
// lazy copy of the per-slot birth state (RBPHDFilter.hpp:1005-1011, below) orders the slots: `level` then holds, per slot, the
// length of its chain of LOWER-slot parents, and one launch takes one level; the static step is done by the first launch only.
struct BirthLevel {
  const int *level;   // nullptr: every particle
  int cur;
  int doStatic;
  __device__ bool mine(int i) const { return level == nullptr || level[i] == cur; }
};

// Section timing for kernel tuning (tools/kernel_sections.py builds a separate -DRFS_PROFILE library):
// particle `RFS_PROFILE_PARTICLE`'s lane 0 stamps s_memtime at section boundaries.
#ifdef RFS_PROFILE
#define DBG_T(base, k) do { if (B.dbg && i == 7 && lane == 0) B.dbg[(base) + (k)] = (long long)__builtin_readcyclecounter(); } while (0)
#define DBG_TB(base, k) do { if (B.dbg && i == 7 && threadIdx.x == 0) B.dbg[(base) + (k)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define DBG_T(base, k) do { } while (0)
#define DBG_TB(base, k) do { } while (0)
#endif
// Tuning aid (tools/cut_profile.py): a -DRFS_STOP_AT=k build ends every wave at cut point k, so the SQ counters of the
// truncated kernel give the cumulative instruction counts up to that point.  Absent from product builds.
#ifdef RFS_STOP_AT
#define RFS_CUT(k) do { if (RFS_STOP_AT == (k)) __builtin_amdgcn_endpgm(); } while (0)
#else
#define RFS_CUT(k) do { } while (0)
#endif

__device__ __forceinline__ double *plane(double *slab, int cap, int particle, int pl) {
  return slab + ((size_t)particle * PL_COUNT + pl) * (size_t)cap;
}

// ---- wave64 helpers -------------------------------------------------------------------------------
// Ordering point for LDS traffic between lanes of ONE wave (wave-synchronous code): formally a wavefront-scope
// release/acquire pair around a wave barrier; costs no hardware barrier.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// a wave-uniform double the compiler computed with the vector ALU -> an SGPR pair (no VGPR held across loops, no spill)
__device__ __forceinline__ double uniform_f64(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int srcLane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), srcLane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), srcLane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int srcLane) {
  unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(v & 0xffffffffull), srcLane);
  unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(v >> 32), srcLane);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// ---- DPP wave reductions (gfx9 row_shr / row_bcast): ~10x cheaper than ds_bpermute-based shuffles ----------
// Pattern of rocPRIM's warp_reduce_dpp: quad swaps, row_shr:4/8 inside each 16-lane row, then row_bcast:15 / :31
// across rows; the full-wave result lands in lane 63 and is broadcast with v_readlane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, true);  // bound_ctrl: invalid source lanes read 0
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
  return __hiloint2double(hi, lo);
}
// Sum over the 64 lanes, result returned in every lane.  Lanes that receive no partner add +0.0.
__device__ __forceinline__ double wave_sum_dpp(double v) {
  v += dpp_f64<0xb1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4e, 0xf>(v);   // quad_perm [2,3,0,1]
  v += dpp_f64<0x114, 0xf>(v);  // row_shr:4
  v += dpp_f64<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of each row holds the row sum
  v += dpp_f64<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_f64<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
  return readlane_f64(v, 63);
}
__device__ __forceinline__ int wave_sum_i_dpp(int v) {
  v += dpp_i32<0xb1, 0xf>(v);
  v += dpp_i32<0x4e, 0xf>(v);
  v += dpp_i32<0x114, 0xf>(v);
  v += dpp_i32<0x118, 0xf>(v);
  v += dpp_i32<0x142, 0xa>(v);
  v += dpp_i32<0x143, 0xc>(v);
  return __builtin_amdgcn_readlane(v, 63);
}

// min / max of a u32 over the wave with DPP moves (lanes without a partner keep their own value)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_keep_u32(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  v = max(v, dpp_keep_u32<0xb1, 0xf>(v));
  v = max(v, dpp_keep_u32<0x4e, 0xf>(v));
  v = max(v, dpp_keep_u32<0x114, 0xf>(v));
  v = max(v, dpp_keep_u32<0x118, 0xf>(v));
  v = max(v, dpp_keep_u32<0x142, 0xa>(v));
  v = max(v, dpp_keep_u32<0x143, 0xc>(v));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) { return ~wave_max_u32(~v); }
// min / max of a float over the wave (exact in any order; minNum / maxNum: a NaN loses against a number), OR of a 64-bit mask:
// DPP moves, result in every lane (r1-r3: six ds_bpermute shuffles each -- an LDS round trip per step)
#define RFS_DPP_KEEP(x, CTRL, MASK) __builtin_amdgcn_update_dpp((x), (x), CTRL, MASK, 0xf, false)
#define RFS_DPP_F32(CTRL, MASK) __builtin_bit_cast(float, RFS_DPP_KEEP(__builtin_bit_cast(int, v), CTRL, MASK))
__device__ __forceinline__ float wave_max_f32(float v) {
  v = __builtin_fmaxf(v, RFS_DPP_F32(0xb1, 0xf));
  v = __builtin_fmaxf(v, RFS_DPP_F32(0x4e, 0xf));
  v = __builtin_fmaxf(v, RFS_DPP_F32(0x114, 0xf));
  v = __builtin_fmaxf(v, RFS_DPP_F32(0x118, 0xf));
  v = __builtin_fmaxf(v, RFS_DPP_F32(0x142, 0xa));
  v = __builtin_fmaxf(v, RFS_DPP_F32(0x143, 0xc));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_min_f32(float v) {
  v = __builtin_fminf(v, RFS_DPP_F32(0xb1, 0xf));
  v = __builtin_fminf(v, RFS_DPP_F32(0x4e, 0xf));
  v = __builtin_fminf(v, RFS_DPP_F32(0x114, 0xf));
  v = __builtin_fminf(v, RFS_DPP_F32(0x118, 0xf));
  v = __builtin_fminf(v, RFS_DPP_F32(0x142, 0xa));
  v = __builtin_fminf(v, RFS_DPP_F32(0x143, 0xc));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max_f32_dpp(float v) { return wave_max_f32(v); }
#undef RFS_DPP_F32
__device__ __forceinline__ unsigned wave_or_u32(unsigned u) {
  int x = (int)u;
  x |= RFS_DPP_KEEP(x, 0xb1, 0xf);
  x |= RFS_DPP_KEEP(x, 0x4e, 0xf);
  x |= RFS_DPP_KEEP(x, 0x114, 0xf);
  x |= RFS_DPP_KEEP(x, 0x118, 0xf);
  x |= RFS_DPP_KEEP(x, 0x142, 0xa);
  x |= RFS_DPP_KEEP(x, 0x143, 0xc);
  return (unsigned)__builtin_amdgcn_readlane(x, 63);
}
__device__ __forceinline__ unsigned long long wave_or_u64(unsigned long long v) {
  return ((unsigned long long)wave_or_u32((unsigned)(v >> 32)) << 32) | wave_or_u32((unsigned)v);
}
#undef RFS_DPP_KEEP

// exclusive prefix sum over the wave (small ints): Hillis-Steele inside each 16-lane row with row_shr DPP moves, then the row
// totals across rows with row_bcast:15 / :31 -- six VALU instructions, no LDS permute and no lane-address registers (the
// ds_bpermute form kept six shuffle addresses alive across whole kernels, which the fused kernel paid for in scratch)
__device__ __forceinline__ int wave_excl_scan(int v, int lane) {
  (void)lane;
  int x = v;
  x += dpp_i32<0x111, 0xf>(x);  // row_shr:1
  x += dpp_i32<0x112, 0xf>(x);  // row_shr:2
  x += dpp_i32<0x114, 0xf>(x);  // row_shr:4
  x += dpp_i32<0x118, 0xf>(x);  // row_shr:8  -> inclusive scan inside every row
  x += dpp_i32<0x142, 0xa>(x);  // row_bcast:15 into rows 1 and 3
  x += dpp_i32<0x143, 0xc>(x);  // row_bcast:31 into rows 2 and 3
  return x - v;
}

// ---- MeasurementModel_RngBrg pieces (reference src/MeasurementModel_RngBrg.cpp) -----------------------
struct PoseReg {
  double x, y, th;
  double P[9];
};

__device__ __forceinline__ void load_pose(const Buffers &B, const Params &P, int i, PoseReg &pr) {
  pr.x = B.pose[3 * i + 0];
  pr.y = B.pose[3 * i + 1];
  pr.th = B.pose[3 * i + 2];
  const double *pc = B.poseCov + (size_t)P.poseCovStride * i;
#pragma unroll
  for (int k = 0; k < 9; k++) pr.P[k] = pc[k];
}

// probabilityOfDetection(): src/MeasurementModel_RngBrg.cpp:138-167
__device__ __forceinline__ double rb_pd(const Params &P, double range, bool &close) {
  close = false;
  double pd;
  if (range <= P.rmax && range >= P.rmin) {
    pd = P.Pd;
    if (range >= P.rmaxIn || range <= P.rminIn) close = true;
  } else {
    pd = 0;
    if (range <= P.rmaxOut && range >= P.rminOut) close = true;
  }
  return pd;
}

// while(a > PI) a -= 2PI; while(a < -PI) a += 2PI;  -- the reference's wrap (src/KalmanFilter_RngBrg.cpp:58-61).
// First iteration branch-free (the common case: both bearings lie in [-pi, pi]); further iterations only if needed.
// Angles beyond 64 pi -- the heading of a particle whose Ackerman step divided by ~0, nothing a sane pose or bearing produces --
// are first reduced with one rounded division instead of up to 10^7 subtractions (measured: ONE such particle turned a 60 us
// Victoria Park map update into a 31 ms one); an infinite angle, on which the reference's loop never ends, becomes NaN and is
// rejected by every gate downstream.  Up to 64 pi the loop below is the reference's, subtraction by subtraction.
__device__ __forceinline__ double wrap_pi(double a) {
  if (fabs(a) > 64.0 * RFS_PI) a = a - (2 * RFS_PI) * rint(a / (2 * RFS_PI));
  a = (a > RFS_PI) ? a - 2 * RFS_PI : a;
  if (a > RFS_PI) { do { a -= 2 * RFS_PI; } while (a > RFS_PI); }
  a = (a < -RFS_PI) ? a + 2 * RFS_PI : a;
  if (a < -RFS_PI) { do { a += 2 * RFS_PI; } while (a < -RFS_PI); }
  return a;
}

// Expected measurement + innovation covariance for landmark (mx,my,Sigma) seen from pose:
// measure(): src/MeasurementModel_RngBrg.cpp:70-115.  Sigma packed (sxx,sxy,syy).
struct MeasOut {
  double z0, z1;            // expected range, bearing
  double h00, h01, h10, h11;  // H_lmk
  double s00, s01, s10, s11;  // S
  double range;
  bool inRange;             // measure() return value
};
__device__ __forceinline__ void rb_measure(const Params &P, const PoseReg &pr, double mx, double my, double sxx, double sxy, double syy,
                                           MeasOut &o) {
  double dx = mx - pr.x, dy = my - pr.y;
  double range2 = dx * dx + dy * dy;
  double range = sqrt(range2);
  o.range = range;
  o.z0 = range;
  o.z1 = wrap_pi(atan2(dy, dx) - pr.th);
  o.h00 = dx / range;   o.h01 = dy / range;
  o.h10 = -dy / range2; o.h11 = dx / range2;
  // A = H * Sigma * H^T
  double t00 = o.h00 * sxx + o.h01 * sxy, t01 = o.h00 * sxy + o.h01 * syy;
  double t10 = o.h10 * sxx + o.h11 * sxy, t11 = o.h10 * sxy + o.h11 * syy;
  double a00 = t00 * o.h00 + t01 * o.h01, a01 = t00 * o.h10 + t01 * o.h11;
  double a10 = t10 * o.h00 + t11 * o.h01, a11 = t10 * o.h10 + t11 * o.h11;
  // B = Hr * Ppose * Hr^T, Hr = [-h00 -h01 0; -h10 -h11 -1]
  double r00 = -o.h00, r01 = -o.h01, r10 = -o.h10, r11 = -o.h11;
  double u0[3], u1[3];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    // (0.0 * P: kept for the reference's inf / NaN propagation; per-particle uniform, so it lives in SGPRs)
    u0[j] = r00 * pr.P[j] + r01 * pr.P[3 + j] + uniform_f64(0.0 * pr.P[6 + j]);
    u1[j] = r10 * pr.P[j] + r11 * pr.P[3 + j] + (-1.0) * pr.P[6 + j];
  }
  double b00 = u0[0] * r00 + u0[1] * r01 + u0[2] * 0.0;
  double b01 = u0[0] * r10 + u0[1] * r11 + u0[2] * (-1.0);
  double b10 = u1[0] * r00 + u1[1] * r01 + u1[2] * 0.0;
  double b11 = u1[0] * r10 + u1[1] * r11 + u1[2] * (-1.0);
  o.s00 = (a00 + b00) + P.R[0];
  o.s01 = (a01 + b01) + P.R[1];
  o.s10 = (a10 + b10) + P.R[2];
  o.s11 = (a11 + b11) + P.R[3];
  o.inRange = !(range > P.rmax || range < P.rmin);
}

// Gaussian pdf pieces of a 2x2 covariance: inverse (Eigen closed form) and sqrt((2pi)^2 det).
__device__ __forceinline__ void inv2(double s00, double s01, double s10, double s11, double &i00, double &i01, double &i10, double &i11,
                                     double &det) {
  det = s00 * s11 - s10 * s01;
  double invdet = 1.0 / det;
  i00 = s11 * invdet;
  i10 = -s10 * invdet;
  i01 = -s01 * invdet;
  i11 = s00 * invdet;
}
__device__ __forceinline__ double pdf_factor2(double det) { return sqrt((2 * RFS_PI) * (2 * RFS_PI) * det); }

// exp(x) for the Gaussian likelihoods of the hot loops (the one transcendental they are made of: ~100 wave-level evaluations per
// particle and step).  Argument reduction x = k ln2 + r (|r| <= ln2 / 2, two-constant Cody-Waite), Taylor polynomial of degree
// 11 by Horner, v_ldexp: 17 VALU instructions against ~35 of the library routine (whose extra work -- special-case selects and a
// coefficient table re-materialised with v_mov at every call site -- is what the loops were spending their issue slots on).
// Relative error <= 1e-14 (truncation 0.3466^12 / 12! = 6e-15 + Horner rounding): four orders inside the tightest tolerance
// (1e-10 on mixture components).
#ifndef RFS_FAST_EXP
#define RFS_FAST_EXP 1
#endif
// A double constant held in an SGPR pair: the two s_mov_b32 are opaque to the optimiser (it can hoist and share them, but not
// turn them back into a vector-register constant), and a scalar operand costs the VALU instruction that uses it nothing.
constexpr unsigned double_bits_lo(double v) { return (unsigned)(__builtin_bit_cast(unsigned long long, v) & 0xffffffffull); }
constexpr unsigned double_bits_hi(double v) { return (unsigned)(__builtin_bit_cast(unsigned long long, v) >> 32); }
template <unsigned LO, unsigned HI>
__device__ __forceinline__ double sgpr_const_f64() {
  int lo, hi;
  asm("s_mov_b32 %0, %1" : "=s"(lo) : "n"(LO));
  asm("s_mov_b32 %0, %1" : "=s"(hi) : "n"(HI));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rfs_exp(double x) {
#if RFS_FAST_EXP
  // branch-free (a wave-uniform escape to the library routine would split the callers' unrolled pair loops into separate
  // blocks and cost them their instruction-level parallelism): the argument is clamped to [-760, 710] -- exp is exactly 0
  // below and +inf above that range in fp64, and the reduction below returns exactly that at the clamps -- and a NaN argument
  // is passed through at the end.
  const double xin = x;
  x = fmin(fmax(x, -760.0), 710.0);
  // The coefficients live in SGPR pairs and enter the fused multiply-adds as the scalar operand: written as inline asm because
  // the compiler, left to itself, copies every coefficient into a vector register first (v_mov x 2 per term, or 24 VGPRs
  // hoisted around the unrolled loops and spilled).
#define RFS_C(v) sgpr_const_f64<double_bits_lo(v), double_bits_hi(v)>()
#define RFS_FMA_S(acc, x_, c_) asm("v_fma_f64 %0, %1, %2, %3" : "=v"(acc) : "v"(acc), "v"(x_), "s"(c_))
#define RFS_MUL_S(out, x_, c_) asm("v_mul_f64 %0, %1, %2" : "=v"(out) : "v"(x_), "s"(c_))
  double kx;
  { const double c = RFS_C(1.4426950408889634074); RFS_MUL_S(kx, x, c); }
  const double k = __builtin_rint(kx);
  double r = x;
  { const double c = RFS_C(-6.93147180369123816490e-01); asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(k), "s"(c), "v"(r)); }
  { const double c = RFS_C(-1.90821492927058770002e-10); asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(k), "s"(c), "v"(r)); }
  double p;
  { const double c11 = RFS_C(2.5052108385441718775e-08), c10 = RFS_C(2.7557319223985890653e-07);   // 1/11!, 1/10!
    p = c11; RFS_FMA_S(p, r, c10); }
  { const double c = RFS_C(2.7557319223985892511e-06); RFS_FMA_S(p, r, c); }  // 1/9!
  { const double c = RFS_C(2.4801587301587301566e-05); RFS_FMA_S(p, r, c); }  // 1/8!
  { const double c = RFS_C(1.9841269841269841253e-04); RFS_FMA_S(p, r, c); }  // 1/7!
  { const double c = RFS_C(1.3888888888888889419e-03); RFS_FMA_S(p, r, c); }  // 1/6!
  { const double c = RFS_C(8.3333333333333332177e-03); RFS_FMA_S(p, r, c); }  // 1/5!
  { const double c = RFS_C(4.1666666666666664354e-02); RFS_FMA_S(p, r, c); }  // 1/4!
  { const double c = RFS_C(1.6666666666666665741e-01); RFS_FMA_S(p, r, c); }  // 1/3!
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
#undef RFS_C
#undef RFS_FMA_S
#undef RFS_MUL_S
  const double res = __builtin_amdgcn_ldexp(p, (int)k);
  return (xin != xin) ? xin : res;
#else
  return exp(x);
#endif
}

// exp(-0.5*md2)/factor with the reference's NaN->0 guard (include/RandomVec.hpp:417-434).
// md2 > 1500 => exp(-750) is exactly 0 in fp64, so the transcendental is skipped (bit-identical result).
// the same with the reciprocal of the factor formed once by the caller (intensity sums: one factor per Gaussian, many points)
__device__ __forceinline__ double gauss_from_md2_r(double md2, double rfactor) {
  if (md2 > 1500.0) return 0.0;
  double l = rfs_exp(-0.5 * md2) * rfactor;
  if (l != l) l = 0.0;
  return l;
}
__device__ __forceinline__ double gauss_from_md2(double md2, double factor) {
  if (md2 > 1500.0) return 0.0;
  double l = rfs_exp(-0.5 * md2) / factor;
  if (l != l) l = 0.0;
  return l;
}
