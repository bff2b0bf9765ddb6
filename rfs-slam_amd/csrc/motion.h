// motion.h -- ParticleFilter::propagate (reference include/ParticleFilter.hpp:322-339) for the Victoria Park driver's process
// model on the device: MotionModel_Ackerman2d::step (src/ProcessModel_Ackerman2D.cpp:47-78) with ProcessModel::sample's
// input-noise branch (include/ProcessModel.hpp:126-150: every particle steps with its own input u + N(0, diag(var))).
//
// Why here: the poses are 24 B per particle and the step is a dozen flops, but at 5000 particles the host loop (two normal
// draws + sin/cos/tan per particle, then the push of the poses) is longer than the device work of an odometry message, and a
// Victoria Park run is 90 % odometry messages -- the host loop, not a kernel, set the driver's wall time.  The reference draws
// from one serial boost stream, which no parallel form reproduces; what is kept is the distribution.  The normal deviates come
// from a counter-based generator, Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3",
// SC'11) keyed by the caller's seed with counter (particle, call number) -- one block of four words gives the two deviates a
// particle needs (Box-Muller), so results do not depend on how particles map to threads, and a run is reproducible from its seed.
#pragma once
#include "common.h"

__host__ __device__ inline void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned out[4]) {
  for (int r = 0; r < 10; r++) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// (a, b) -> uniform in (0, 1]: 53 bits, never 0 (the logarithm below)
__host__ __device__ inline double philox_u01(unsigned a, unsigned b) {
  const unsigned long long m = (((unsigned long long)a << 32) | b) >> 11;
  return ((double)m + 1.0) * (1.0 / 9007199254740992.0);
}

struct AckermanStep {
  double uv, ur;          // input: speed, steering angle
  double sv, sr;          // standard deviations of the input noise (0: none)
  double dt;
  double h, l, dx, dy;    // vehicle geometry (MotionModel_Ackerman2d::setAckermanParams)
  unsigned long long seed, call;
};

// A run of steps (consecutive odometry messages, each with its own input, noise and interval) in one launch: every particle
// takes them one after the other, so the poses have the same bits as one launch per message.
#define ACKERMAN_RUN_MAX 16
struct AckermanRun {
  int n;
  double uv[ACKERMAN_RUN_MAX], ur[ACKERMAN_RUN_MAX], sv[ACKERMAN_RUN_MAX], sr[ACKERMAN_RUN_MAX], dt[ACKERMAN_RUN_MAX];
  double h, l, dx, dy;
  unsigned long long seed, call0;
};

__device__ __forceinline__ void ackerman_step_particle(double *pose, int i, const AckermanStep &A) {
  double uv = A.uv, ur = A.ur;
  if (A.sv != 0.0 || A.sr != 0.0) {
    unsigned r[4];
    philox4x32_10((unsigned)i, 0u, (unsigned)(A.call & 0xffffffffull), (unsigned)(A.call >> 32), (unsigned)(A.seed & 0xffffffffull), (unsigned)(A.seed >> 32), r);
    const double u1 = philox_u01(r[0], r[1]), u2 = philox_u01(r[2], r[3]);
    const double rad = sqrt(-2.0 * log(u1)), ang = 2.0 * RFS_PI * u2;
    uv += A.sv * (rad * cos(ang));
    ur += A.sr * (rad * sin(ang));
  }
  const double x = pose[3 * i], y = pose[3 * i + 1], th0 = pose[3 * i + 2];
  const double c = cos(th0), s = sin(th0), t = tan(ur);
  const double v = uv / (1 - t * A.h / A.l);
  pose[3 * i] = x + A.dt * (v * c - v / A.l * t * (A.dx * s + A.dy * c));
  pose[3 * i + 1] = y + A.dt * (v * s + v / A.l * t * (A.dx * c - A.dy * s));
  double th = th0 + A.dt * v / A.l * t;
  if (th > RFS_PI) th -= 2 * RFS_PI;
  else if (th < -RFS_PI) th += 2 * RFS_PI;
  pose[3 * i + 2] = th;
}

__global__ __launch_bounds__(256) void propagate_ackerman_kernel(double *pose, int N, AckermanStep A) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) ackerman_step_particle(pose, i, A);
}

__global__ __launch_bounds__(256) void propagate_ackerman_run_kernel(double *pose, int N, AckermanRun R) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  for (int k = 0; k < R.n; k++) {
    AckermanStep A;
    A.uv = R.uv[k]; A.ur = R.ur[k]; A.sv = R.sv[k]; A.sr = R.sr[k]; A.dt = R.dt[k];
    A.h = R.h; A.l = R.l; A.dx = R.dx; A.dy = R.dy;
    A.seed = R.seed; A.call = R.call0 + (unsigned long long)k;
    ackerman_step_particle(pose, i, A);
  }
}
