// Single-index dense form of the Koopmans operator: Tw = 1 + beta * (H @ w^theta)^(1/theta) with a
// materialised N x N matrix H.  Reference: code/ssy/discrete/temp_ssy.py:109-159 (compute_H_single_index,
// single_index_T) and :204-216 (the analytic Jacobian beta * diag(F) H diag(G) - I) -- "only purpose is for
// cross-checking solutions produced by the multi-index code" (:14-16).  Same role here: an independent
// route (one GEMV instead of factorised passes) to the same numbers for N up to a few 10^4.
// GEMV is HBM-bound on H (8 N^2 bytes per application); one 256-thread workgroup per row.
#pragma once
#include <hip/hip_runtime.h>
#include "pass_kernel.hpp"

namespace sdfs {

struct DenseIO {
  const double* H;        // [N][N] row-major
  const double* x;        // T modes: w^theta;  JVP: v
  const double* c1;       // JVP: w^(theta-1)
  const double* c2_in;    // JVP: beta u / S
  double* c2_out;         // T_LIN
  double* out;
  const double* old;      // T modes: residual against this vector (or null);  JVP: v for "- v"
  unsigned long long* resid;
  const unsigned long long* gate;
  double gate_tol;
  long long N;
  double beta, inv_theta;
  int minus_identity;
};

// x = w^theta (and c1 = x / w for the linearisation); whole waves, uniform trip count
template <bool LIN>
__global__ void __launch_bounds__(256) dense_pow_kernel(const double* __restrict__ w, double* __restrict__ x,
                                                        double* __restrict__ c1, long long N, double theta,
                                                        const unsigned long long* gate, double gate_tol) {
  if (gate != nullptr && *gate <= (unsigned long long)__double_as_longlong(gate_tol)) return;
  const PowLane PT = pow_lane_init(threadIdx.x & 63);
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long trips = (N + stride - 1) / stride;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long long t = 0; t < trips; ++t, i += stride) {
    const bool valid = i < N;
    const double xin[1] = {valid ? w[i] : 1.0};
    double r[1];
    pow_fast_n<true, 1>(xin, theta, PT, r);
    if (valid) {
      x[i] = r[0];
      if (LIN) c1[i] = r[0] / xin[0];
    }
  }
}

enum DenseMode { D_T = 0, D_TLIN = 1, D_JVP = 2 };

template <int MODE>
__global__ void __launch_bounds__(256) dense_gemv_kernel(const DenseIO io) {
  __shared__ double red[4];
  if (io.gate != nullptr && *io.gate <= (unsigned long long)__double_as_longlong(io.gate_tol)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long m = blockIdx.x;
  const double* __restrict__ row = io.H + m * io.N;
  double acc = 0.0;
  for (long long j = tid; j < io.N; j += 256) {
    const double xv = (MODE == D_JVP) ? io.c1[j] * io.x[j] : io.x[j];
    acc = fma(row[j], xv, acc);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (wave != 0) return;
  const double S = (red[0] + red[1]) + (red[2] + red[3]);
  if (MODE == D_JVP) {
    if (lane == 0) {
      double r = io.c2_in[m] * S;
      if (io.minus_identity) r -= io.old[m];
      io.out[m] = r;
    }
    return;
  }
  const PowLane PT = pow_lane_init(lane);
  const double Sv[1] = {S};
  double u[1];
  pow_fast_n<false, 1>(Sv, io.inv_theta, PT, u);
  if (lane == 0) {
    const double tw = 1.0 + io.beta * u[0];
    if (io.resid != nullptr) {
      double r = fabs(tw - io.old[m]);
      if (!(r == r)) r = __longlong_as_double(0x7ff0000000000000LL);
      const unsigned long long rb = (unsigned long long)__double_as_longlong(r);
      if (rb > *(volatile unsigned long long*)io.resid) atomicMax(io.resid, rb);
    }
    io.out[m] = tw;
    if (MODE == D_TLIN) io.c2_out[m] = io.beta * u[0] / S;
  }
}

}  // namespace sdfs
