// vec_kernels.hpp -- fused BLAS-1 kernels for the Krylov / Anderson loops.
//
// They replace the vector arithmetic that jax.scipy.sparse.linalg.bicgstab
// (called at code/solvers.py:91-93) and jaxopt.AndersonAcceleration
// (code/solvers.py:104-114) generate through XLA.  All are HBM-bound streams:
// 16-byte accesses, grid-stride, wave-shuffle + LDS reductions, deterministic
// two-stage sums (per-block partials, then one finishing block that also does the
// scalar recurrences so no scalar ever round-trips to the host inside an iteration).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "anderson_step.hpp"
#include "wave_reduce.hpp"

namespace sdfs {

constexpr int VEC_BLOCK = 256;
constexpr int MAX_PARTIAL_BLOCKS = 2048;

// device scalar block of the BiCGSTAB recurrence
enum {
  SC_RHO = 0, SC_ALPHA, SC_OMEGA, SC_RHO_NEW, SC_BETA, SC_RHAT_Q, SC_SS, SC_TS, SC_TT,
  SC_RR, SC_BB, SC_ATOL2, SC_EARLY, SC_BREAK, SC_STEPMAX, SC_ITERS, SC_COUNT = 16
};

// Gate of the BiCGSTAB kernels: a device word that is non-zero while the inner solve runs and 0 once it has
// converged or broken down (set by k_bicg_iter_finish).  Every kernel of an iteration returns at once when
// it reads 0, so the host can enqueue several iterations per synchronisation (or replay them from a hipGraph)
// and the launches behind the last real iteration are no-ops: same iterates as one sync per iteration.
#define SDFS_GATED(gate) do { if ((gate) != nullptr && *(gate) == 0ULL) return; } while (0)

// (every lane active; the result is wave-uniform)
__device__ __forceinline__ double wave_sum(double v) { return wave_sum_f64(v); }
__device__ __forceinline__ double wave_max(double v) { return wave_max_f64(v); }

// block-level sum of up to NV values per thread -> partial[blockIdx.x + k*gridDim.x]
// (nv: streams in use, the rest is skipped)
template <int NV>
__device__ __forceinline__ void block_partials(const double (&v)[NV], double* __restrict__ partial, int nv = NV) {
  __shared__ double sm[NV][VEC_BLOCK / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    if (k < nv) {
      const double s = wave_sum(v[k]);
      if (lane == 0) sm[k][wave] = s;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      if (k < nv) {
        double s = 0.0;
        for (int w = 0; w < VEC_BLOCK / 64; ++w) s += sm[k][w];
        partial[blockIdx.x + (size_t)k * gridDim.x] = s;
      }
    }
  }
}

// sum `nb` partials of stream k inside ONE block (finishing kernels)
// nb <= 0: `partial` already holds the finished (multi-GPU: all-reduced) sums, one per stream
__device__ __forceinline__ double finish_sum(const double* __restrict__ partial, int nb, int k) {
  if (nb <= 0) return partial[k];
  __shared__ double sm[16];                  // (up to 1024 threads: the finishing kernels of the fused iteration sum 10^4 partials)
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) s += partial[i + (size_t)k * nb];
  s = wave_sum(s);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sm[w];
  return t;
}

// one block: out[k] = sum of partial stream k (the local sums a rank hands to the all-reduce)
__global__ void __launch_bounds__(VEC_BLOCK)
k_presum(const double* __restrict__ partial, int nb, int nsums, double* __restrict__ out,
         const unsigned long long* gate = nullptr) {
  SDFS_GATED(gate);
  for (int k = 0; k < nsums; ++k) {
    const double s = finish_sum(partial, nb, k);
    if (threadIdx.x == 0) out[k] = s;
    __syncthreads();
  }
}

// ---- generic helpers --------------------------------------------------------
// out = a - b ; partial sums of out.out ; block max of |out| folded into stepmax (optional)
__global__ void __launch_bounds__(VEC_BLOCK)
k_sub_dot(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out,
          long long n, double* __restrict__ partial) {
  double acc[1] = {0.0};
  for (long long i = (long long)blockIdx.x * VEC_BLOCK + threadIdx.x; i < n;
       i += (long long)gridDim.x * VEC_BLOCK) {
    const double d = a[i] - b[i];
    out[i] = d;
    acc[0] += d * d;
  }
  block_partials<1>(acc, partial);
}

// 16-byte packets: W consecutive elements per lane and access (2 doubles / 4 floats); a stream of
// another type rides along in as many 16-byte pieces as it needs.  Converted to double in registers.
template <typename T> struct PkW { static constexpr int W = 16 / (int)sizeof(T); };

// bf16r: an fp32 container whose STORES round to bfloat16 (round to nearest even on the upper 16 bits) -- the numerics
// of bf16 storage of the Krylov vectors and J.v streams (opts.krylov_f32 = 2, BASELINE config 5's bf16 question) at the
// bytes of fp32: what a 2-byte container would converge like, measured before anyone builds it.
__device__ __forceinline__ float round_bf16(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7f800000u) == 0x7f800000u) return f;           // Inf / NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return __uint_as_float(u & 0xffff0000u);
}
struct bf16r {
  float v;
  bf16r() = default;
  __device__ __forceinline__ explicit bf16r(double d) : v(round_bf16((float)d)) {}     // "(T)x": the value as it will be stored
  __device__ __forceinline__ explicit operator double() const { return (double)v; }
};
// SDFS_VEC_NT / SDFS_VEC_NT_F32 (build time, bit masks): the packets of the BLAS-1 streams as non-temporal requests -- bit 0
// loads, bit 1 stores.  Three builds side by side at GCY 20^6 (tools/ab_libs.sh, profiles/round4_vec_nt_ab.txt): fp64
// Newton 0.782 s plain, 0.765 with both, 0.777 with loads only; Anderson 0.317 / 0.304 / 0.310; Newton with fp32 vectors
// (krylov_f32 = 3) 0.560 / 0.539 / 0.532 -- half-size vectors still find part of themselves in the 256 MB Infinity Cache,
// so their stores stay cacheable.
#ifndef SDFS_VEC_NT
#define SDFS_VEC_NT 3
#endif
#ifndef SDFS_VEC_NT_F32
#define SDFS_VEC_NT_F32 1
#endif
// ... on vectors beyond the L2-resident sizes only: at SSY 15^4 (405 KB vectors) the non-temporal packets cost the graph-
// replayed Newton solve 3 % (13.5 against 13.0 ms)
constexpr long long VEC_NT_MIN = 1LL << 22;
typedef float vnt4f __attribute__((ext_vector_type(4)));
typedef double vnt2d __attribute__((ext_vector_type(2)));
template <int W>
__device__ __forceinline__ void ldv(const double* __restrict__ p, long long e, double (&v)[W], const bool nt) {
#pragma unroll
  for (int j = 0; j < W; j += 2) {
    const vnt2d t = ((SDFS_VEC_NT & 1) && nt) ? __builtin_nontemporal_load(reinterpret_cast<const vnt2d*>(p + e + j)) : *reinterpret_cast<const vnt2d*>(p + e + j);
    v[j] = t.x; v[j + 1] = t.y;
  }
}
template <int W>
__device__ __forceinline__ void ldv(const float* __restrict__ p, long long e, double (&v)[W], const bool nt) {
  static_assert(W == 4, "float packets hold four elements");
  const vnt4f t = ((SDFS_VEC_NT_F32 & 1) && nt) ? __builtin_nontemporal_load(reinterpret_cast<const vnt4f*>(p + e)) : *reinterpret_cast<const vnt4f*>(p + e);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <int W>
__device__ __forceinline__ void stv(double* __restrict__ p, long long e, const double (&v)[W], const bool nt) {
#pragma unroll
  for (int j = 0; j < W; j += 2) {
    const vnt2d t = {v[j], v[j + 1]};
    if ((SDFS_VEC_NT & 2) && nt) __builtin_nontemporal_store(t, reinterpret_cast<vnt2d*>(p + e + j)); else *reinterpret_cast<vnt2d*>(p + e + j) = t;
  }
}
template <int W>
__device__ __forceinline__ void stv(float* __restrict__ p, long long e, const double (&v)[W], const bool nt) {
  static_assert(W == 4, "float packets hold four elements");
  const vnt4f t = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  if ((SDFS_VEC_NT_F32 & 2) && nt) __builtin_nontemporal_store(t, reinterpret_cast<vnt4f*>(p + e)); else *reinterpret_cast<vnt4f*>(p + e) = t;
}
template <int W>
__device__ __forceinline__ void ldv(const bf16r* __restrict__ p, long long e, double (&v)[W], const bool nt) {
  ldv<W>(reinterpret_cast<const float*>(p), e, v, nt);
}
template <int W>
__device__ __forceinline__ void stv(bf16r* __restrict__ p, long long e, const double (&v)[W], const bool) {
  static_assert(W == 4, "float packets hold four elements");
  float4 t; t.x = round_bf16((float)v[0]); t.y = round_bf16((float)v[1]); t.z = round_bf16((float)v[2]); t.w = round_bf16((float)v[3]);
  *reinterpret_cast<float4*>(p + e) = t;
}
// grid-stride over packets, then over the < W leftover elements; BODY(e, W_) sees `e` (first element)
// and processes W_ elements through the LD / ST helpers below
#define SDFS_PACKET_LOOP(T, n, BODY)                                                        \
  {                                                                                         \
    constexpr int W = PkW<T>::W;                                                            \
    const long long gid_ = (long long)blockIdx.x * VEC_BLOCK + threadIdx.x;                 \
    const long long str_ = (long long)gridDim.x * VEC_BLOCK;                                \
    const long long np_ = (n) / W;                                                          \
    const bool nt_ = (n) > VEC_NT_MIN;                                                      \
    for (long long ip_ = gid_; ip_ < np_; ip_ += str_) { const long long e = ip_ * W; BODY(W) } \
    for (long long e = np_ * W + gid_; e < (n); e += str_) { BODY(1) }                      \
  }
template <int W>
__device__ __forceinline__ void LDx(const double* __restrict__ p, long long e, double (&v)[W], const bool nt) {
  if constexpr (W == 1) v[0] = p[e]; else ldv<W>(p, e, v, nt);
}
template <int W>
__device__ __forceinline__ void LDx(const float* __restrict__ p, long long e, double (&v)[W], const bool nt) {
  if constexpr (W == 1) v[0] = (double)p[e]; else ldv<W>(p, e, v, nt);
}
template <int W>
__device__ __forceinline__ void STx(double* __restrict__ p, long long e, const double (&v)[W], const bool nt) {
  if constexpr (W == 1) p[e] = v[0]; else stv<W>(p, e, v, nt);
}
template <int W>
__device__ __forceinline__ void STx(float* __restrict__ p, long long e, const double (&v)[W], const bool nt) {
  if constexpr (W == 1) p[e] = (float)v[0]; else stv<W>(p, e, v, nt);
}
template <int W>
__device__ __forceinline__ void LDx(const bf16r* __restrict__ p, long long e, double (&v)[W], const bool nt) {
  LDx<W>(reinterpret_cast<const float*>(p), e, v, nt);
}
template <int W>
__device__ __forceinline__ void STx(bf16r* __restrict__ p, long long e, const double (&v)[W], const bool nt) {
  if constexpr (W == 1) p[e].v = round_bf16((float)v[0]); else stv<W>(p, e, v, nt);
}
// the J.v kernels write fp32 streams: under bf16r emulation their output is rounded by a launch of its own
__global__ void __launch_bounds__(256) k_round_bf16(float* __restrict__ p, long long n, const unsigned long long* __restrict__ gate) {
  if (gate != nullptr && *gate == 0ULL) return;
  const long long n4 = n / 4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 t = reinterpret_cast<float4*>(p)[i];
    t.x = round_bf16(t.x); t.y = round_bf16(t.y); t.z = round_bf16(t.z); t.w = round_bf16(t.w);
    reinterpret_cast<float4*>(p)[i] = t;
  }
  for (long long i = n4 * 4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = round_bf16(p[i]);
}

// BiCGSTAB kernels.  T is the storage type of the Krylov vectors (double, or float for
// opts.krylov_f32): arithmetic and every reduction are fp64 either way.
// start: r = rhat = p = q = b (x0 = 0, so r0 = b); x = 0  (b is always fp64: it is g(x) = T(x) - x)
template <typename T>
__global__ void __launch_bounds__(VEC_BLOCK)
k_bicg_init(const double* __restrict__ b, T* __restrict__ r, T* __restrict__ rhat,
            T* __restrict__ p, T* __restrict__ q, T* __restrict__ x, long long n) {
#define BODY(W_) { double v[W_], z[W_]; LDx<W_>(b, e, v, nt_); _Pragma("unroll") for (int j = 0; j < W_; ++j) z[j] = 0.0; \
                   STx<W_>(r, e, v, nt_); STx<W_>(rhat, e, v, nt_); STx<W_>(p, e, v, nt_); STx<W_>(q, e, v, nt_); STx<W_>(x, e, z, nt_); }
  SDFS_PACKET_LOOP(T, n, BODY)
#undef BODY
}

// partial sums of <a, b>
template <typename T>
__global__ void __launch_bounds__(VEC_BLOCK)
k_dot(const T* __restrict__ a, const T* __restrict__ b, long long n, double* __restrict__ partial,
      const unsigned long long* gate = nullptr) {
  SDFS_GATED(gate);
  double acc[1] = {0.0};
#define BODY(W_) { double x_[W_], y_[W_]; LDx<W_>(a, e, x_, nt_); LDx<W_>(b, e, y_, nt_); \
                   _Pragma("unroll") for (int j = 0; j < W_; ++j) acc[0] += x_[j] * y_[j]; }
  SDFS_PACKET_LOOP(T, n, BODY)
#undef BODY
  block_partials<1>(acc, partial);
}

// one block: bb = sum(partials); atol2 = max(rtol^2 bb, atol^2); rho=alpha=omega=1;
// rho_new = <rhat, r> = bb; beta = rho_new/rho * alpha/omega; rr = bb
__global__ void __launch_bounds__(VEC_BLOCK)
k_bicg_init_finish(const double* __restrict__ partial, int nb, double* __restrict__ sc,
                   double rtol, double atol, unsigned long long* gate) {
  const double bb = finish_sum(partial, nb, 0);
  if (threadIdx.x == 0) {
    const double a2 = fmax(rtol * rtol * bb, atol * atol);
    sc[SC_ITERS] = 0.0;
    if (gate != nullptr) *gate = (bb > a2) ? ~0ULL : 0ULL;          // NaN: not greater -> closed (the host loop's test)
    sc[SC_BB] = bb;
    sc[SC_ATOL2] = fmax(rtol * rtol * bb, atol * atol);
    sc[SC_RHO] = 1.0; sc[SC_ALPHA] = 1.0; sc[SC_OMEGA] = 1.0;
    sc[SC_RHO_NEW] = bb;
    sc[SC_BETA] = bb;            // rho_new / 1 * 1 / 1
    sc[SC_RR] = bb;
    sc[SC_EARLY] = 0.0; sc[SC_BREAK] = 0.0;
  }
}

// p = r + beta (p - omega q)
template <typename T>
__global__ void __launch_bounds__(VEC_BLOCK)
k_bicg_update_p(const T* __restrict__ r, T* __restrict__ p, const T* __restrict__ q,
                long long n, const double* __restrict__ sc, const unsigned long long* gate = nullptr) {
  SDFS_GATED(gate);
  const double beta = sc[SC_BETA], omega = sc[SC_OMEGA];
#define BODY(W_) { double r_[W_], p_[W_], q_[W_]; LDx<W_>(r, e, r_, nt_); LDx<W_>((const T*)p, e, p_, nt_); LDx<W_>(q, e, q_, nt_); \
                   _Pragma("unroll") for (int j = 0; j < W_; ++j) p_[j] = r_[j] + beta * (p_[j] - omega * q_[j]); \
                   STx<W_>(p, e, p_, nt_); }
  SDFS_PACKET_LOOP(T, n, BODY)
#undef BODY
}

// alpha = rho_new / <rhat, q>
__global__ void __launch_bounds__(1024)
k_bicg_alpha_finish(const double* __restrict__ partial, int nb, double* __restrict__ sc,
                    const unsigned long long* gate = nullptr) {
  SDFS_GATED(gate);
  const double d = finish_sum(partial, nb, 0);
  if (threadIdx.x == 0) { sc[SC_RHAT_Q] = d; sc[SC_ALPHA] = sc[SC_RHO_NEW] / d; }
}

// s = r - alpha q (in place in r); partial sums of <s,s>
template <typename T>
__global__ void __launch_bounds__(VEC_BLOCK)
k_bicg_s(T* __restrict__ r, const T* __restrict__ q, long long n,
         const double* __restrict__ sc, double* __restrict__ partial, const unsigned long long* gate = nullptr) {
  SDFS_GATED(gate);
  const double alpha = sc[SC_ALPHA];
  double acc[1] = {0.0};
#define BODY(W_) { double r_[W_], q_[W_]; LDx<W_>((const T*)r, e, r_, nt_); LDx<W_>(q, e, q_, nt_); \
                   _Pragma("unroll") for (int j = 0; j < W_; ++j) { const double s = (double)(T)(r_[j] - alpha * q_[j]); r_[j] = s; acc[0] += s * s; } \
                   STx<W_>(r, e, r_, nt_); }
  SDFS_PACKET_LOOP(T, n, BODY)
#undef BODY
  block_partials<1>(acc, partial);
}

__global__ void __launch_bounds__(VEC_BLOCK)
k_bicg_s_finish(const double* __restrict__ partial, int nb, double* __restrict__ sc,
                const unsigned long long* gate = nullptr) {
  SDFS_GATED(gate);
  const double ss = finish_sum(partial, nb, 0);
  if (threadIdx.x == 0) { sc[SC_SS] = ss; sc[SC_EARLY] = (ss < sc[SC_ATOL2]) ? 1.0 : 0.0; }
}

// the same from MANY partial sums (one per wave tile of the fused first pass, krylov_kernels.hpp: 40 000 at GCY 20^6):
// 1024 threads, strided sums, wave shuffle + LDS
__global__ void __launch_bounds__(1024)
k_bicg_s_finish_wide(const double* __restrict__ partial, int nb, double* __restrict__ sc, const unsigned long long* gate = nullptr) {
  SDFS_GATED(gate);
  __shared__ double sm[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += 1024) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double ss = 0.0;
    for (int w = 0; w < 16; ++w) ss += sm[w];
    sc[SC_SS] = ss; sc[SC_EARLY] = (ss < sc[SC_ATOL2]) ? 1.0 : 0.0;
  }
}

// partial sums of <t,s> and <t,t>
template <typename T>
__global__ void __launch_bounds__(VEC_BLOCK)
k_dot2(const T* __restrict__ t, const T* __restrict__ s, long long n, double* __restrict__ partial,
       const unsigned long long* gate = nullptr) {
  SDFS_GATED(gate);
  double acc[2] = {0.0, 0.0};
#define BODY(W_) { double t_[W_], s_[W_]; LDx<W_>(t, e, t_, nt_); LDx<W_>(s, e, s_, nt_); \
                   _Pragma("unroll") for (int j = 0; j < W_; ++j) { acc[0] += t_[j] * s_[j]; acc[1] += t_[j] * t_[j]; } }
  SDFS_PACKET_LOOP(T, n, BODY)
#undef BODY
  block_partials<2>(acc, partial);
}

__global__ void __launch_bounds__(1024)
k_bicg_omega_finish(const double* __restrict__ partial, int nb, double* __restrict__ sc,
                    const unsigned long long* gate = nullptr) {
  SDFS_GATED(gate);
  const double ts = finish_sum(partial, nb, 0);
  const double tt = finish_sum(partial, nb, 1);
  if (threadIdx.x == 0) { sc[SC_TS] = ts; sc[SC_TT] = tt; sc[SC_OMEGA] = ts / tt; }
}

// x += alpha p (+ omega s);  r = s (- omega t);  partial sums of <r,r>, <rhat,r>
// (`s` lives in r on entry.)  early = s already below tolerance (JAX's exit_early select).
template <typename T>
__global__ void __launch_bounds__(VEC_BLOCK)
k_bicg_update_xr(T* __restrict__ x, T* __restrict__ r, const T* __restrict__ p,
                 const T* __restrict__ t, const T* __restrict__ rhat, long long n,
                 const double* __restrict__ sc, double* __restrict__ partial, const unsigned long long* gate = nullptr) {
  SDFS_GATED(gate);
  const double alpha = sc[SC_ALPHA];
  const bool early = sc[SC_EARLY] != 0.0;
  const double omega = early ? 0.0 : sc[SC_OMEGA];
  double acc[2] = {0.0, 0.0};
#define BODY(W_) { double x_[W_], r_[W_], p_[W_], t_[W_], h_[W_]; LDx<W_>((const T*)x, e, x_, nt_); LDx<W_>((const T*)r, e, r_, nt_); \
                   LDx<W_>(p, e, p_, nt_); LDx<W_>(t, e, t_, nt_); LDx<W_>(rhat, e, h_, nt_); \
                   _Pragma("unroll") for (int j = 0; j < W_; ++j) { \
                     const double s = r_[j]; \
                     const double rn = (double)(T)(early ? s : s - omega * t_[j]); \
                     x_[j] = x_[j] + alpha * p_[j] + omega * s; r_[j] = rn; \
                     acc[0] += rn * rn; acc[1] += h_[j] * rn; } \
                   STx<W_>(x, e, x_, nt_); STx<W_>(r, e, r_, nt_); }
  SDFS_PACKET_LOOP(T, n, BODY)
#undef BODY
  block_partials<2>(acc, partial);
}

// rr, next rho/beta and JAX's breakdown flags (k = -10 / -11 in its while_loop)
__global__ void __launch_bounds__(VEC_BLOCK)
k_bicg_iter_finish(const double* __restrict__ partial, int nb, double* __restrict__ sc,
                   unsigned long long* gate = nullptr) {
  SDFS_GATED(gate);
  const double rr = finish_sum(partial, nb, 0);
  const double rho_next = finish_sum(partial, nb, 1);
  if (threadIdx.x == 0) {
    const double rho_new = sc[SC_RHO_NEW], alpha = sc[SC_ALPHA], omega = sc[SC_OMEGA];
    double brk = 0.0;
    if (omega == 0.0 || alpha == 0.0) brk = 11.0;
    if (rho_new == 0.0) brk = 10.0;
    sc[SC_BREAK] = brk;
    sc[SC_RR] = rr;
    sc[SC_RHO] = rho_new;
    sc[SC_RHO_NEW] = rho_next;
    sc[SC_BETA] = rho_next / rho_new * alpha / omega;
    sc[SC_ITERS] += 1.0;
    // the inner solve goes on while |r|^2 > atol2 and nothing broke down (a NaN |r|^2 fails the comparison)
    if (gate != nullptr && !(rr > sc[SC_ATOL2] && brk == 0.0)) *gate = 0ULL;
  }
}

// ---- merged forms for small grids (launch-bound: one launch less per finishing kernel) --------------------------------
// Every workgroup finishes the sums it needs itself, in the same order, from the per-workgroup partials of the
// kernel before; workgroup 0 also leaves the scalars in `sc` for the later kernels.  The partial sums of the
// different reductions live in different regions of the buffer (a kernel reads one region while writing another).
// s = r - alpha q with alpha = rho_new / <rhat, q>  (k_bicg_alpha_finish + k_bicg_s)
template <typename T>
__global__ void __launch_bounds__(VEC_BLOCK)
k_bicg_s_m(T* __restrict__ r, const T* __restrict__ q, long long n, double* __restrict__ sc,
           const double* __restrict__ pdot, int nbdot, double* __restrict__ pss, const unsigned long long* gate) {
  SDFS_GATED(gate);
  const double d = finish_sum(pdot, nbdot, 0);
  const double alpha = sc[SC_RHO_NEW] / d;
  if (blockIdx.x == 0 && threadIdx.x == 0) { sc[SC_RHAT_Q] = d; sc[SC_ALPHA] = alpha; }
  double acc[1] = {0.0};
#define BODY(W_) { double r_[W_], q_[W_]; LDx<W_>((const T*)r, e, r_, nt_); LDx<W_>(q, e, q_, nt_); \
                   _Pragma("unroll") for (int j = 0; j < W_; ++j) { const double s = (double)(T)(r_[j] - alpha * q_[j]); r_[j] = s; acc[0] += s * s; } \
                   STx<W_>(r, e, r_, nt_); }
  SDFS_PACKET_LOOP(T, n, BODY)
#undef BODY
  block_partials<1>(acc, pss);
}

// x, r update with early = <s,s> < atol2 and omega = <t,s>/<t,t>  (k_bicg_s_finish + k_bicg_omega_finish + k_bicg_update_xr)
template <typename T>
__global__ void __launch_bounds__(VEC_BLOCK)
k_bicg_update_xr_m(T* __restrict__ x, T* __restrict__ r, const T* __restrict__ p,
                   const T* __restrict__ t, const T* __restrict__ rhat, long long n, double* __restrict__ sc,
                   const double* __restrict__ pss, int nbss, const double* __restrict__ pt, int nbt,
                   double* __restrict__ prr, const unsigned long long* gate) {
  SDFS_GATED(gate);
  const double ss = finish_sum(pss, nbss, 0);
  const double ts = finish_sum(pt, nbt, 0);
  const double tt = finish_sum(pt, nbt, 1);
  const double alpha = sc[SC_ALPHA];
  const bool early = ss < sc[SC_ATOL2];
  const double omega_full = ts / tt;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    sc[SC_SS] = ss; sc[SC_EARLY] = early ? 1.0 : 0.0; sc[SC_TS] = ts; sc[SC_TT] = tt; sc[SC_OMEGA] = omega_full;
  }
  const double omega = early ? 0.0 : omega_full;
  double acc[2] = {0.0, 0.0};
#define BODY(W_) { double x_[W_], r_[W_], p_[W_], t_[W_], h_[W_]; LDx<W_>((const T*)x, e, x_, nt_); LDx<W_>((const T*)r, e, r_, nt_); \
                   LDx<W_>(p, e, p_, nt_); LDx<W_>(t, e, t_, nt_); LDx<W_>(rhat, e, h_, nt_); \
                   _Pragma("unroll") for (int j = 0; j < W_; ++j) { \
                     const double s = r_[j]; \
                     const double rn = (double)(T)(early ? s : s - omega * t_[j]); \
                     x_[j] = x_[j] + alpha * p_[j] + omega * s; r_[j] = rn; \
                     acc[0] += rn * rn; acc[1] += h_[j] * rn; } \
                   STx<W_>(x, e, x_, nt_); STx<W_>(r, e, r_, nt_); }
  SDFS_PACKET_LOOP(T, n, BODY)
#undef BODY
  block_partials<2>(acc, prr);
}

// Newton update: x_new = x - step ; stepmax = max|step| (bits, atomicMax)
template <typename T>
__global__ void __launch_bounds__(VEC_BLOCK)
k_newton_update(const double* __restrict__ x, const T* __restrict__ step,
                double* __restrict__ xnew, long long n, unsigned long long* __restrict__ stepmax) {
  __shared__ double sm[VEC_BLOCK / 64];
  double mx = 0.0;
  for (long long i = (long long)blockIdx.x * VEC_BLOCK + threadIdx.x; i < n;
       i += (long long)gridDim.x * VEC_BLOCK) {
    const double st = (double)step[i];
    xnew[i] = x[i] - st;
    double a = fabs(st);
    if (!(a == a)) a = __longlong_as_double(0x7ff0000000000000LL);
    mx = fmax(mx, a);
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < VEC_BLOCK / 64; ++w) mx = fmax(mx, sm[w]);
    atomicMax(stepmax, (unsigned long long)__double_as_longlong(mx));
  }
}

// ---- Anderson ---------------------------------------------------------------
// r = fx - x stored into R[pos]; X[pos] = x; partial sums of <r, R[j]> for j < m
// (row `pos` of the Gram matrix; R[pos] itself is the fresh r).
// (AND_MAX_M, AndPtrs: anderson_step.hpp)

__global__ void __launch_bounds__(VEC_BLOCK)
k_and_push(const double* __restrict__ x, const double* __restrict__ fx, AndPtrs h, int m, int pos,
           long long n, double* __restrict__ partial, const unsigned long long* gate = nullptr) {
  SDFS_GATED(gate);
  double acc[AND_MAX_M];
#pragma unroll
  for (int j = 0; j < AND_MAX_M; ++j) acc[j] = 0.0;
  // every history stream is requested before the first one is used (a load inside the branch that consumes it
  // would cost one memory round trip per stream)
#define BODY(W_) { double x_[W_], f_[W_], r_[W_], o_[AND_MAX_M][W_]; LDx<W_>(x, e, x_, nt_); LDx<W_>(fx, e, f_, nt_); \
                   _Pragma("unroll") for (int j = 0; j < AND_MAX_M; ++j) { if (j < m && j != pos) LDx<W_>((const double*)h.R[j], e, o_[j], nt_); } \
                   _Pragma("unroll") for (int q = 0; q < W_; ++q) r_[q] = f_[q] - x_[q]; \
                   STx<W_>(h.X[pos], e, x_, nt_); STx<W_>(h.R[pos], e, r_, nt_); \
                   _Pragma("unroll") for (int j = 0; j < AND_MAX_M; ++j) { \
                     if (j < m) { \
                       if (j == pos) { _Pragma("unroll") for (int q = 0; q < W_; ++q) acc[j] += r_[q] * r_[q]; } \
                       else { _Pragma("unroll") for (int q = 0; q < W_; ++q) acc[j] += r_[q] * o_[j][q]; } } } }
  SDFS_PACKET_LOOP(double, n, BODY)
#undef BODY
  block_partials<AND_MAX_M>(acc, partial, m);
}

// Gram row from the per-workgroup partial sums: wave w of the (single) workgroup sums the streams w, w + 4, ...
// (one routine for the host-controlled and the device-resident loop: the Gram matrix is ill-conditioned and the
// iteration path follows its last bits, so both loops must add in the same order)
__device__ __forceinline__ void gram_row_sums(const double* __restrict__ partial, int nb, int m, double* __restrict__ row) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = wave; j < m; j += VEC_BLOCK / 64) {
    double s = 0.0;
    for (int i = lane; i < nb; i += 64) s += partial[i + (size_t)j * nb];
    s = wave_sum(s);
    if (lane == 0) row[j] = s;
  }
}
__global__ void __launch_bounds__(VEC_BLOCK)
k_and_push_finish(const double* __restrict__ partial, int nb, int m, double* __restrict__ gram_row) {
  gram_row_sums(partial, nb, m, gram_row);
}

struct AndCoef { double a[AND_MAX_M]; };
// x_next = sum_j alpha_j (X[j] + beta R[j])
__global__ void __launch_bounds__(VEC_BLOCK)
k_and_mix(AndPtrs h, AndCoef c, int m, double beta, double* __restrict__ xnext, long long n) {
#define BODY(W_) { double xa[W_], ra[W_]; \
                   _Pragma("unroll") for (int q = 0; q < W_; ++q) { xa[q] = 0.0; ra[q] = 0.0; } \
                   _Pragma("unroll") for (int j = 0; j < AND_MAX_M; ++j) { \
                     if (j < m) { double x_[W_], r_[W_]; LDx<W_>((const double*)h.X[j], e, x_, nt_); LDx<W_>((const double*)h.R[j], e, r_, nt_); \
                       _Pragma("unroll") for (int q = 0; q < W_; ++q) { xa[q] += c.a[j] * x_[q]; ra[q] += c.a[j] * r_[q]; } } } \
                   _Pragma("unroll") for (int q = 0; q < W_; ++q) xa[q] = xa[q] + beta * ra[q]; \
                   STx<W_>(xnext, e, xa, nt_); }
  SDFS_PACKET_LOOP(double, n, BODY)
#undef BODY
}

// ---- Anderson, device-resident control -------------------------------------------------------------------------
// The loop of solve_anderson (sdfs_api.hip) -- Gram matrix, the (m+1) x (m+1) solve of jaxopt's parametrisation
// (code/solvers.py:98-124), the rejection safeguard and the stopping test -- as state in device memory (AndState,
// and_step_wave: anderson_step.hpp), advanced by one single-workgroup kernel per iteration, so that the host enqueues
// (or replays from a hipGraph) whole chunks of iterations and synchronises once per chunk.  Sout = Sin: in place
// (every read of the state precedes the first write).
__global__ void __launch_bounds__(VEC_BLOCK)
k_and_step(const double* __restrict__ partial, int nb, int m, int pos, int rel, const AndState* Sin, AndState* Sout,
           double* __restrict__ err_slot, int* __restrict__ kind_slot, double tol, double max_iter, int mixing_freq, double ridge,
           const unsigned* __restrict__ flag = nullptr) {
  __shared__ AndStepLds sh;
  if (Sin->gate == 0ULL) {
    if (threadIdx.x < 64) and_state_carry((int)threadIdx.x, Sin, Sout);
    return;
  }
  gram_row_sums(partial, nb, m, sh.row);
  __syncthreads();
  if (threadIdx.x >= 64) return;         // the rest is one wave's work: LDS operations of a wave execute in order
  AndStepPar par;
  par.tol = tol; par.max_iter = max_iter; par.ridge = ridge; par.mixing_freq = mixing_freq;
  and_step_wave<AND_MAX_M + 1>(sh, (int)threadIdx.x, m, pos, rel, Sin, Sout, true, err_slot, kind_slot, par, flag != nullptr ? (int)(*flag != 0u) : -1);
}

// the update of x that pass `rel` decided on (runs also for the pass that ended the loop)
__global__ void __launch_bounds__(VEC_BLOCK)
k_and_mix_dev(AndPtrs h, const AndState* __restrict__ S, int m, double beta, double* __restrict__ x,
              const double* __restrict__ fx, int pos, int rel, long long n) {
  if (S->mix_rel != rel) return;
  const int mode = S->mix_mode;
  if (mode == 0) {
#define BODY(W_) { double f_[W_]; LDx<W_>(fx, e, f_, nt_); STx<W_>(x, e, f_, nt_); }
    SDFS_PACKET_LOOP(double, n, BODY)
#undef BODY
    return;
  }
  double ca[AND_MAX_M];
#pragma unroll
  for (int j = 0; j < AND_MAX_M; ++j) ca[j] = j < m ? S->coef[j] : 0.0;
  const double be = mode == 2 ? S->mix_beta : beta;
#define BODY(W_) { double xa[W_], ra[W_], xs_[AND_MAX_M][W_], rs_[AND_MAX_M][W_]; \
                   _Pragma("unroll") for (int j = 0; j < AND_MAX_M; ++j) { \
                     if (j < m && !(mode == 2 && j == pos)) { LDx<W_>((const double*)h.X[j], e, xs_[j], nt_); LDx<W_>((const double*)h.R[j], e, rs_[j], nt_); } } \
                   _Pragma("unroll") for (int q = 0; q < W_; ++q) { xa[q] = 0.0; ra[q] = 0.0; } \
                   _Pragma("unroll") for (int j = 0; j < AND_MAX_M; ++j) { \
                     if (j < m && !(mode == 2 && j == pos)) { \
                       _Pragma("unroll") for (int q = 0; q < W_; ++q) { xa[q] += ca[j] * xs_[j][q]; ra[q] += ca[j] * rs_[j][q]; } } } \
                   _Pragma("unroll") for (int q = 0; q < W_; ++q) xa[q] = xa[q] + be * ra[q]; \
                   STx<W_>(x, e, xa, nt_); \
                   if (mode == 2) { double z_[W_]; _Pragma("unroll") for (int q = 0; q < W_; ++q) z_[q] = 0.0; STx<W_>(h.R[pos], e, z_, nt_); } }
  SDFS_PACKET_LOOP(double, n, BODY)
#undef BODY
}

// ---- Anderson on large grids: the Gram matrix when a solve is due, not a row per pass ---------------------------------
// A pass of the loop above reads the whole residual history (m - 1 streams) for one row of the Gram matrix, yet the
// matrix is only used by the solve of every `mixing_freq`-th pass; the stopping test and the safeguard need the diagonal
// entry <r, r> alone.  At GCY 20^6 that row was 1.3 of the 3.1 ms of a pass.  Here: the push writes its slot
// (Y = x + beta r, see AndArgs, and r) and <r, r>; a pass whose step can mix recomputes the WHOLE matrix from the m
// residual streams in one sweep (m streams per mixing_freq passes instead of m - 1 per pass); the update reads the m
// streams Y_j.  Per four passes at m = 10: 71 grid streams instead of 107.
constexpr int AND_LAZY_M = 12;                                   // register budget of the sweep: m (m + 1) / 2 sums
constexpr int AND_LAZY_PAIRS = AND_LAZY_M * (AND_LAZY_M + 1) / 2;
constexpr int AND_LAZY_BLOCKS = 512;

__global__ void __launch_bounds__(VEC_BLOCK)
k_and_push_lite(const double* __restrict__ x, const double* __restrict__ fx, double* __restrict__ ypos, double* __restrict__ rpos,
                double beta, long long n, double* __restrict__ partial, const unsigned long long* gate) {
  SDFS_GATED(gate);
  double acc[1] = {0.0};
#define BODY(W_) { double x_[W_], f_[W_], r_[W_], y_[W_]; LDx<W_>(x, e, x_, nt_); LDx<W_>(fx, e, f_, nt_); \
                   _Pragma("unroll") for (int q = 0; q < W_; ++q) { r_[q] = f_[q] - x_[q]; y_[q] = fma(beta, r_[q], x_[q]); acc[0] = fma(r_[q], r_[q], acc[0]); } \
                   STx<W_>(ypos, e, y_, nt_); STx<W_>(rpos, e, r_, nt_); }
  SDFS_PACKET_LOOP(double, n, BODY)
#undef BODY
  block_partials<1>(acc, partial);
}

// does the step of the pass that is running mix?  (the state before that step; the same test as and_step_wave's, but for
// the finiteness of the new residual, which only ever cancels a mixing step)
__device__ __forceinline__ bool and_mix_due(const AndState* __restrict__ S, int m, int mixing_freq) {
  const double it1 = S->it + 1.0;
  return it1 >= m && it1 >= S->no_mix_until && ((long long)it1) % mixing_freq == 0;
}

// partial sums of <R_i, R_j>, i <= j < m, pair (i, j) at index i m - i (i - 1) / 2 + (j - i): partial[pair * gridDim.x + block]
__global__ void __launch_bounds__(VEC_BLOCK)
k_and_gram_full(AndPtrs h, int m, int mixing_freq, long long n, double* __restrict__ partial, const AndState* __restrict__ S) {
  if (S->gate == 0ULL || !and_mix_due(S, m, mixing_freq)) return;
  double acc[AND_LAZY_PAIRS];
#pragma unroll
  for (int p = 0; p < AND_LAZY_PAIRS; ++p) acc[p] = 0.0;
  for (long long e = ((long long)blockIdx.x * VEC_BLOCK + threadIdx.x) * 2; e < n; e += (long long)gridDim.x * VEC_BLOCK * 2) {
    double2 r[AND_LAZY_M];
    const bool two = e + 1 < n;
#pragma unroll
    for (int j = 0; j < AND_LAZY_M; ++j) {
      const double* const pj = h.R[j < m ? j : 0];
      r[j] = two ? *reinterpret_cast<const double2*>(pj + e) : make_double2(pj[e], 0.0);      // (n even or the tail: e is even, the streams 16-byte aligned)
    }
    int p = 0;
#pragma unroll
    for (int i = 0; i < AND_LAZY_M; ++i)
#pragma unroll
      for (int j = i; j < AND_LAZY_M; ++j, ++p)
        if (j < m) acc[p] = fma(r[i].y, r[j].y, fma(r[i].x, r[j].x, acc[p]));
  }
  // block sums: 16 lanes per DPP row, then rows and waves through LDS
  __shared__ double red[AND_LAZY_PAIRS * 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int p = 0; p < AND_LAZY_PAIRS; ++p) {
    double sj = acc[p];
    sj += dpp_mov_f64<0xB1>(sj); sj += dpp_mov_f64<0x4E>(sj); sj += dpp_mov_f64<0x141>(sj); sj += dpp_mov_f64<0x140>(sj);
    if ((lane & 15) == 0) red[p * 16 + wave * 4 + (lane >> 4)] = sj;
  }
  __syncthreads();
  for (int p = threadIdx.x; p < AND_LAZY_PAIRS; p += VEC_BLOCK) {
    double sj = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) sj += red[p * 16 + q];
    partial[(size_t)p * gridDim.x + blockIdx.x] = sj;
  }
}

// the control step of that loop: <r, r> of the pass, the fresh matrix where a solve is due, then and_step_wave in place
__global__ void __launch_bounds__(1024)          // (256 threads, or 1024 where <r, r> arrives as one partial sum per line tile)
k_and_step_lazy(const double* __restrict__ partial_rr, int nb, const double* __restrict__ gram_partial, int gb, int refresh,
                int m, int pos, int rel, AndState* S, double* __restrict__ err_slot, int* __restrict__ kind_slot,
                double tol, double max_iter, int mixing_freq, double ridge) {
  __shared__ AndStepLds sh;
  __shared__ double rr_s;
  if (S->gate == 0ULL) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool fresh = refresh != 0 && and_mix_due(S, m, mixing_freq);          // uniform
  {
    const double s = finish_sum(partial_rr, nb, 0);
    if (tid == 0) rr_s = s;
  }
  if (fresh) {
    // pair p = (i, j): wave w takes pairs w, w + 4, ...
    int p = 0;
    for (int i = 0; i < m; ++i)
      for (int j = i; j < AND_LAZY_M; ++j, ++p) {
        if (j >= m || (p & 3) != wave) continue;
        double sj = 0.0;
        for (int q = lane; q < gb; q += 64) sj += gram_partial[(size_t)p * gb + q];
        sj = wave_sum(sj);
        if (lane == 0) { S->G[i * m + j] = sj; S->G[j * m + i] = sj; }
      }
  }
  __threadfence_block();
  __syncthreads();
  if (tid >= 64) return;
  if (lane < AND_MAX_M) sh.row[lane] = lane == pos ? rr_s : (lane < m ? S->G[pos * m + lane] : 0.0);
  and_wsync();
  AndStepPar par;
  par.tol = tol; par.max_iter = max_iter; par.ridge = ridge; par.mixing_freq = mixing_freq;
  and_step_wave<AND_LAZY_M + 1, true>(sh, lane, m, pos, rel, S, S, true, err_slot, kind_slot, par);
}

// the update of x that pass `rel` decided on, history as Y_j = x_j + beta r_j
__global__ void __launch_bounds__(VEC_BLOCK)
k_and_mix_y(AndPtrs h, const AndState* __restrict__ S, int m, double beta, double* x, const double* fx,
            int pos, int rel, long long n) {
  if (S->mix_rel != rel) return;
  const int mode = S->mix_mode;
  if (mode == 0) {
    if (x == fx) return;                  // (the caller alternates its buffers: T x already sits where the next pass reads)
#define BODY(W_) { double f_[W_]; LDx<W_>(fx, e, f_, nt_); STx<W_>(x, e, f_, nt_); }
    SDFS_PACKET_LOOP(double, n, BODY)
#undef BODY
    return;
  }
  if (mode == 1) {
    double ca[AND_LAZY_M];
#pragma unroll
    for (int j = 0; j < AND_LAZY_M; ++j) ca[j] = j < m ? S->coef[j] : 0.0;
#define BODY(W_) { double xa[W_], ys_[AND_LAZY_M][W_]; \
                   _Pragma("unroll") for (int j = 0; j < AND_LAZY_M; ++j) { if (j < m) LDx<W_>((const double*)h.X[j], e, ys_[j], nt_); } \
                   _Pragma("unroll") for (int q = 0; q < W_; ++q) xa[q] = 0.0; \
                   _Pragma("unroll") for (int j = 0; j < AND_LAZY_M; ++j) { if (j < m) { _Pragma("unroll") for (int q = 0; q < W_; ++q) xa[q] = fma(ca[j], ys_[j][q], xa[q]); } } \
                   STx<W_>(x, e, xa, nt_); }
    SDFS_PACKET_LOOP(double, n, BODY)
#undef BODY
    return;
  }
  // rejected step: x = x_prev + r_prev = Y_prev + (1 - beta) R_prev; the poisoned slot is cleared
  const int prev = (int)S->prev_pos;
  const double* py = h.X[0];
  const double* pr = h.R[0];
#pragma unroll
  for (int j = 1; j < AND_LAZY_M; ++j)
    if (j == prev) { py = h.X[j]; pr = h.R[j]; }
#define BODY(W_) { double y_[W_], r_[W_], z_[W_]; LDx<W_>(py, e, y_, nt_); LDx<W_>(pr, e, r_, nt_); \
                   _Pragma("unroll") for (int q = 0; q < W_; ++q) { y_[q] = fma(1.0 - beta, r_[q], y_[q]); z_[q] = 0.0; } \
                   STx<W_>(x, e, y_, nt_); STx<W_>(yclr, e, y_, nt_); STx<W_>(rclr, e, z_, nt_); }
  double* yclr = h.X[0];
  double* rclr = h.R[0];
#pragma unroll
  for (int j = 1; j < AND_LAZY_M; ++j)
    if (j == pos) { yclr = h.X[j]; rclr = h.R[j]; }
  SDFS_PACKET_LOOP(double, n, BODY)
#undef BODY
}

// ---- streaming copy: the ceiling of a 16-byte-per-point pass on this box --------------------------------------------------
// One chunk of 256 x 8 16-byte units per workgroup, all eight loads of a lane in flight, non-temporal loads and stores --
// the fastest of the copy forms tools/probes/kernel_bench.hip measured (5.75-5.86 TB/s at 512 MB; grid-stride and
// default-policy forms 5.1-5.7, hipMemcpyAsync 4.6-5.5).  bench.py prints it as `copy_ceiling_GBps`.
constexpr int COPY_UNITS = 8;
__global__ void __launch_bounds__(VEC_BLOCK) k_stream_copy(const double* __restrict__ in, double* __restrict__ out, long long units, long long n) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && (n & 1)) out[n - 1] = in[n - 1];
  if (units == 0) return;
  const long long c0 = (long long)blockIdx.x * (VEC_BLOCK * COPY_UNITS) + threadIdx.x;
  const v2d* const src = reinterpret_cast<const v2d*>(in);
  v2d* const dst = reinterpret_cast<v2d*>(out);
  v2d v[COPY_UNITS];
  // (unconditional loads from clamped indices: a load inside `if (i < units)` makes the compiler wait for each one)
#pragma unroll
  for (int k = 0; k < COPY_UNITS; ++k) {
    const long long i = c0 + (long long)VEC_BLOCK * k;
    v[k] = __builtin_nontemporal_load(src + (i < units ? i : units - 1));
  }
#pragma unroll
  for (int k = 0; k < COPY_UNITS; ++k) {
    const long long i = c0 + (long long)VEC_BLOCK * k;
    if (i < units) __builtin_nontemporal_store(v[k], dst + i);
  }
}

// ---- re-shard pack / unpack (multi-GPU exchange buffers) -----------------------------------------------------------------
// grid = [outer][n_axis][inner] (C order), packed = concat_j [outer][size_j][inner] with block j = axis indices
// offs[j] .. offs[j+1].  One launch moves the whole shard (the host side used one strided copy per peer).  U = unit type
// (16 bytes when the inner run allows it, else one element); innerU = inner run in units.
constexpr int PACK_MAX_BLOCKS = 16;
struct PackBlocks { int n; unsigned off[PACK_MAX_BLOCKS + 1]; };
// SUB (unpack only): dst = unpacked - sub, `sub` laid out like dst -- the "- v" of a Krylov operator application
// (J - I) v folded into the pass that scatters the exchanged J v back (distributed.py: HipKrylov; three grid streams
// instead of the two of the unpack plus the five of a separate subtraction and copy)
__device__ __forceinline__ double2 unit_sub(const double2 a, const double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double unit_sub(const double a, const double b) { return a - b; }
__device__ __forceinline__ float unit_sub(const float a, const float b) { return a - b; }
__device__ __forceinline__ float4 unit_sub(const float4 a, const float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
template <typename U, bool UNPACK, bool SUB = false>
__global__ void __launch_bounds__(VEC_BLOCK)
k_pack_blocks(const U* __restrict__ src, U* __restrict__ dst, unsigned outer, unsigned n_axis, unsigned innerU, PackBlocks B,
              const U* __restrict__ sub = nullptr) {
  static_assert(!SUB || UNPACK, "the subtraction rides on the unpack");
  const unsigned total = outer * n_axis * innerU;          // < 2^31 (host check)
  for (unsigned e = blockIdx.x * VEC_BLOCK + threadIdx.x; e < total; e += gridDim.x * VEC_BLOCK) {
    const unsigned q = e / innerU, r = e - q * innerU;
    const unsigned o = q / n_axis, i = q - o * n_axis;
    int j = 0;
#pragma unroll
    for (int t = 1; t < PACK_MAX_BLOCKS; ++t) j += (t < B.n && i >= B.off[t]) ? 1 : 0;
    const unsigned sz = B.off[j + 1] - B.off[j];
    const unsigned p = (outer * B.off[j] + o * sz + (i - B.off[j])) * innerU + r;
    if (SUB) dst[e] = unit_sub(src[p], sub[e]);
    else if (UNPACK) dst[e] = src[p]; else dst[p] = src[e];
  }
}

}  // namespace sdfs
