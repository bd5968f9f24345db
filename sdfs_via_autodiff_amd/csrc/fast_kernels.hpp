// fast_kernels.hpp -- the expectation kernels of the "pair plan" (DESIGN 4.1b).
//
// When every transition tensor of a model is unconditional (Rouwenhorst / Tauchen chains: all slices of
// the reference's conditional tensors are one matrix, checked bit for bit at create time) and the axes come
// in adjacent pairs of equal extent n in {16, 20, 24, 32}, one application of
//     sum_{next states} H * w^theta        (code/ssy/discrete/ssy_wc_ratio.py:143-145,
//                                           code/gcy/discrete/gcy_wc_ratio.py:230-232)
// is evaluated as D/2 passes of two contractions each:
//
//   slice_kernel  the two FASTEST axes: the grid is a sequence of contiguous n x n slices; a wave owns G
//                 consecutive slices in a private LDS region (no workgroup barrier anywhere), applies the
//                 prologue (w^theta, or c1 * v for the Jacobian-vector product), contracts both axes and
//                 streams the result out.  Reads and writes are contiguous.
//   line_kernel   a slower pair (X, Y): a tile is all n x n (x, y) rows of one 16-double chunk (exactly one
//                 128-byte line) of the contiguous remainder of the grid behind Y.  One workgroup per tile,
//                 two contractions, then -- in the last pass -- the Epstein-Zin aggregator
//                 Tw = 1 + beta (a3 S)^(1/theta) (ssy_wc_ratio.py:148, gcy_wc_ratio.py:235), the sup-norm
//                 residual of code/solvers.py:36 or the J.v scaling.
//
// Why this shape: on gfx950 an fp64 MFMA holds its SIMD for its whole 64 (16x16x4) or 20 (4x4x4) cycles and
// every VALU instruction of any wave on that SIMD -- integer ones included -- takes 4-5 cycles of the same
// issue slot (tools/probes/coissue_probe.hip: MFMA-only 2.56 M ticks + int-VALU-only 3.44 M ticks = 6.00 M
// ticks together).  A pass is therefore bound by the SUM of its MFMA and VALU issue cycles, and the generic
// kernel (pass_kernel.hpp) spends as many cycles on index arithmetic per column tile as on the MFMAs.  Here
// every extent and LDS stride is a compile-time constant: column tiles are unrolled, LDS addresses are one
// lane-constant register plus an immediate, Q fragments come straight from L2 into registers once per tile.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "anderson_step.hpp"
#include "pass_kernel.hpp"
#include "wave_reduce.hpp"

namespace sdfs {

// roles (compile time)
// S_MID: plain contraction of the pair (successive approximation, see L_TFUSED).
// L_TFUSED: the last pass of one application of T and the first pass of the next in one kernel: contract the pair,
// aggregator, residual, store Tw, then (Tw)^theta in the same LDS slots, the same two contractions again, and the
// tile goes out as the next application's intermediate.  With LineDesc::first_only only the second half runs (the
// prologue of the loop).  The contraction order of an application is free, so successive approximation on a 6-D
// grid runs [slices, plain] [lines, fused] per iteration, the fused pass alternating between the two line pairs.
// S_TFIRST32: the first pass of T with its OUTPUT (the intermediate between the passes) stored as scaled floats
// (opts.t_f32, BASELINE config 5): x = w^theta * 2^k with k from the mid-grid point, see t32_scale_of.
enum SliceMode { S_TFIRST = 0, S_TFIRST_LIN = 1, S_JFIRST = 2, S_MID = 3, S_TFIRST32 = 4, S_NMODES = 5 };
enum LineMode { L_MID = 0, L_TLAST = 1, L_TLAST_LIN = 2, L_JLAST = 3, L_TFUSED = 4, L_NMODES = 5 };

template <int N> struct MShape {
  static_assert(N == 16 || N == 20 || N == 24 || N == 32, "pair plan extents");
  static constexpr int N16 = N == 32 ? 2 : 1;
  static constexpr int N4 = (N - 16 * N16) / 4;
  static constexpr int KT = N / 4;
  static constexpr int A16 = N16, A4 = N4 > 0 ? N4 : 1;
};

// Q fragments of one n x n matrix (row major, y[i] = sum_I Q[i][I] x[I]) in MFMA A-operand order:
// 16x16x4: lane l holds A[row = l & 15][k = l >> 4];  4x4x4_4b: lane l holds A[row = l & 3][k = l >> 4]
// (the same A for its four column blocks).  Lane maps: tools/probes/mfma_f64_probe.hip.
template <int N> struct QFrag {
  double a16[MShape<N>::A16][MShape<N>::KT];
  double a4[MShape<N>::A4][MShape<N>::KT];
  __device__ __forceinline__ void load(const double* __restrict__ Q, int lane) {
    using S = MShape<N>;
    const int li = lane & 15, lk = lane >> 4, l4 = lane & 3;
#pragma unroll
    for (int kk = 0; kk < S::KT; ++kk) {
#pragma unroll
      for (int t = 0; t < S::N16; ++t) a16[t][kk] = Q[(16 * t + li) * N + 4 * kk + lk];
#pragma unroll
      for (int t = 0; t < S::N4; ++t) a4[t][kk] = Q[(16 * S::N16 + 4 * t + l4) * N + 4 * kk + lk];
    }
  }
};

// One column tile: y[:, 16 columns] = Q x[:, 16 columns], in place in LDS.  `p` points at this lane's
// element of row (lane >> 4) of the tile's column (lane & 15); RS = row stride in doubles (compile time,
// so every access below is p + immediate).  All N rows are read before any is written.
template <int N, int RS>
__device__ __forceinline__ void ctile(double* __restrict__ p, const QFrag<N>& q) {
  using S = MShape<N>;
  double b[S::KT];
#pragma unroll
  for (int kk = 0; kk < S::KT; ++kk) b[kk] = p[4 * kk * RS];
  v4d acc[S::A16];
  double d[S::A4];
#pragma unroll
  for (int t = 0; t < S::A16; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int t = 0; t < S::A4; ++t) d[t] = 0.0;
#pragma unroll
  for (int kk = 0; kk < S::KT; ++kk) {
#pragma unroll
    for (int t = 0; t < S::N16; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(q.a16[t][kk], b[kk], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < S::N4; ++t) d[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(q.a4[t][kk], b[kk], d[t], 0, 0, 0);
  }
  // 16x16x4 D: col = lane & 15, row = (lane >> 4) + 4 r;  4x4x4_4b D: row = lane >> 4, col = lane & 15
#pragma unroll
  for (int t = 0; t < S::N16; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) p[(16 * t + 4 * r) * RS] = acc[t][r];
  }
#pragma unroll
  for (int t = 0; t < S::N4; ++t) p[(16 * S::N16 + 4 * t) * RS] = d[t];
}

// fp32 storage of the linearisation (opts.krylov_f32, BASELINE config 5): c1 = w^(theta-1) is ~1e-50 and
// c2 ~1e+50 at theta = -16 .. -36 -- outside fp32 -- while only their product matters.  Both are stored scaled
// by an exact power of two taken from the point in the middle of the grid (c1 * 2^k, c2 * 2^-k,
// k = -ilogb(w_mid^(theta-1)): the spread (w / w_mid)^(theta-1) then splits evenly over fp32's range); the
// first pass reads w_mid from its input, the last pass from the grid it forms the residual against.
__device__ __forceinline__ double lin_scale_of(double wref, double theta, const PowLane& PT, bool inverse) {
  const double wr[1] = {wref};
  double xr[1];
  pow_fast_n<true, 1>(wr, theta, PT, xr);
  const int k = -ilogb(xr[0] / wref);
  return ldexp(1.0, inverse ? -k : k);
}

// fp32 intermediates of T (opts.t_f32): w^theta is ~1e-50 .. 1e-100 at theta = -16 .. -36 -- outside fp32 -- but T is
// linear between its two powers, so the first pass stores x * 2^k (k = -ilogb(w_mid^theta): the spread
// (w / w_mid)^theta then sits around 1) and the last pass multiplies its sums by 2^-k before the aggregator's power.
// Both derive k from the same value, the iterate at the mid-grid point (the last pass reads it from the grid it forms
// the residual against).
__device__ __forceinline__ double t32_scale_of(double wref, double theta, const PowLane& PT, bool inverse) {
  const double wr[1] = {wref};
  double xr[1];
  pow_fast_n<true, 1>(wr, theta, PT, xr);
  const int k = -ilogb(xr[0]);
  return ldexp(1.0, inverse ? -k : k);
}

// The power that opens the next application inside a fused kernel (L_TFUSED, SM_FUSED_T).  There y = 1 + beta u
// with u = ks^(1/theta) just computed, so
//     y^theta = (beta u)^theta (1 + 1/(beta u))^theta = beta^theta * ks * exp(theta * log1p(1 / (beta u))),
// and with beta u in the hundreds (wealth-consumption ratios) the exponent is a few hundredths: log1p through
// 2 atanh(1 / (2 beta u + 1)) and a degree-10 exponential in plain fp64 -- ~25 instructions instead of the
// ~50 of the general double-double power.  Valid (all truncation terms < 1e-17) for beta u >= max(8 |theta|, 256);
// returns false (wave-uniform) if any lane is outside that range or not finite: the caller then takes the general
// routine.  cbt = beta^theta.  The result differs from pow(fl(y), theta) by the rounding of y times theta, a few
// 1e-15 relative in x and |theta| times less in the next T w.
template <int N>
__device__ __forceinline__ bool next_power_fast(const double (&bu)[N], const double (&ks)[N], double theta, double cbt,
                                                double (&x)[N]) {
  const double lim = fmax(8.0 * fabs(theta), 256.0);
  bool okl = true;
#pragma unroll
  for (int j = 0; j < N; ++j) okl = okl && (bu[j] >= lim) && (bu[j] < 1e300);
  if (!__all(okl)) return false;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const double dd = fma(2.0, bu[j], 1.0);
    double z = __builtin_amdgcn_rcp(dd);
    z = fma(fma(-dd, z, 1.0), z, z);
    z = fma(fma(-dd, z, 1.0), z, z);
    const double z2 = z * z;
    const double L = 2.0 * z * fma(z2, fma(z2, 0.2, 1.0 / 3.0), 1.0);         // log1p(1 / bu), z <= 1/513
    const double a = theta * L;                                               // |a| <= 0.125
    double e = 1.0 / 3628800.0;
    e = fma(e, a, 1.0 / 362880.0);
    e = fma(e, a, 1.0 / 40320.0);
    e = fma(e, a, 1.0 / 5040.0);
    e = fma(e, a, 1.0 / 720.0);
    e = fma(e, a, 1.0 / 120.0);
    e = fma(e, a, 1.0 / 24.0);
    e = fma(e, a, 1.0 / 6.0);
    e = fma(e, a, 0.5);
    e = fma(e, a, 1.0);
    e = fma(e, a, 1.0);
    x[j] = (cbt * ks[j]) * e;
  }
  return true;
}

__device__ __forceinline__ void wave_lds_fence() {
  // LDS operations of one wave execute in issue order; this only keeps the compiler from moving
  // accesses of the wave-private region across a phase boundary
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------------------------
struct SliceDesc {
  long long nslices;        // number of contiguous n x n slices in the grid
  const double* Qf;         // matrix of the fastest axis (n x n)
  const double* Qe;         // matrix of the second-fastest axis
  double theta;
  long long ref_off;        // C-order offset of the mid-grid point (fp32 linearisation scale)
  double t32_ref;           // > 0: reference value of the t_f32 scale (sharded stages: the mid-grid point may live on another rank)
};

struct SliceIO {
  const double* in;         // T: w;  J.v: v
  double* out;
  const double* aux_in;     // J.v: c1
  double* aux_out;          // T + linearise: c1 = w^(theta-1)
  const unsigned long long* gate;
  double gate_tol;
  unsigned long long* zero; // if non-null: cleared by workgroup 0 (the residual word the last pass will atomicMax into)
  unsigned* sched;          // persistent form (stream_kernels.hpp): TK_WORDS scheduler words, zero between launches
};

// WV = waves per workgroup (every wave owns its tile: no barrier, so WV only sets how LDS is handed out), PAD20: the
// padded row stride for N = 20 as well (tools/probes/kernel_bench.hip measures both; the library uses the defaults)
// GX = slices per wave tile (0: the default; G N need not be a multiple of 16 -- the last column tile then computes a few
// columns twice, see slice_kernel).
template <int N, int WV = 4, bool PAD20 = false, int GX = 0> struct SliceGeo {
  static constexpr int G = GX > 0 ? GX : (N == 32 ? 2 : (N == 24 ? 2 : 4));     // slices per wave tile
  // LDS row stride in doubles.  The contraction over the fastest axis reads 16 rows at this stride at once: at a
  // stride of N doubles = 2N banks that is a 2-way bank conflict for N = 20 but 8-way for 16, 4-way for 24, 16-way for
  // 32 (64 banks of 4 bytes) -- the first pass ran at 0.35 of the HBM peak at GCY 16^6 against 0.52 at 20^6.  N + 2
  // makes the 16 rows conflict-free for every N; 20 keeps its linear image (padding it was measured: no gain).
  static constexpr int RS = (N == 20 && !PAD20) ? N : N + 2;
  static constexpr int LTILE = G * N * RS;                        // doubles of LDS per wave tile
  static constexpr int TILE = G * N * N;                          // doubles
  static constexpr int UNITS = TILE / 2;                          // double2 units
  static constexpr int EPT = (UNITS + 63) / 64;
  static constexpr int NCT = (G * N + 15) / 16;                   // column tiles of either contraction
  static constexpr int CPAD = NCT * 16 - G * N;                   // columns of the last tile beyond the wave tile: they repeat
                                                                  // the columns CPAD places back (same values to the same LDS slots)
  static constexpr int WAVES = WV;
  static constexpr int UNITS4 = TILE / 4;                         // float4 units (fp32 streams)
  static constexpr int EPT4 = (UNITS4 + 63) / 64;
  static_assert(CPAD <= 8, "the repeated columns lie inside the last column tile");
};

// F32 (opts.krylov_f32): J.v -- every stream (v, c1, the result) holds floats, 16-byte units of four;
// linearising T -- its own streams stay fp64, c1 is written as scaled floats.  Arithmetic stays fp64.
template <int N, int MODE, bool F32, int WV = 4, bool PAD20 = false, int GX = 0, int OCC = 3>
__global__ void __launch_bounds__(64 * WV, OCC)
slice_kernel(const SliceDesc P, const SliceIO io) {
  using Geo = SliceGeo<N, WV, PAD20, GX>;
  constexpr bool T32 = MODE == S_TFIRST32;                        // fp64 in, scaled floats out
  constexpr bool POWP = MODE == S_TFIRST || MODE == S_TFIRST_LIN || T32;
  constexpr bool LIN = MODE == S_TFIRST_LIN;
  constexpr bool MULP = MODE == S_JFIRST;
  constexpr bool JV32 = F32 && MULP;
  static_assert(!T32 || !F32, "S_TFIRST32 is its own storage form");
  constexpr bool POWREG = MODE == S_TFIRST || T32;                // the power on the registers, before the tile is parked
  static_assert(!F32 || (MODE != S_TFIRST && MODE != S_MID), "plain T has no fp32 form");
  extern __shared__ double lds[];
  if (io.gate != nullptr) {
    const unsigned long long g = *io.gate;
    if (g <= (unsigned long long)__double_as_longlong(io.gate_tol)) return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform: addresses below stay scalar
  if (io.zero != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *io.zero = 0ULL;   // ordered before the last pass by the launches in between
  const long long tile = (long long)blockIdx.x * Geo::WAVES + wave;
  const long long s0 = tile * Geo::G;
  if (s0 >= P.nslices) return;                                   // no workgroup barrier below
  const long long rem = (P.nslices - s0) * (N * N / 2);          // valid double2 units of a trailing partial tile
  const int nvalid = rem < Geo::UNITS ? (int)rem : Geo::UNITS;
  double* const wl = lds + wave * Geo::LTILE;
  // element e of the tile (linear image of the global layout) -> LDS offset: row e / N at stride RS
  auto lofs = [](const int e) -> int { return Geo::RS == N ? e : (e / N) * Geo::RS + (e % N); };
  const long long gbase = s0 * (N * N);
  const unsigned lb = (unsigned)lane * 16u;

  if (JV32) {
    // ---- fp32 streams: unit u = lane + 64 k holds elements 4u .. 4u+3 --------------------------------------
    const int nvalid4 = nvalid / 2;
    const char* const inb = reinterpret_cast<const char*>(reinterpret_cast<const float*>(io.in) + gbase);
    const char* const auxb = reinterpret_cast<const char*>(reinterpret_cast<const float*>(io.aux_in) + gbase);
    float4 v[Geo::EPT4], c1v[Geo::EPT4];
#pragma unroll
    for (int k = 0; k < Geo::EPT4; ++k) {
      const int u = lane + 64 * k;
      if (u < nvalid4) {
        v[k] = *reinterpret_cast<const float4*>(inb + (lb + 1024u * k));
        c1v[k] = *reinterpret_cast<const float4*>(auxb + (lb + 1024u * k));
      } else {
        v[k] = make_float4(0.f, 0.f, 0.f, 0.f); c1v[k] = v[k];
      }
    }
#pragma unroll
    for (int k = 0; k < Geo::EPT4; ++k) {
      const int u = lane + 64 * k;
      if (Geo::UNITS4 % 64 == 0 || u < Geo::UNITS4) {
        *reinterpret_cast<double2*>(wl + lofs(4 * u)) = make_double2((double)v[k].x * (double)c1v[k].x, (double)v[k].y * (double)c1v[k].y);
        *reinterpret_cast<double2*>(wl + lofs(4 * u + 2)) = make_double2((double)v[k].z * (double)c1v[k].z, (double)v[k].w * (double)c1v[k].w);
      }
    }
  } else {
    // uniform 64-bit bases + one 32-bit byte offset per lane: global_load/store with an SGPR base
    const char* const inb = reinterpret_cast<const char*>(io.in + gbase);
    const char* const auxb = reinterpret_cast<const char*>(io.aux_in + gbase);
    // ---- loads (all in flight at once) -----------------------------------------------------------------------
    double2 v[Geo::EPT];
    double2 c1v[MULP ? Geo::EPT : 1];
#pragma unroll
    for (int k = 0; k < Geo::EPT; ++k) {
      const int u = lane + 64 * k;
      if (u < nvalid) {
        v[k] = ldg_stream2(inb + (lb + 1024u * k));
        if (MULP) c1v[MULP ? k : 0] = ldg_stream2(auxb + (lb + 1024u * k));
      } else {
        v[k] = make_double2(1.0, 1.0);
        if (MULP) c1v[MULP ? k : 0] = make_double2(0.0, 0.0);
      }
    }
    // ---- park the tile in LDS (linear image of the global layout); J.v: x = c1 * v on the way ------------
    // plain T: w^theta on the registers as they are parked -- one LDS write per unit instead of write + read + write
    // (unrolled: EPT copies of the power routine; the LDS pipe is this pass's second bottleneck: 0.262 -> 0.254 ms at
    // GCY 20^6, tools/probes/kernel_bench.hip).  The linearising and fp32-intermediate forms keep the rolled loop below.
    PowK<true> PTr;
    if (POWREG) PTr.init(P.theta, lane);
    double t32s = 1.0;                                             // fp32-intermediate form: x * 2^k (t32_scale_of)
    if (T32) { const PowLane PL0 = pow_lane_init(lane); t32s = t32_scale_of(P.t32_ref > 0.0 ? P.t32_ref : io.in[P.ref_off], P.theta, PL0, false); }
#pragma unroll
    for (int k = 0; k < Geo::EPT; ++k) {
      const int u = lane + 64 * k;
      if (MULP) { v[k].x *= c1v[MULP ? k : 0].x; v[k].y *= c1v[MULP ? k : 0].y; }
      if (POWREG) {
        const double xin[2] = {v[k].x, v[k].y};       // (masked lanes were loaded as 1)
        double xw[2];
        PTr.template run<2>(xin, xw);
        v[k] = T32 ? make_double2(xw[0] * t32s, xw[1] * t32s) : make_double2(xw[0], xw[1]);
      }
      if (Geo::UNITS % 64 == 0 || u < Geo::UNITS) *reinterpret_cast<double2*>(wl + lofs(2 * u)) = v[k];
    }
  }
  QFrag<N> qf;
  qf.load(P.Qf, lane);
  wave_lds_fence();
  // ---- prologue x = w^theta in place (rolled: one copy of the power routine; every lane stays active) --
  if (POWP && !POWREG) {
    const PowLane PT = pow_lane_init(lane);
    double lin_scale = 1.0;
    if (LIN && F32) lin_scale = lin_scale_of(io.in[P.ref_off], P.theta, PT, false);
    const double t32_scale = T32 ? t32_scale_of(P.t32_ref > 0.0 ? P.t32_ref : io.in[P.ref_off], P.theta, PT, false) : 1.0;
    char* const auxo = LIN ? (F32 ? reinterpret_cast<char*>(reinterpret_cast<float*>(io.aux_out) + gbase)
                                  : reinterpret_cast<char*>(io.aux_out + gbase)) : nullptr;
#pragma unroll 1
    for (int k = 0; k < Geo::EPT; ++k) {
      const int u = lane + 64 * k;
      const bool in_tile = Geo::UNITS % 64 == 0 || u < Geo::UNITS;
      const int lo = in_tile ? lofs(2 * u) : 0;
      const double2 x2 = *reinterpret_cast<const double2*>(wl + lo);
      const double xin[2] = {in_tile ? x2.x : 1.0, in_tile ? x2.y : 1.0};
      double xw[2];
      pow_fast_n<true, 2>(xin, P.theta, PT, xw);
      if (in_tile) {
        *reinterpret_cast<double2*>(wl + lo) = T32 ? make_double2(xw[0] * t32_scale, xw[1] * t32_scale) : make_double2(xw[0], xw[1]);
        if (LIN && u < nvalid) {                                  // c1 = w^(theta-1)
          if (F32) *reinterpret_cast<float2*>(auxo + ((unsigned)lane * 8u + 512u * k)) =
              make_float2((float)(xw[0] / xin[0] * lin_scale), (float)(xw[1] / xin[1] * lin_scale));
          else *reinterpret_cast<double2*>(auxo + (lb + 1024u * k)) = make_double2(xw[0] / xin[0], xw[1] / xin[1]);
        }
      }
    }
    wave_lds_fence();
  }

  const int li = lane & 15, lk = lane >> 4;
  // ---- contraction over the fastest axis: column c = (slice, e) at wl + c * RS, rows contiguous -------
  {
    double* const p0 = wl + li * Geo::RS + lk;
#pragma unroll
    for (int ct = 0; ct < Geo::NCT; ++ct) {
      const bool rep = Geo::CPAD > 0 && ct == Geo::NCT - 1;       // (compile time)
      ctile<N, 1>(p0 + (ct * 16 - ((rep && li >= 16 - Geo::CPAD) ? Geo::CPAD : 0)) * Geo::RS, qf);
    }
  }
  wave_lds_fence();
  // ---- contraction over the second axis: column c = (slice g, f) at wl + g N RS + f, row stride RS ---
  {
    QFrag<N> qe;
    qe.load(P.Qe, lane);
#pragma unroll
    for (int ct = 0; ct < Geo::NCT; ++ct) {
      const bool rep = Geo::CPAD > 0 && ct == Geo::NCT - 1;
      const int c = 16 * ct + li - ((rep && li >= 16 - Geo::CPAD) ? Geo::CPAD : 0);
      const int g = c / N, f = c - g * N;
      ctile<N, Geo::RS>(wl + g * (N * Geo::RS) + f + lk * Geo::RS, qe);
    }
  }
  wave_lds_fence();
  // ---- stream out -------------------------------------------------------------------------------------
  if (JV32 || T32) {
    const int nvalid4 = nvalid / 2;
    char* const outb = reinterpret_cast<char*>(reinterpret_cast<float*>(io.out) + gbase);
#pragma unroll
    for (int k = 0; k < Geo::EPT4; ++k) {
      const int u = lane + 64 * k;
      if (u < nvalid4) {
        const double2 a = *reinterpret_cast<const double2*>(wl + lofs(4 * u)), b = *reinterpret_cast<const double2*>(wl + lofs(4 * u + 2));
        *reinterpret_cast<float4*>(outb + (lb + 1024u * k)) = make_float4((float)a.x, (float)a.y, (float)b.x, (float)b.y);
      }
    }
  } else {
    char* const outb = reinterpret_cast<char*>(io.out + gbase);
#pragma unroll
    for (int k = 0; k < Geo::EPT; ++k) {
      const int u = lane + 64 * k;
      if (u < nvalid) stg_stream2<NT_SLICE>(outb + (lb + 1024u * k), *reinterpret_cast<const double2*>(wl + lofs(2 * u)));
    }
  }
}

// ---------------------------------------------------------------------------------------------------
constexpr int LINE_R = 16;        // doubles per row of a line tile: one 128-byte line

struct LineDesc {
  long long lrest;          // contiguous doubles behind axis Y (= element stride of Y); X's stride is n * lrest
  int nchunks;              // ceil(lrest / 16)
  long long nouter;         // product of the extents before X; outer stride = n * n * lrest
  long long ntiles;         // nouter * nchunks
  const double* Qx;
  const double* Qy;
  double inv_theta, beta;
  double theta;             // L_TFUSED: exponent of the next application's first pass
  double cbt;               // beta^theta (next_power_fast)
  int first_only;           // L_TFUSED: skip the closing half (the tile holds w, not an intermediate)
  // aggregator scale a3 (current state), index = out_idx[o] + x * a3x + y * a3y + rest_idx[pos]
  const double* a3;
  const int* out_idx;
  const int* rest_idx;
  int a3x, a3y;
  int minus_identity;
  long long ref_off;        // C-order offset of the mid-grid point (fp32 linearisation scale)
  double t32_ref;           // > 0: reference value of the t_f32 scale (sharded stages: the mid-grid point may live on another rank)
  int cached_out;           // the grid is small enough for the next application's first pass to find T w in the Infinity Cache:
                            // the last pass then stores it with the default policy, not non-temporally
  // a3 as two small tables where it factorises (build_fast_plan): a3 = f1[o * n + x] * f2[o * lrest + position]; else null.
  // Read by the streamed last pass (stream_kernels.hpp, A3F).  The same form as a run-time branch in line_kernel's own
  // epilogue lost at GCY 16^6 (last pass 0.0905 against 0.0863 ms): that kernel stays with the gathers.
  const double* f1;
  const double* f2;
};

struct LineIO {
  const double* in;
  double* out;
  const double* aux_in;     // J.v: c2
  double* aux_out;          // T + linearise: c2 = beta u / S;  L_TFUSED: the next intermediate
  const double* old;        // T: w (residual);  J.v: v
  unsigned long long* resid;
  const unsigned long long* gate;
  double gate_tol;
  double* dotp;             // J.v with minus_identity: per-workgroup partial sums <out, v>, <out, out>: [2][gridDim.x]
  unsigned* sched;          // persistent form: {next tile ticket, finished workgroups}, both zero between launches
  const double* dot_with;   // streamed last pass of J.v (stream_kernels.hpp): also <out, dot_with> into dotp[2][gridDim.x]
                            // (BiCGSTAB's <rhat, q>, krylov_kernels.hpp)
  // streamed last pass of T, Anderson on large grids (round 4): the push of the pass rides on it -- r = T x - x into
  // and_r, y = x + beta r into and_y, the workgroup's part of <r, r> into and_dot[blockIdx.x] (k_and_push_lite's outputs)
  double* and_y;
  double* and_r;
  double and_beta;
  double* and_dot;
};

template <int N> struct LineGeo {
  static constexpr int B = N <= 24 ? 256 : 512;                   // threads per workgroup (N = 24: 18 units per thread, two workgroups per CU)
  static constexpr int W = B / 64;
  static constexpr int UNITS = N * N * LINE_R / 2;                // double2 units per tile
  static constexpr int EPT = (UNITS + B - 1) / B;
  static constexpr int LX = N * LINE_R;                           // LDS stride of X (doubles); Y's is LINE_R
  static constexpr int BPC = N <= 16 ? 4 : (N == 20 ? 3 : (N == 24 ? 2 : 1));   // workgroups per CU (LDS)
  static constexpr int UNITS4 = N * N * LINE_R / 4;               // float4 units per tile (fp32 streams)
  static constexpr int EPT4 = (UNITS4 + B - 1) / B;
  static_assert(N % W == 0, "column tiles split evenly over the waves");
};

// Per-tile addressing of a line tile: tile t -> (outer o, chunk); unit u = tid + k B sits in row u >> 3 at
// double2 c2 = tid & 7; global byte offset = b0 + k * bstep against the tile's uniform base (X, Y adjacent
// axes: row * lrest + position).
struct LineTile {
  long long tbase;          // element offset of the tile's first row, first double
  unsigned o;
  long long pos;            // this thread's position in the contiguous remainder behind Y
  bool cok;                 // inside the remainder (a trailing chunk may be partial)
};
__device__ __forceinline__ LineTile line_tile(const LineDesc& P, unsigned t, int c2, int nn) {
  LineTile T;
  T.o = t / (unsigned)P.nchunks;
  const int chunk = (int)(t - T.o * (unsigned)P.nchunks);
  T.pos = (long long)chunk * LINE_R + 2 * c2;
  T.cok = T.pos < P.lrest;
  T.tbase = (long long)T.o * nn * P.lrest + (long long)chunk * LINE_R;
  return T;
}

// Persistent: workgroup b starts on tile b and then draws tiles gridDim.x, gridDim.x + 1, ... from a ticket
// counter (the host launches at most BPC workgroups per CU; 10000 tiles over 768 workgroups would otherwise
// leave 752 of them idle during a fourteenth round).  The ticket for the tile after next is drawn by one lane
// before the contractions and handed round through LDS behind their barriers; the last workgroup to leave
// resets the two counters, so no memset sits between launches.  The epilogue of tile t -- in the last pass the long VALU phase, two powers per unit -- runs unit by
// unit with a LOOK-unit window of global loads ahead of it: the unit's side streams (a3 and the residual's w,
// or the J.v's c2 and v) and the same unit of tile t+1, which is parked in the LDS slot the unit has just
// been read from.  So a workgroup has global requests in flight through every phase, the next tile is in
// LDS when the epilogue ends, and nothing of it waits in registers (a whole prefetched tile, 52 VGPRs, next
// to the power routine does not fit the 168 VGPRs of three workgroups per CU).  A thread's epilogue touches
// only LDS slots it owns, so no barrier separates it from the parking.
// PERSIST = false: one workgroup per tile (grid = ntiles, XCD-contiguous tile order), no look-ahead into a
// next tile: the better form for the middle pass, whose epilogue is a plain store (measured: persistent
// 0.24 ms vs 0.20 ms at GCY 20^6).
// FULLC: the remainder behind Y is a multiple of 16 doubles, i.e. no tile has a partial chunk: loads need no
// lane mask (a masked row of the last, partial unit reads unit 0's address instead) and the selects go away.
// F32 (opts.krylov_f32, needs FULLC): middle / last pass of J.v -- every stream holds floats, 16-byte units of
// four, one tile per workgroup; last pass of a linearising T -- its own streams stay fp64, c2 is written as
// scaled floats.  Arithmetic, LDS tile and reductions stay fp64.
template <int N, int MODE, bool PERSIST, bool FULLC, bool F32>
__global__ void __launch_bounds__(LineGeo<N>::B, LineGeo<N>::BPC * LineGeo<N>::B / 256)
line_kernel(const LineDesc P, const LineIO io) {
  using Geo = LineGeo<N>;
  constexpr int B = Geo::B;
  constexpr int EPT = Geo::EPT;
  static_assert(!F32 || (FULLC && !PERSIST && MODE != L_TLAST), "fp32 forms: whole chunks, one tile per workgroup");
  if constexpr (F32 && (MODE == L_MID || MODE == L_JLAST)) {
    constexpr bool MUL = MODE == L_JLAST;
    constexpr int EPT4 = Geo::EPT4;
    constexpr bool PART4 = Geo::UNITS4 % B != 0;
    extern __shared__ double lds[];
    __shared__ double red4[16];
    if (io.gate != nullptr) {
      const unsigned long long g = *io.gate;
      if (g <= (unsigned long long)__double_as_longlong(io.gate_tol)) return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const unsigned t = (unsigned)xcd_remap((long long)blockIdx.x, P.ntiles);
    const unsigned o = t / (unsigned)P.nchunks;
    const int chunk = (int)(t - o * (unsigned)P.nchunks);
    const long long tbase = (long long)o * (N * N) * P.lrest + (long long)chunk * LINE_R;     // elements
    // unit u = tid + k B: row u >> 2, float4 tid & 3 of its 16 elements; LDS doubles 4u .. 4u+3
    const unsigned b0 = ((unsigned)(tid >> 2) * (unsigned)P.lrest + 4u * (tid & 3)) * 4u;
    const unsigned bstep = (unsigned)(B / 4) * (unsigned)P.lrest * 4u;
    const char* const inb = reinterpret_cast<const char*>(reinterpret_cast<const float*>(io.in) + tbase);
    const char* const auxb = reinterpret_cast<const char*>(reinterpret_cast<const float*>(io.aux_in) + tbase);
    const char* const oldb = reinterpret_cast<const char*>(reinterpret_cast<const float*>(io.old) + tbase);
    char* const outb = reinterpret_cast<char*>(reinterpret_cast<float*>(io.out) + tbase);
    const bool need_old = MUL && P.minus_identity;
    float4 v[EPT4];
#pragma unroll
    for (int k = 0; k < EPT4; ++k) {
      const bool rowok = !PART4 || tid + k * B < Geo::UNITS4;
      v[k] = *reinterpret_cast<const float4*>(inb + (rowok ? b0 + k * bstep : b0));
    }
    QFrag<N> q;
    q.load(P.Qx, lane);
#pragma unroll
    for (int k = 0; k < EPT4; ++k) {
      const int u = tid + k * B;
      if (!PART4 || u < Geo::UNITS4) {
        *reinterpret_cast<double2*>(lds + 4 * u) = make_double2((double)v[k].x, (double)v[k].y);
        *reinterpret_cast<double2*>(lds + 4 * u + 2) = make_double2((double)v[k].z, (double)v[k].w);
      }
    }
    // side streams of the J.v epilogue: v travels across the contractions, c2 is fetched behind them (both
    // at once, next to the Q fragments, do not fit the register budget of three workgroups per CU)
    float4 c2v[MUL ? EPT4 : 1], oldv[MUL ? EPT4 : 1];
    if (MUL && need_old) {
#pragma unroll
      for (int k = 0; k < EPT4; ++k) {
        const bool rowok = !PART4 || tid + k * B < Geo::UNITS4;
        oldv[MUL ? k : 0] = *reinterpret_cast<const float4*>(oldb + (rowok ? b0 + k * bstep : b0));
      }
    }
    __syncthreads();
    {
      double* const p0 = lds + li + lk * Geo::LX;
#pragma unroll
      for (int j = 0; j < N / Geo::W; ++j) ctile<N, Geo::LX>(p0 + (wave + j * Geo::W) * 16, q);
    }
    q.load(P.Qy, lane);
    __syncthreads();
    {
      double* const p0 = lds + li + lk * LINE_R;
#pragma unroll
      for (int j = 0; j < N / Geo::W; ++j) ctile<N, LINE_R>(p0 + (wave + j * Geo::W) * Geo::LX, q);
    }
    __syncthreads();
    if (MUL) {
#pragma unroll
      for (int k = 0; k < EPT4; ++k) {
        const bool rowok = !PART4 || tid + k * B < Geo::UNITS4;
        c2v[MUL ? k : 0] = *reinterpret_cast<const float4*>(auxb + (rowok ? b0 + k * bstep : b0));
      }
    }
    double dot_yv = 0.0, dot_yy = 0.0;
#pragma unroll
    for (int k = 0; k < EPT4; ++k) {
      const int u = tid + k * B;
      if (!PART4 || u < Geo::UNITS4) {
        const double2 a = *reinterpret_cast<const double2*>(lds + 4 * u), b = *reinterpret_cast<const double2*>(lds + 4 * u + 2);
        double y[4] = {a.x, a.y, b.x, b.y};
        float yr[4];
        if (MUL) {
          const float4 c = c2v[MUL ? k : 0];
          const float cc[4] = {c.x, c.y, c.z, c.w};
          const float4 ov4 = need_old ? oldv[MUL ? k : 0] : make_float4(0.f, 0.f, 0.f, 0.f);
          const float ov[4] = {ov4.x, ov4.y, ov4.z, ov4.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            y[j] *= (double)cc[j];
            if (need_old) {
              y[j] -= (double)ov[j];
              yr[j] = (float)y[j];                                // what gets stored is what the dots refer to
              dot_yv = fma((double)yr[j], (double)ov[j], dot_yv);
              dot_yy = fma((double)yr[j], (double)yr[j], dot_yy);
            } else yr[j] = (float)y[j];
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) yr[j] = (float)y[j];
        }
        *reinterpret_cast<float4*>(outb + (b0 + k * bstep)) = make_float4(yr[0], yr[1], yr[2], yr[3]);
      }
    }
    if (MUL && io.dotp != nullptr) {
#pragma unroll
      for (int s = 32; s > 0; s >>= 1) { dot_yv += __shfl_xor(dot_yv, s); dot_yy += __shfl_xor(dot_yy, s); }
      if (lane == 0) { red4[wave] = dot_yv; red4[8 + wave] = dot_yy; }
      __syncthreads();
      if (tid == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < Geo::W; ++w) { a += red4[w]; b += red4[8 + w]; }
        io.dotp[blockIdx.x] = a;
        io.dotp[gridDim.x + blockIdx.x] = b;
      }
    }
    return;
  }
  constexpr bool FUSED = MODE == L_TFUSED;
  constexpr bool CES = MODE == L_TLAST || MODE == L_TLAST_LIN || FUSED;
  constexpr bool LINE = MODE == L_TLAST_LIN;
  constexpr bool MULE = MODE == L_JLAST;
  constexpr bool PARTIAL = Geo::UNITS % B != 0;
  static_assert(!FUSED || !PERSIST, "the fused form runs one tile per workgroup");
  constexpr int LOOK = 4;                                         // units of look-ahead in the epilogue
  extern __shared__ double lds[];
  __shared__ double red[16];
  if (io.gate != nullptr) {
    const unsigned long long g = *io.gate;
    if (g <= (unsigned long long)__double_as_longlong(io.gate_tol)) return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int c2 = tid & 7;                                        // this thread's double2 inside the 16-double row
  const unsigned b0 = ((unsigned)(tid >> 3) * (unsigned)P.lrest + 2u * c2) * 8u;    // a tile spans < 4 GB (host check)
  const unsigned bstep = (unsigned)(B / 8) * (unsigned)P.lrest * 8u;
  const bool first_only = FUSED && P.first_only != 0;            // uniform
  const bool need_old = CES ? (io.resid != nullptr && !first_only) : (MULE && P.minus_identity);
  const unsigned ntiles = PERSIST ? (unsigned)P.ntiles : 0u, gstep = gridDim.x;
  const char* const a3b = reinterpret_cast<const char*>(P.a3);
  const unsigned a3x = (unsigned)P.a3x, a3y = (unsigned)P.a3y;   // < 2^24 (host check)

  __shared__ unsigned next_tile[2];     // alternating slots: a slot is redrawn two tiles (six barriers) after its last read
  int parity = 0;
  unsigned t = PERSIST ? blockIdx.x : (unsigned)xcd_remap((long long)blockIdx.x, P.ntiles);   // gridDim.x <= ntiles
  LineTile T = line_tile(P, t, c2, N * N);
  {
    // first tile: all loads in flight at once, then parked
    double2 v[EPT];
    const char* const inb = reinterpret_cast<const char*>(io.in + T.tbase);
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const bool rowok = !PARTIAL || tid + k * B < Geo::UNITS;
      if (FULLC) v[k] = *reinterpret_cast<const double2*>(inb + (rowok ? b0 + k * bstep : b0));
      else v[k] = (T.cok && rowok) ? *reinterpret_cast<const double2*>(inb + (b0 + k * bstep)) : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int k = 0; k < EPT; ++k)
      if (!PARTIAL || tid + k * B < Geo::UNITS) *reinterpret_cast<double2*>(lds + 2 * (tid + k * B)) = v[k];
  }
  double rmax = 0.0, dot_yv = 0.0, dot_yy = 0.0;
  bool rnan = false;
  PowLane PT;
  if (CES) PT = pow_lane_init(lane);
  PowK<false> PK;                                                // the aggregator's power (exponent fixed for the launch)
  if (CES) PK.init(P.inv_theta, lane);
  double lin_scale = 1.0;                                        // fp32 c2 = beta u / S * 2^-k
  if (LINE && F32) lin_scale = lin_scale_of(io.old[P.ref_off], 1.0 / P.inv_theta, PT, true);

  for (;;) {
    if (PERSIST && tid == 0)                                      // ticket of the tile after this one
      next_tile[parity] = gstep + __hip_atomic_fetch_add(&io.sched[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    auto contract_tile = [&]() {
      QFrag<N> q;
      q.load(P.Qx, lane);
      __syncthreads();
      // ---- contraction over X: column = (y, r) = LDS offset, row stride LX -----------------------------
      {
        double* const p0 = lds + li + lk * Geo::LX;
#pragma unroll
        for (int j = 0; j < N / Geo::W; ++j) ctile<N, Geo::LX>(p0 + (wave + j * Geo::W) * 16, q);
      }
      q.load(P.Qy, lane);
      __syncthreads();
      // ---- contraction over Y: column = (x, r) at x * LX + r, row stride 16 ----------------------------
      {
        double* const p0 = lds + li + lk * LINE_R;
#pragma unroll
        for (int j = 0; j < N / Geo::W; ++j) ctile<N, LINE_R>(p0 + (wave + j * Geo::W) * Geo::LX, q);
      }
      __syncthreads();
    };
    if (!first_only) contract_tile();                              // (first_only: a thread's units are its own slots)

    const unsigned tn = PERSIST ? next_tile[parity] : 0u;        // written before the first of the three barriers above
    const bool has_next = PERSIST && tn < ntiles;                // uniform over the workgroup
    const LineTile Tn = line_tile(P, has_next ? tn : t, c2, N * N);
    const bool cok = T.cok;
    const char* const nxb = reinterpret_cast<const char*>(io.in + Tn.tbase);
    const char* const oldb = reinterpret_cast<const char*>(io.old + T.tbase);
    const char* const auxb = reinterpret_cast<const char*>(io.aux_in + T.tbase);
    char* const outb = reinterpret_cast<char*>(io.out + T.tbase);
    char* const auxo = reinterpret_cast<char*>(io.aux_out + T.tbase);
    char* const auxo32 = reinterpret_cast<char*>(reinterpret_cast<float*>(io.aux_out) + T.tbase);
    unsigned ia3a = 0u, ia3b = 0u;                               // a3 index of this thread's two elements, less (x, y)
    if (CES && cok && !first_only) {
      ia3a = (unsigned)(P.out_idx[T.o] + P.rest_idx[T.pos]);
      ia3b = (unsigned)(P.out_idx[T.o] + P.rest_idx[T.pos + 1]);
    }

    // ---- epilogue: unit k of tile t, with the loads of unit k + LOOK (and of tile t+1) ahead of it --------
    double2 vw[LOOK], sw[LOOK], cw[LOOK];     // next tile's unit; side stream 1 (w / v); side stream 2 (a3 pair / c2)
    auto issue = [&](const int k, double2& vn, double2& s1, double2& s2) {
      if (first_only) return;
      const int u = tid + k * B;
      const bool rowok = k < EPT && (!PARTIAL || u < Geo::UNITS);
      const unsigned off = b0 + (unsigned)k * bstep;
      if (FULLC) {
        // every address of the tile is readable: masked rows read unit 0 (no exec branch, no zero fill)
        const unsigned offc = rowok ? off : b0;
        if (has_next) vn = *reinterpret_cast<const double2*>(nxb + offc);
        if ((CES || MULE) && need_old) s1 = *reinterpret_cast<const double2*>(oldb + offc);
        if (CES) {
          const int row = rowok ? (u >> 3) : 0;
          const int x = row / N, y = row - x * N;
          const unsigned ixy = __umul24((unsigned)x, a3x) + __umul24((unsigned)y, a3y);
          s2 = make_double2(*reinterpret_cast<const double*>(a3b + (ia3a + ixy) * 8u),
                            *reinterpret_cast<const double*>(a3b + (ia3b + ixy) * 8u));
        } else if (MULE) {
          s2 = *reinterpret_cast<const double2*>(auxb + offc);
        }
        return;
      }
      vn = (has_next && rowok && Tn.cok) ? *reinterpret_cast<const double2*>(nxb + off) : make_double2(0.0, 0.0);
      const bool ok = rowok && cok;
      s1 = ((CES || MULE) && ok && need_old) ? *reinterpret_cast<const double2*>(oldb + off) : make_double2(0.0, 0.0);
      if (CES) {
        // unconditional gathers (index 0 when masked)
        const int row = rowok ? (u >> 3) : 0;
        const int x = row / N, y = row - x * N;
        const unsigned ixy = __umul24((unsigned)x, a3x) + __umul24((unsigned)y, a3y);
        s2 = make_double2(*reinterpret_cast<const double*>(a3b + (ok ? (ia3a + ixy) * 8u : 0u)),
                          *reinterpret_cast<const double*>(a3b + (ok ? (ia3b + ixy) * 8u : 0u)));
      } else if (MULE) {
        s2 = ok ? *reinterpret_cast<const double2*>(auxb + off) : make_double2(0.0, 0.0);
      }
    };
    auto unit = [&](const int k, const double2 vn, const double2 s1, const double2 s2) {
      const int u = tid + k * B;
      const bool rowok = !PARTIAL || u < Geo::UNITS;
      const bool ok = (FULLC || cok) && rowok;
      const unsigned off = b0 + (unsigned)k * bstep;
      double2* const slot = reinterpret_cast<double2*>(lds + 2 * (rowok ? u : tid));
      const double2 sv = *slot;
      if (first_only) {
        // the tile holds w: x = w^theta in place (masked lanes feed the power 1)
        const double xin[2] = {ok ? sv.x : 1.0, ok ? sv.y : 1.0};
        double xw[2];
        pow_fast_n<true, 2>(xin, P.theta, PT, xw);
        if (rowok) *slot = make_double2(xw[0], xw[1]);
        return;
      }
      if (CES) {
        // Tw = 1 + beta (a3 S)^(1/theta), c2 = beta u / S, |Tw - w|.  Uniform trip: every lane runs the power
        // (its table gathers need the whole wave); masked lanes feed it 1 (under FULLC: real values of the
        // tile, whatever they are, the result is dropped).
        const double ks[2] = {(FULLC || ok) ? s2.x * sv.x : 1.0, (FULLC || ok) ? s2.y * sv.y : 1.0};
        double uu[2];
        PK.template run<2>(ks, uu);
        const double2 y2 = make_double2(1.0 + P.beta * uu[0], 1.0 + P.beta * uu[1]);
        if (ok) {
          if (LINE) {
            if (F32) *reinterpret_cast<float2*>(auxo32 + (off >> 1)) =
                make_float2((float)(P.beta * uu[0] / sv.x * lin_scale), (float)(P.beta * uu[1] / sv.y * lin_scale));
            else *reinterpret_cast<double2*>(auxo + off) = make_double2(P.beta * uu[0] / sv.x, P.beta * uu[1] / sv.y);
          }
          if (need_old) {
            // fmax drops NaNs: they are collected in a flag and turned into +inf after the last tile
            const double r0 = fabs(y2.x - s1.x), r1 = fabs(y2.y - s1.y);
            rnan |= (r0 != r0) | (r1 != r1);
            rmax = fmax(rmax, fmax(r0, r1));
          }
          *reinterpret_cast<double2*>(outb + off) = y2;
        }
        if (FUSED) {
          // the next application's first pass on this pair: x = (Tw)^theta into the slot just read
          const double bu[2] = {ok ? P.beta * uu[0] : 1e4, ok ? P.beta * uu[1] : 1e4};
          double xw[2];
          if (!next_power_fast<2>(bu, ks, P.theta, P.cbt, xw)) {
            const double xin[2] = {ok ? y2.x : 1.0, ok ? y2.y : 1.0};
            pow_fast_n<true, 2>(xin, P.theta, PT, xw);
          }
          if (rowok) *slot = make_double2(xw[0], xw[1]);
        }
      } else if (ok) {
        double2 y2 = sv;
        if (MULE) {
          y2.x *= s2.x; y2.y *= s2.y;
          if (P.minus_identity) {
            y2.x -= s1.x; y2.y -= s1.y;
            dot_yv = fma(y2.x, s1.x, dot_yv); dot_yv = fma(y2.y, s1.y, dot_yv);
            dot_yy = fma(y2.x, y2.x, dot_yy); dot_yy = fma(y2.y, y2.y, dot_yy);
          }
        }
        *reinterpret_cast<double2*>(outb + off) = y2;
      }
      if (has_next && rowok) *slot = vn;                         // park tile t+1's unit in the slot just read
    };
#pragma unroll
    for (int j = 0; j < LOOK; ++j) issue(j, vw[j], sw[j], cw[j]);
    int kk = 0;
#pragma unroll 1
    for (; kk + LOOK <= EPT; kk += LOOK) {
#pragma unroll
      for (int j = 0; j < LOOK; ++j) {
        unit(kk + j, vw[j], sw[j], cw[j]);
        issue(kk + j + LOOK, vw[j], sw[j], cw[j]);
        __builtin_amdgcn_sched_barrier(0);                       // keep the window at LOOK units (register budget)
      }
    }
#pragma unroll
    for (int j = 0; j < EPT % LOOK; ++j) unit(EPT - EPT % LOOK + j, vw[j], sw[j], cw[j]);

    if (FUSED) {
      contract_tile();                                             // (its first barrier closes the epilogue's LDS writes)
      char* const tb = reinterpret_cast<char*>(io.aux_out + T.tbase);
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const int u = tid + k * B;
        if ((FULLC || cok) && (!PARTIAL || u < Geo::UNITS))
          *reinterpret_cast<double2*>(tb + (b0 + (unsigned)k * bstep)) = *reinterpret_cast<const double2*>(lds + 2 * u);
      }
    }
    if (!has_next) break;
    t = tn;
    T = Tn;
    parity ^= 1;
  }
  if (PERSIST && tid == 0) {
    // last workgroup out clears the scheduler words for the next launch (kernel boundaries order the launches)
    const unsigned d = __hip_atomic_fetch_add(&io.sched[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (d == gridDim.x - 1) {
      __hip_atomic_store(&io.sched[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&io.sched[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }

  // per-workgroup reductions over all its tiles
  if (MULE && io.dotp != nullptr) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) { dot_yv += __shfl_xor(dot_yv, s); dot_yy += __shfl_xor(dot_yy, s); }
    if (lane == 0) { red[wave] = dot_yv; red[8 + wave] = dot_yy; }
    __syncthreads();
    if (tid == 0) {
      double a = 0.0, b = 0.0;
      for (int w = 0; w < Geo::W; ++w) { a += red[w]; b += red[8 + w]; }
      io.dotp[blockIdx.x] = a;
      io.dotp[gridDim.x + blockIdx.x] = b;
    }
  }
  if (CES && io.resid != nullptr) {
    if (rnan) rmax = __longlong_as_double(0x7ff0000000000000LL);                // NaN -> +inf
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) rmax = fmax(rmax, __shfl_xor(rmax, s));
    if (lane == 0) red[wave] = rmax;
    __syncthreads();
    if (tid == 0) {
      double r = red[0];
      for (int w = 1; w < Geo::W; ++w) r = fmax(r, red[w]);
      atomicMax(io.resid, (unsigned long long)__double_as_longlong(r));
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Small grids (every extent <= 16; SSY 15^4 of BASELINE configs 1 and 2 is 405 KB): the same pair plan with
// run-time extents.  Such a grid lives in L2 and a pass is bound by its own critical path -- launch (1.55 us
// per graph node, tools/probes/launch_floor_probe.hip), one global round trip, the power's ~280 dependent
// instructions per two points, one more round trip -- so the plan wants few launches with little work per
// wave: D/2 passes, one WAVE per tile, no workgroup barrier on the data path.
// A tile is all nx x ny (x, y) rows of R consecutive positions of the contiguous remainder behind Y (R = 1:
// every pass is then a set of strided 2-D slices, <= 4 points per lane; R = 4: 32-byte runs for the larger of
// these grids).  The tile sits in LDS padded to 16 x 16 x R with zeros, the matrices are padded to 16 x 16
// with zero rows and columns: padded outputs are exact zeros, so ctile<16, .> runs unmasked.
// SM_FUSED_T: the last pass of one application of T and the first pass of the next in one kernel (successive
// approximation with the pair order reversed every iteration, so that both act on the same pair): contract,
// aggregator, residual, store Tw, then (Tw)^theta and the same two contractions again, into the intermediate.
#ifndef SDFS_SMALL_STAMP
#define SDFS_SMALL_STAMP(i)       // tools/probes/small_fused_probe.hip defines it: phase time stamps of workgroup 0
#endif
#ifndef SDFS_AND_STAMP
#define SDFS_AND_STAMP(i)         // tools/probes/anderson_fused_probe.hip: phase time stamps of the control workgroup
#endif
enum SmallMode { SM_FIRST_T = 0, SM_FIRST_TLIN = 1, SM_FIRST_J = 2, SM_MID = 3, SM_LAST_T = 4, SM_LAST_TLIN = 5, SM_LAST_J = 6, SM_FUSED_T = 7,
                 SM_AND_FIRST = 8, SM_AND_LAST = 9 };   // Anderson: first pass with the previous pass's control step and the update of x; last pass with the push

struct SmallDesc {
  int nx, ny;               // extents of the contracted pair (X slower)
  unsigned my;              // ceil(65536 / ny): row / ny == (row * my) >> 16 for row < 256
  unsigned sx, sy;          // element strides of X and Y
  long long ostride;        // element stride of the outer index (all axes before X, or the slice index)
  unsigned lrest;           // positions behind Y
  unsigned nchunks;         // ceil(lrest / R)
  long long ntiles;         // nouter * nchunks
  unsigned cpx;             // XCD-aware launch of a strided pass (0: workgroup b works tile group b): workgroups per XCD;
                            // workgroup b -- dispatched to XCD b mod 8 -- takes group (b mod 8) * cpx + b / 8, so that the
                            // tiles of neighbouring positions, which share their 128-byte lines, meet in one L2
  const double* Qxp;        // 16 x 16 zero-padded matrices
  const double* Qyp;
  double theta, inv_theta, beta;
  double cbt;               // beta^theta (next_power_fast)
  const double* a3;         // aggregator scale: index = out_idx[o] + x * a3x + y * a3y + rest_idx[pos]
  const int* out_idx;
  const int* rest_idx;
  int a3x, a3y;
  int minus_identity;
};

struct SmallIO {
  const double* in;
  double* out;
  const double* aux_in;     // first pass of J.v: c1;  last pass of J.v: c2
  double* aux_out;          // linearising T: first pass writes c1, last pass c2;  fused form: the next intermediate
  const double* old;        // last pass of T: w (residual);  of J.v: v
  unsigned long long* resid;
  const unsigned long long* gate;
  double gate_tol;
  double* dotp;             // J.v with minus_identity: per-workgroup partial sums <out, v>, <out, out>: [2][gridDim.x]
  const double* dot_with;   // J.v: if given, dotp receives the single stream <out, dot_with> instead: [gridDim.x]
  unsigned long long* zero; // first pass: cleared by workgroup 0 (the residual word the last pass maximises into)
  // Successive approximation without atomics (225 workgroups maximising into one word cost 2.9 of the 7.5 us of a
  // fused kernel, tools/probes/small_fused_probe.hip): every workgroup stores its own maximum, the kernels of the
  // next iteration reduce those gate_n values themselves (their gate), and workgroup 0 leaves the result in slot_out
  // for the host.  A closed gate writes zeros, so it stays closed.
  const double* gate_part;
  int gate_n;
  double* part_out;
  unsigned long long* slot_out;
};

// LDS stride of X in the padded tile.  The contraction over Y reads 16 columns at once, one per x (R = 1) or four
// x times four positions (R = 4): at a stride of 16 (64) doubles that is an 8-way (4-way) bank conflict; 18 (72)
// doubles put the 16 columns on 16 different bank pairs.
template <int R> struct SmallLds { static constexpr int SX = R == 1 ? 18 : 72; static constexpr int TILE = 16 * SX; };

// element e of a tile -> LDS offset, global element offset against the tile base, a3 index part, position
template <int R>
__device__ __forceinline__ void small_decode(const SmallDesc& P, int e, int& l, unsigned& g, unsigned& ixy, int& r) {
  const int row = R == 1 ? e : (e >> 2);
  r = R == 1 ? 0 : (e & 3);
  const int x = (int)(__umul24((unsigned)row, P.my) >> 16);
  const int y = row - x * P.ny;
  l = R == 1 ? x * SmallLds<1>::SX + y : x * SmallLds<4>::SX + y * 4 + r;
  g = (unsigned)x * P.sx + (unsigned)y * P.sy + (unsigned)r;
  ixy = __umul24((unsigned)x, (unsigned)P.a3x) + __umul24((unsigned)y, (unsigned)P.a3y);
}

// WPT = waves per tile.  1: four wave-private tiles per workgroup, no workgroup barrier on the data path (many
// tiles: every SIMD has waves anyway).  4: one tile per workgroup, an element (R = 1) or four (R = 4) per thread:
// with at most a tile per CU the power -- ~140 instructions per point, issue-bound for a lone wave -- spreads
// over the CU's four SIMDs (tools/probes/small_fused_probe.hip: 1800 of the 13900 cycles of a fused kernel per
// power of four points per lane); the contractions' column tiles go to wave 0 (R = 1) or one to each wave (R = 4).
constexpr int SMALL_RING = 512;      // most workgroups of an end pass for which successive approximation runs without atomics

template <int MODE, int R, int WPT>
__device__ __forceinline__ void small_tile_body(const SmallDesc& P, const SmallIO& io, const AndArgs* an) {
  constexpr bool ANDF = MODE == SM_AND_FIRST;  // Anderson: control step of the previous pass + update of x, then the first pass of T
  constexpr bool ANDL = MODE == SM_AND_LAST;   // Anderson: last pass of T, then r = Tx - x into the history and its Gram row
  constexpr bool POWP = MODE == SM_FIRST_T || MODE == SM_FIRST_TLIN || ANDF;
  constexpr bool LINP = MODE == SM_FIRST_TLIN;
  constexpr bool MULP = MODE == SM_FIRST_J;
  constexpr bool FUSED = MODE == SM_FUSED_T;
  constexpr bool CES = MODE == SM_LAST_T || MODE == SM_LAST_TLIN || FUSED || ANDL;
  constexpr bool LINE = MODE == SM_LAST_TLIN;
  constexpr bool MULE = MODE == SM_LAST_J;
  constexpr int TILE = 256 * R;                // doubles per tile
  constexpr int TPT = 64 * WPT;                // threads per tile
  constexpr int EPL = TILE / TPT;              // elements per thread (padded tile)
  constexpr int PG = EPL < 4 ? EPL : 4;        // points per call of the power routine
  static_assert(R == 1 || R == 4, "run lengths");
  static_assert(WPT == 1 || WPT == 4, "waves per tile");
  constexpr int SX = SmallLds<R>::SX;           // LDS stride of X (conflict-free columns, see SmallLds)
  constexpr int LT = SmallLds<R>::TILE;         // doubles of LDS per tile
  __shared__ __attribute__((aligned(16))) double lds[(4 / WPT) * LT];
  __shared__ double red[12];
  constexpr int AND_LDS = (int)(sizeof(AndStepLds) / 8);
  __shared__ __attribute__((aligned(16))) double and_raw[ANDF ? AND_LDS : 1];         // the control step of wave 0
  __shared__ double redj[ANDL ? AND_FUSE_M * 256 : 1];
  // the gate word is fetched first and tested behind the tile loads (nothing is written before the test)
  const unsigned long long gate_word = io.gate != nullptr ? *io.gate : ~0ULL;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  double gate_err = 0.0;                       // the previous iteration's error, reduced from per-workgroup maxima
  if (io.gate_part != nullptr) {
    // all loads at once: the values were written by the previous kernel on other XCDs, a round trip each
    double gp[SMALL_RING / 64];
#pragma unroll
    for (int k = 0; k < SMALL_RING / 64; ++k) gp[k] = io.gate_part[lane + 64 * k < io.gate_n ? lane + 64 * k : 0];
#pragma unroll
    for (int k = 0; k < SMALL_RING / 64; ++k) gate_err = fmax(gate_err, gp[k]);
  }
  // Anderson, first pass.  The step of the previous pass is due; what the tiles need of it is known when the chunk is
  // captured (AndArgs::step_kind): nothing (0: the step can neither mix nor reject, x = fx; should it end the loop, the
  // launches behind this one are gated and x = fx is the final update anyway), whether the pass is rejected (1: the
  // diagonal entry of the Gram row), everything (2: a mixing step: wave 0 of every workgroup works it out, the same
  // operations everywhere).  The state is written by an extra workgroup that has no tile (the last one), off every
  // tile's path.  (The first form of this code, ring sums and solve unrolled per wave, was 130 KB of instructions that
  // every launch fetched cold -- tools/probes/anderson_fused_probe.hip: 15 us for the row sums alone.)
  int and_mode = 0, and_open = 1, and_prev = 0;
  // the whole step, by a workgroup: Gram row (all waves), control step and solve (wave 0), decision to every wave
  auto and_step_site = [&](bool writer, int nonfinite) {
    AndStepLds& sh = *reinterpret_cast<AndStepLds*>(and_raw);
    SDFS_AND_STAMP(0);
    const AndPreload pre = and_state_preload(an->Sin, an->m);
    and_row_sums16(an->partial, an->nb, an->m, -1, sh.row);
    and_state_park(sh, pre, an->m);
    __syncthreads();
    SDFS_AND_STAMP(1);
    if (wave == 0)
      and_step_wave<AND_FUSE_M + 1, true, true>(sh, lane, an->m, an->pos, an->rel, an->Sin, an->Sout, writer, an->err_slot, an->kind_slot, an->par, nonfinite);
    __syncthreads();
    SDFS_AND_STAMP(2);
    and_mode = __builtin_amdgcn_readfirstlane(sh.mix_mode);
    and_open = __builtin_amdgcn_readfirstlane(sh.open);
    and_prev = __builtin_amdgcn_readfirstlane(sh.prev_pos);
  };
  int and_nonfinite = 0;
  if (ANDF) {
    const bool ctrl = blockIdx.x == gridDim.x - 1;
    if (an->Sin->gate == 0ULL) {                // the loop has ended: carry its final state along
      if (ctrl && wave == 0) and_state_carry(lane, an->Sin, an->Sout);
      return;
    }
    const int nonfinite = (int)(*an->flag != 0u);
    and_nonfinite = nonfinite;
    if (!ctrl && an->step_kind == 1) {
      // the pass behind a mixing step: is it rejected?  (its push has recorded whether it met a non-finite residual)
      if (nonfinite) {
        const AndState* const Sp = an->Sin;
        const int pp = (int)Sp->prev_pos;
        if (Sp->last_mixed != 0.0 && pp >= 0 && Sp->rejected < 1000.0) {
          and_mode = 2; and_prev = pp;
          and_open = (sqrt(Sp->G[pp * an->m + pp]) > an->par.tol && Sp->it + 1.0 < an->par.max_iter) ? 1 : 0;
        }
      }
    } else if (ctrl) {
      and_step_site(true, nonfinite);
      return;
    }
  }
  double acc[ANDL ? AND_FUSE_M : 1];            // Anderson, last pass: <r, R_j> over this thread's elements
#pragma unroll
  for (int j = 0; j < (ANDL ? AND_FUSE_M : 1); ++j) acc[j] = 0.0;
  const int tl = WPT == 4 ? tid : lane;        // this thread's index inside its tile
  const unsigned wgi = P.cpx != 0u ? (blockIdx.x & 7u) * P.cpx + (blockIdx.x >> 3) : blockIdx.x;
  const long long t = WPT == 4 ? (long long)wgi : (long long)wgi * 4 + wave;
  SDFS_SMALL_STAMP(0);
  const bool active = t < P.ntiles;            // uniform per tile; idle waves (WPT = 1) fall through to the reductions
  if (ANDF && !active) return;                 // (no reductions in that form, and its workgroup barrier is for live waves)
  auto tile_sync = [&]() { if (WPT == 4) __syncthreads(); else wave_lds_fence(); };
  double rmax = 0.0, dot_yv = 0.0, dot_yy = 0.0;
  bool rnan = false;
  if (active) {
    const unsigned o = (unsigned)t / P.nchunks, chunk = (unsigned)t - o * P.nchunks;
    const long long base = (long long)o * P.ostride + (long long)chunk * R;
    const int total = P.nx * P.ny * R;
    const unsigned pos0 = chunk * R;
    double* const wl = WPT == 4 ? lds : lds + wave * LT;
    const bool need_old = CES ? (ANDL || io.resid != nullptr || io.part_out != nullptr) : (MULE && P.minus_identity);
    // ---- decode this thread's elements once ------------------------------------------------------------------
    int l[EPL], rr[EPL]; unsigned g[EPL], ia3[EPL]; bool ok[EPL];
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      const int e = tl + TPT * k;
      ok[k] = e < total;
      small_decode<R>(P, ok[k] ? e : 0, l[k], g[k], ia3[k], rr[k]);
      ok[k] = ok[k] && (R == 1 || pos0 + (unsigned)rr[k] < P.lrest);
      if (!ok[k]) g[k] = 0u;
    }
    // ---- loads: the tile and, for the last pass, its side streams -- all in flight at once; the a3 gather, whose
    // index comes out of two tables, goes last ---------------------------------------------------------------------
    const double* const inb = io.in + base;
    double v[EPL], s1[EPL], s2[EPL], s3[MULE ? EPL : 1];
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      if (TPT * k < total) {
        v[k] = inb[g[k]];
        if (MULP) s1[k] = io.aux_in[base + g[k]];
        if ((CES || MULE) && need_old) s1[k] = io.old[base + g[k]];
        if (MULE) s2[k] = io.aux_in[base + g[k]];
        if (MULE && io.dot_with != nullptr) s3[k] = io.dot_with[base + g[k]];
      }
    }
    QFrag<16> q, q2;
    const bool mfma_wave = WPT == 1 || R == 4 || wave == 0;
    if (mfma_wave) { q.load(P.Qxp, lane); q2.load(P.Qyp, lane); }
    if (CES) {
      const unsigned io0 = (unsigned)P.out_idx[o];
#pragma unroll
      for (int k = 0; k < EPL; ++k)
        if (TPT * k < total) s2[k] = P.a3[ok[k] ? io0 + (unsigned)P.rest_idx[pos0 + (unsigned)rr[k]] + ia3[k] : 0u];
    }
    // Anderson, last pass: the residual history of this thread's points travels with the tile (few elements per thread),
    // else it is requested group by group in front of the power.  Behind the a3 gather: loads return in order, and its
    // index tables would otherwise wait for these -- strided, slow -- streams before the gather could even be issued
    // (tools/probes/anderson_fused_probe.hip: 9100 cycles to that point against 2500).
    constexpr bool HIST_EARLY = ANDL && EPL <= 4;
    double hoe[HIST_EARLY ? AND_FUSE_M : 1][HIST_EARLY ? EPL : 1];
    if (HIST_EARLY) {
      // (a fixed number of requests from valid addresses, whatever m: the waits behind them stay counted ones)
#pragma unroll
      for (int jh = 0; jh < AND_FUSE_M; ++jh) {
        const double* const pr = an->h.R[jh < an->m ? jh : 0];
#pragma unroll
        for (int k = 0; k < EPL; ++k)
          if (k == 0 || TPT * k < total) hoe[HIST_EARLY ? jh : 0][HIST_EARLY ? k : 0] = pr[base + g[k]];
      }
    }
    SDFS_SMALL_STAMP(1);
    if (gate_word <= (unsigned long long)__double_as_longlong(io.gate_tol)) return;      // uniform over the grid
    SDFS_SMALL_STAMP(2);
    if (io.zero != nullptr && blockIdx.x == 0 && tid == 0) *io.zero = 0ULL;
    if (ANDF) {
      // ---- Anderson: this tile of the new iterate ------------------------------------------------------------------
      AndStepLds& sh = *reinterpret_cast<AndStepLds*>(and_raw);
      // the history is requested before the solve (only a pass whose step can mix needs it)
      double hx[ANDF ? AND_FUSE_M : 1][EPL];                     // Y_j = x_j + beta r_j (see AndArgs)
      const bool full = an->step_kind == 2;
      if (full) {
#pragma unroll
        for (int j = 0; j < AND_FUSE_M; ++j) {
          const double* const px = an->h.X[j < an->m ? j : 0];
#pragma unroll
          for (int k = 0; k < EPL; ++k)
            if (k == 0 || TPT * k < total) hx[ANDF ? j : 0][k] = px[base + g[k]];
        }
      }
      // a mixing step: every workgroup works it out (the same operations everywhere), with the tile and the history in flight
      if (full) and_step_site(false, and_nonfinite);
      const int mix_mode = and_mode;
      if (mix_mode == 1 && full) {
        // x = sum_j coef_j Y_j
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
          double xa = 0.0;
#pragma unroll
          for (int j = 0; j < AND_FUSE_M; ++j)
            if (j < an->m) xa = fma(sh.coef[j], hx[ANDF ? j : 0][k], xa);
          v[k] = xa;
        }
      } else if (mix_mode == 2) {
        // rejected step: x = x_prev + r_prev = Y_prev + (1 - beta) R_prev, the plain step from the last good iterate;
        // the poisoned slot is cleared (residual zero, Y the new iterate: finite, and out of every sum until it is refilled)
        const double* px = an->h.X[0];
        const double* pr = an->h.R[0];
#pragma unroll
        for (int j = 1; j < AND_FUSE_M; ++j)
          if (j == and_prev) { px = an->h.X[j]; pr = an->h.R[j]; }
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
          if (TPT * k < total) {
            v[k] = fma(1.0 - an->beta, pr[base + g[k]], px[base + g[k]]);
            if (ok[k]) { an->r_pos[base + g[k]] = 0.0; an->x_pos[base + g[k]] = v[k]; }
          }
        }
      }
#pragma unroll
      for (int k = 0; k < EPL; ++k)
        if (TPT * k < total && ok[k]) an->x[base + g[k]] = v[k];
      if (and_open == 0) return;      // uniform over the grid: the loop ended with that update
    }
    // ---- zero the padded tile, then park the data ------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < (LT / 2 + TPT - 1) / TPT; ++k)
      if (LT / 2 % TPT == 0 || tl + TPT * k < LT / 2) *reinterpret_cast<double2*>(wl + 2 * (tl + TPT * k)) = make_double2(0.0, 0.0);
    tile_sync();
    PowLane PT;
    if (POWP || CES) PT = pow_lane_init(lane);
    if (POWP) {
      // x = w^theta (c1 = w^(theta-1)); masked lanes feed the power 1
#pragma unroll
      for (int k = 0; k < EPL; k += PG) {
        if (TPT * k < total) {
          double xin[PG], xw[PG];
#pragma unroll
          for (int j = 0; j < PG; ++j) xin[j] = ok[k + j] ? v[k + j] : 1.0;
          pow_fast_n<true, PG>(xin, P.theta, PT, xw);
#pragma unroll
          for (int j = 0; j < PG; ++j) {
            v[k + j] = xw[j];
            if (LINP && ok[k + j]) io.aux_out[base + g[k + j]] = xw[j] / xin[j];
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      if (TPT * k < total) {
        if (MULP) v[k] *= s1[k];
        if (ok[k]) wl[l[k]] = v[k];
      }
    }
    tile_sync();
    const int li = lane & 15, lk = lane >> 4;
    // ---- contraction over X (row stride 16 R), then over Y (row stride R) --------------------------------------
    auto contract_pair = [&]() {
      if (R == 1) {
        if (mfma_wave) {
          ctile<16, SX>(wl + li + lk * SX, q);
          wave_lds_fence();
          ctile<16, 1>(wl + li * SX + lk, q2);
        }
      } else if (WPT == 1) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          if (4 * ct < P.ny) ctile<16, SX>(wl + 16 * ct + li + lk * SX, q);
        wave_lds_fence();
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          if (4 * ct < P.nx) ctile<16, 4>(wl + (4 * ct + (li >> 2)) * SX + (li & 3) + lk * 4, q2);
      } else {
        if (4 * wave < P.ny) ctile<16, SX>(wl + 16 * wave + li + lk * SX, q);
        __syncthreads();
        if (4 * wave < P.nx) ctile<16, 4>(wl + (4 * wave + (li >> 2)) * SX + (li & 3) + lk * 4, q2);
      }
      tile_sync();
    };
    SDFS_SMALL_STAMP(3);
    contract_pair();
    SDFS_SMALL_STAMP(4);
    if (io.gate_part != nullptr) {
      // tested here, behind the contractions: nothing has been written to memory yet
      gate_err = wave_max_f64(gate_err);
      if (blockIdx.x == 0 && tid == 0 && io.slot_out != nullptr) *io.slot_out = (unsigned long long)__double_as_longlong(gate_err);
      if (gate_err <= io.gate_tol) {                             // uniform over the grid
        if (tid == 0 && io.part_out != nullptr) io.part_out[blockIdx.x] = 0.0;
        return;
      }
    }
    // ---- epilogue -----------------------------------------------------------------------------------------------
    double* const outb = io.out + base;
    if (CES) {
      // Tw = 1 + beta (a3 S)^(1/theta), c2 = beta u / S, |Tw - w|
#pragma unroll
      for (int k = 0; k < EPL; k += PG) {
        if (TPT * k < total) {
          double sv[PG], ks[PG], uu[PG];
          // Anderson: the residual history of these points is requested in front of the power
          double ho[ANDL ? AND_FUSE_M : 1][PG];
          if (ANDL) {
#pragma unroll
            for (int jh = 0; jh < AND_FUSE_M; ++jh) {
              const double* const pr = an->h.R[jh < an->m ? jh : 0];
#pragma unroll
              for (int j = 0; j < PG; ++j) ho[ANDL ? jh : 0][j] = HIST_EARLY ? hoe[HIST_EARLY ? jh : 0][HIST_EARLY ? k + j : 0] : pr[base + g[k + j]];
            }
          }
#pragma unroll
          for (int j = 0; j < PG; ++j) { sv[j] = wl[l[k + j]]; ks[j] = ok[k + j] ? s2[k + j] * sv[j] : 1.0; }
          pow_fast_n<false, PG>(ks, P.inv_theta, PT, uu);
          double yv[PG];
#pragma unroll
          for (int j = 0; j < PG; ++j) {
            yv[j] = 1.0 + P.beta * uu[j];
            if (ok[k + j]) {
              const double y = yv[j];
              if (LINE) io.aux_out[base + g[k + j]] = P.beta * uu[j] / sv[j];
              if (need_old) {
                const double r0 = fabs(y - s1[k + j]);
                rnan |= (r0 != r0);
                rmax = fmax(rmax, r0);
              }
              outb[g[k + j]] = y;
              if (ANDL) {
                // Y[pos] = x + beta r, R[pos] = r = Tx - x, and this point's share of row `pos` of the Gram matrix
                const double xo = s1[k + j], r = y - xo;
                rnan |= !isfinite(r);
                an->x_pos[base + g[k + j]] = fma(an->beta, r, xo);
                an->r_pos[base + g[k + j]] = r;
#pragma unroll
                for (int jh = 0; jh < AND_FUSE_M; ++jh) acc[ANDL ? jh : 0] += r * (jh == an->pos ? r : ho[ANDL ? jh : 0][j]);    // (slots >= m: unused sums)
              }
            }
          }
          if (FUSED) {
            // the next application's first pass on the same pair: x = (Tw)^theta (next_power_fast, else the general routine)
            double bu[PG], xw[PG];
#pragma unroll
            for (int j = 0; j < PG; ++j) bu[j] = ok[k + j] ? P.beta * uu[j] : 1e4;
            if (!next_power_fast<PG>(bu, ks, P.theta, P.cbt, xw)) {
              double xin[PG];
#pragma unroll
              for (int j = 0; j < PG; ++j) xin[j] = ok[k + j] ? yv[j] : 1.0;
              pow_fast_n<true, PG>(xin, P.theta, PT, xw);
            }
#pragma unroll
            for (int j = 0; j < PG; ++j) v[k + j] = xw[j];
          }
        }
      }
      SDFS_SMALL_STAMP(5);
      if (FUSED) {
        // both contractions again, into the intermediate (the padding of the LDS tile is still exact zeros)
#pragma unroll
        for (int k = 0; k < EPL; ++k)
          if (TPT * k < total && ok[k]) wl[l[k]] = v[k];
        tile_sync();
        SDFS_SMALL_STAMP(6);
        contract_pair();
        SDFS_SMALL_STAMP(7);
        double* const tmpb = io.aux_out + base;
#pragma unroll
        for (int k = 0; k < EPL; ++k)
          if (TPT * k < total && ok[k]) tmpb[g[k]] = wl[l[k]];
      }
    } else {
#pragma unroll
      for (int k = 0; k < EPL; ++k) {
        if (TPT * k < total && ok[k]) {
          double y = wl[l[k]];
          if (MULE) {
            y *= s2[k];
            if (P.minus_identity) {
              y -= s1[k];
              if (io.dot_with == nullptr) {
                dot_yv = fma(y, s1[k], dot_yv);
                dot_yy = fma(y, y, dot_yy);
              }
            }
            if (io.dot_with != nullptr) dot_yv = fma(y, s3[MULE ? k : 0], dot_yv);
          }
          outb[g[k]] = y;
        }
      }
    }
  }
  SDFS_SMALL_STAMP(8);
  // ---- per-workgroup reductions (every wave of an open gate arrives) ------------------------------------------------
  if (gate_word <= (unsigned long long)__double_as_longlong(io.gate_tol)) return;
  if (io.gate_part != nullptr && !active) {                      // idle waves of a WPT = 1 workgroup: same test
    gate_err = wave_max_f64(gate_err);
    if (gate_err <= io.gate_tol) return;
  }
  if (MULE && io.dotp != nullptr) {
    dot_yv = wave_sum_f64(dot_yv); dot_yy = wave_sum_f64(dot_yy);
    if (lane == 0) { red[wave] = dot_yv; red[4 + wave] = dot_yy; }
    __syncthreads();
    if (tid == 0) {
      io.dotp[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
      if (io.dot_with == nullptr) io.dotp[gridDim.x + blockIdx.x] = (red[4] + red[5]) + (red[6] + red[7]);
    }
  }
  if (ANDL) {
    if (__any(rnan) && lane == 0) atomicOr(an->flag, 1u);          // (rare: a mixing step left the domain)
    // <r, R_j> of the workgroup through LDS: every thread parks its m sums, 16 lanes per stream add 16 each and meet
    // in one DPP row (ten wave reductions, DPP and v_readlane chains, were 2500 of the kernel's 9000 cycles)
#pragma unroll
    for (int j = 0; j < AND_FUSE_M; ++j)
      if (j < an->m) redj[ANDL ? j * 256 + tid : 0] = acc[ANDL ? j : 0];
    __syncthreads();
    if (wave < 3) {                                     // (m <= 12 streams of 16 lanes; all lanes of a participating wave: DPP)
      const int j = tid >> 4, q = tid & 15;
      double sj = 0.0;
      if (j < an->m) {
#pragma unroll
        for (int i = 0; i < 16; ++i) sj += redj[ANDL ? j * 256 + q + 16 * i : 0];
      }
      sj += dpp_mov_f64<0xB1>(sj);
      sj += dpp_mov_f64<0x4E>(sj);
      sj += dpp_mov_f64<0x141>(sj);
      sj += dpp_mov_f64<0x140>(sj);
      if (q == 0 && j < an->m) an->partial_out[blockIdx.x + j * gridDim.x] = sj;
    }
  }
  if (CES && (io.resid != nullptr || io.part_out != nullptr)) {
    if (rnan) rmax = __longlong_as_double(0x7ff0000000000000LL);                // NaN -> +inf
    rmax = wave_max_f64(rmax);
    if (lane == 0) red[8 + wave] = rmax;
    __syncthreads();
    if (tid == 0) {
      const double bm = fmax(fmax(red[8], red[9]), fmax(red[10], red[11]));
      if (io.part_out != nullptr) io.part_out[blockIdx.x] = bm;
      else atomicMax(io.resid, (unsigned long long)__double_as_longlong(bm));
    }
  }
  SDFS_SMALL_STAMP(9);
}

template <int MODE, int R, int WPT>
__global__ void __launch_bounds__(256)
small_tile_kernel(const SmallDesc P, const SmallIO io) { small_tile_body<MODE, R, WPT>(P, io, nullptr); }

// the Anderson forms: SM_AND_FIRST / SM_AND_LAST
template <int MODE, int R, int WPT>
__global__ void __launch_bounds__(256)
small_and_kernel(const SmallDesc P, const SmallIO io, const AndArgs an) { small_tile_body<MODE, R, WPT>(P, io, &an); }

// End of a chunk of the fused Anderson loop: the control step of the chunk's last pass and the update of x it decides on,
// with the row sums, the solve and the mixing expression of SM_AND_FIRST (the passes must not depend on where the chunks
// end).  Every workgroup works the step out, workgroup 0 writes the state; elements in their memory order.
__global__ void __launch_bounds__(256)
small_and_finish(const AndArgs an, const double* __restrict__ fx, long long n) {
  __shared__ __attribute__((aligned(16))) double and_raw[sizeof(AndStepLds) / 8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (an.Sin->gate == 0ULL) {
    if (blockIdx.x == 0 && wave == 0) and_state_carry(lane, an.Sin, an.Sout);
    return;
  }
  AndStepLds& sh = *reinterpret_cast<AndStepLds*>(and_raw);
  const int nonfinite = (int)(*an.flag != 0u);
  const AndPreload pre = and_state_preload(an.Sin, an.m);
  and_row_sums16(an.partial, an.nb, an.m, -1, sh.row);
  and_state_park(sh, pre, an.m);
  __syncthreads();
  if (wave == 0)
    and_step_wave<AND_FUSE_M + 1, true, true>(sh, lane, an.m, an.pos, an.rel, an.Sin, an.Sout, blockIdx.x == 0, an.err_slot, an.kind_slot, an.par, nonfinite);
  __syncthreads();
  const int mode = sh.mix_mode, prev = sh.prev_pos;
  const double* px = an.h.X[0];
  const double* pr = an.h.R[0];
#pragma unroll
  for (int j = 1; j < AND_FUSE_M; ++j)
    if (j == prev) { px = an.h.X[j]; pr = an.h.R[j]; }
  for (long long e = (long long)blockIdx.x * 256 + tid; e < n; e += (long long)gridDim.x * 256) {
    double v;
    if (mode == 0) v = fx[e];
    else if (mode == 1) {
      v = 0.0;
#pragma unroll
      for (int j = 0; j < AND_FUSE_M; ++j)
        if (j < an.m) v = fma(sh.coef[j], an.h.X[j][e], v);
    } else {
      v = fma(1.0 - an.beta, pr[e], px[e]);
      an.r_pos[e] = 0.0; an.x_pos[e] = v;
    }
    an.x[e] = v;
  }
}

// end of a chunk of iterations: the last iteration's error for the host
__global__ void __launch_bounds__(64) small_sa_finish(const double* part, int n, unsigned long long* slot) {
  double gp[SMALL_RING / 64];
#pragma unroll
  for (int k = 0; k < SMALL_RING / 64; ++k) gp[k] = part[(int)threadIdx.x + 64 * k < n ? (int)threadIdx.x + 64 * k : 0];
  double m = 0.0;
#pragma unroll
  for (int k = 0; k < SMALL_RING / 64; ++k) m = fmax(m, gp[k]);
  m = wave_max_f64(m);
  if (threadIdx.x == 0) *slot = (unsigned long long)__double_as_longlong(m);
}

// The plain first pass of T at N = 20: three slices per wave tile (9.6 KB of LDS per wave, 114 VGPRs: sixteen waves per
// CU instead of twelve; the last of the four column tiles repeats four columns).  Measured with the power routine and the
// non-temporal streams of round 4: 0.231-0.233 against 0.240 ms (profiles/round4_kernel_bench.txt); the other roles keep
// four slices (their streams and register sets differ).
#ifndef SDFS_SLICE16_G
#define SDFS_SLICE16_G 0
#endif
#ifndef SDFS_SLICE16_OCC
#define SDFS_SLICE16_OCC 3
#endif
template <int N> struct SliceTFirst { static constexpr int G = 0, OCC = 3; };
template <> struct SliceTFirst<20> { static constexpr int G = 3, OCC = 4; };
template <> struct SliceTFirst<16> { static constexpr int G = SDFS_SLICE16_G, OCC = SDFS_SLICE16_OCC; };
typedef void (*small_fn)(const SmallDesc, const SmallIO);
typedef void (*small_and_fn)(const SmallDesc, const SmallIO, const AndArgs);
typedef void (*slice_fn)(const SliceDesc, const SliceIO);
typedef void (*line_fn)(const LineDesc, const LineIO);

#ifndef SDFS_NO_VARIANT_TABLES
template <int R, int WPT> inline small_fn small_variant_rw(int mode) {
  switch (mode) {
    case SM_FIRST_T: return (small_fn)small_tile_kernel<SM_FIRST_T, R, WPT>;      // (R = 4: the fused Anderson loop opens on the slowest pair)
    case SM_FIRST_TLIN: return R == 1 ? (small_fn)small_tile_kernel<SM_FIRST_TLIN, 1, WPT> : nullptr;
    case SM_FIRST_J: return R == 1 ? (small_fn)small_tile_kernel<SM_FIRST_J, 1, WPT> : nullptr;
    case SM_MID: return (small_fn)small_tile_kernel<SM_MID, R, WPT>;
    case SM_LAST_T: return (small_fn)small_tile_kernel<SM_LAST_T, R, WPT>;
    case SM_LAST_TLIN: return (small_fn)small_tile_kernel<SM_LAST_TLIN, R, WPT>;
    case SM_LAST_J: return (small_fn)small_tile_kernel<SM_LAST_J, R, WPT>;
    case SM_FUSED_T: return (small_fn)small_tile_kernel<SM_FUSED_T, R, WPT>;
    default: return nullptr;
  }
}
// the first pass walks the two fastest axes: nothing lies behind Y, so its run length is 1
inline small_fn small_variant(int mode, int r, int wpt) {
  if (wpt == 4) return r == 1 ? small_variant_rw<1, 4>(mode) : (r == 4 ? small_variant_rw<4, 4>(mode) : nullptr);
  return r == 1 ? small_variant_rw<1, 1>(mode) : (r == 4 ? small_variant_rw<4, 1>(mode) : nullptr);
}
inline small_and_fn small_and_variant(int mode, int r, int wpt) {
  if (mode == SM_AND_FIRST) {
    if (wpt == 4) return r == 1 ? (small_and_fn)small_and_kernel<SM_AND_FIRST, 1, 4> : (r == 4 ? (small_and_fn)small_and_kernel<SM_AND_FIRST, 4, 4> : nullptr);
    return r == 1 ? (small_and_fn)small_and_kernel<SM_AND_FIRST, 1, 1> : nullptr;     // (runs of 4 on wave tiles: 16 elements per lane, the history of a mixing step would spill)
  }
  if (mode != SM_AND_LAST) return nullptr;
  if (wpt == 4) return r == 1 ? (small_and_fn)small_and_kernel<SM_AND_LAST, 1, 4> : (r == 4 ? (small_and_fn)small_and_kernel<SM_AND_LAST, 4, 4> : nullptr);
  return r == 1 ? (small_and_fn)small_and_kernel<SM_AND_LAST, 1, 1> : (r == 4 ? (small_and_fn)small_and_kernel<SM_AND_LAST, 4, 1> : nullptr);
}
inline unsigned small_grid(long long ntiles, int wpt) { return (unsigned)(wpt == 4 ? ntiles : (ntiles + 3) / 4); }
inline unsigned small_grid(const SmallDesc& d, int wpt) { return d.cpx != 0u ? 8u * d.cpx : small_grid(d.ntiles, wpt); }
inline unsigned small_cpx(long long ntiles, int wpt) { return (small_grid(ntiles, wpt) + 7u) / 8u; }


template <int N> inline slice_fn slice_variant_n(int mode, bool f32) {
  switch (mode) {
    case S_TFIRST: return f32 ? nullptr : (slice_fn)slice_kernel<N, S_TFIRST, false, 4, false, SliceTFirst<N>::G, SliceTFirst<N>::OCC>;
    case S_TFIRST_LIN: return f32 ? (slice_fn)slice_kernel<N, S_TFIRST_LIN, true> : (slice_fn)slice_kernel<N, S_TFIRST_LIN, false>;
    case S_JFIRST: return f32 ? (slice_fn)slice_kernel<N, S_JFIRST, true> : (slice_fn)slice_kernel<N, S_JFIRST, false>;
    case S_MID: return f32 ? nullptr : (slice_fn)slice_kernel<N, S_MID, false>;
    case S_TFIRST32: return f32 ? nullptr : (slice_fn)slice_kernel<N, S_TFIRST32, false, 4, false, SliceTFirst<N>::G, SliceTFirst<N>::OCC>;
    default: return nullptr;
  }
}
// f32: fp32 storage of the J.v streams / of c1 (opts.krylov_f32)
inline slice_fn slice_variant(int n, int mode, bool f32 = false) {
  switch (n) {
    case 16: return slice_variant_n<16>(mode, f32);
    case 20: return slice_variant_n<20>(mode, f32);
    case 24: return slice_variant_n<24>(mode, f32);
    case 32: return slice_variant_n<32>(mode, f32);
    default: return nullptr;
  }
}
template <int N, bool PERSIST, bool FULLC> inline line_fn line_variant_n(int mode) {
  switch (mode) {
    case L_MID: return (line_fn)line_kernel<N, L_MID, PERSIST, FULLC, false>;
    case L_TLAST: return (line_fn)line_kernel<N, L_TLAST, PERSIST, FULLC, false>;
    case L_TLAST_LIN: return (line_fn)line_kernel<N, L_TLAST_LIN, PERSIST, FULLC, false>;
    case L_JLAST: return (line_fn)line_kernel<N, L_JLAST, PERSIST, FULLC, false>;
    case L_TFUSED: if constexpr (!PERSIST) return (line_fn)line_kernel<N, L_TFUSED, false, FULLC, false>; else return nullptr;
    default: return nullptr;
  }
}
template <int N> inline line_fn line_variant_f32(int mode) {
  switch (mode) {
    case L_MID: return (line_fn)line_kernel<N, L_MID, false, true, true>;
    case L_TLAST_LIN: return (line_fn)line_kernel<N, L_TLAST_LIN, false, true, true>;
    case L_JLAST: return (line_fn)line_kernel<N, L_JLAST, false, true, true>;
    default: return nullptr;
  }
}
template <int N> inline line_fn line_variant_pf(int mode, bool persist, bool fullc, bool f32) {
  if (f32) return (fullc && !persist) ? line_variant_f32<N>(mode) : nullptr;
#ifdef SDFS_DIAG
  // round 2's persistent form (SDFS_LINE_PERSIST; superseded by stream_kernels.hpp): diagnostic builds only -- the shipped
  // library neither reads the knob nor carries these forty instantiations
  if (persist) return fullc ? line_variant_n<N, true, true>(mode) : line_variant_n<N, true, false>(mode);
#else
  if (persist) return nullptr;
#endif
  return fullc ? line_variant_n<N, false, true>(mode) : line_variant_n<N, false, false>(mode);
}
// fullc: lrest % 16 == 0 (no partial chunk anywhere in the pass); f32: fp32 storage of the J.v streams / of c2
inline line_fn line_variant(int n, int mode, bool persist, bool fullc, bool f32 = false) {
  switch (n) {
    case 16: return line_variant_pf<16>(mode, persist, fullc, f32);
    case 20: return line_variant_pf<20>(mode, persist, fullc, f32);
    case 24: return line_variant_pf<24>(mode, persist, fullc, f32);
    case 32: return line_variant_pf<32>(mode, persist, fullc, f32);
    default: return nullptr;
  }
}
#endif
// slices per wave tile / LDS bytes per workgroup of slice_variant(n, mode)
inline int slice_tile_slices(int n, int mode) {
  if ((mode == S_TFIRST || mode == S_TFIRST32) && n == 20) return SliceTFirst<20>::G;
  if ((mode == S_TFIRST || mode == S_TFIRST32) && n == 16 && SliceTFirst<16>::G) return SliceTFirst<16>::G;
  return n == 16 ? SliceGeo<16>::G : n == 20 ? SliceGeo<20>::G : n == 24 ? SliceGeo<24>::G : SliceGeo<32>::G;
}
inline size_t slice_lds_bytes(int n, int mode) { return (size_t)slice_tile_slices(n, mode) * n * (n == 20 ? n : n + 2) * 8 * 4; }
inline int line_block(int n) { return n <= 24 ? 256 : 512; }
inline size_t line_lds_bytes(int n) { return (size_t)n * n * LINE_R * 8; }
inline int line_blocks_per_cu(int n) { return n <= 16 ? LineGeo<16>::BPC : n == 20 ? LineGeo<20>::BPC : n == 24 ? LineGeo<24>::BPC : LineGeo<32>::BPC; }

}  // namespace sdfs
