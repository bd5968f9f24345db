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

#include "pass_kernel.hpp"

namespace sdfs {

// roles (compile time)
enum SliceMode { S_TFIRST = 0, S_TFIRST_LIN = 1, S_JFIRST = 2, S_NMODES = 3 };
enum LineMode { L_MID = 0, L_TLAST = 1, L_TLAST_LIN = 2, L_JLAST = 3, L_NMODES = 4 };

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
};

struct SliceIO {
  const double* in;         // T: w;  J.v: v
  double* out;
  const double* aux_in;     // J.v: c1
  double* aux_out;          // T + linearise: c1 = w^(theta-1)
  const unsigned long long* gate;
  double gate_tol;
};

template <int N> struct SliceGeo {
  static constexpr int G = N == 32 ? 2 : (N == 24 ? 2 : 4);     // slices per wave tile
  static constexpr int TILE = G * N * N;                          // doubles
  static constexpr int UNITS = TILE / 2;                          // double2 units
  static constexpr int EPT = (UNITS + 63) / 64;
  static constexpr int NCT = G * N / 16;                          // column tiles of either contraction
  static constexpr int WAVES = 4;
  static_assert((G * N) % 16 == 0, "whole column tiles");
};

template <int N, int MODE>
__global__ void __launch_bounds__(256, 3)
slice_kernel(const SliceDesc P, const SliceIO io) {
  using Geo = SliceGeo<N>;
  constexpr bool POWP = MODE == S_TFIRST || MODE == S_TFIRST_LIN;
  constexpr bool LIN = MODE == S_TFIRST_LIN;
  constexpr bool MULP = MODE == S_JFIRST;
  extern __shared__ double lds[];
  if (io.gate != nullptr) {
    const unsigned long long g = *io.gate;
    if (g <= (unsigned long long)__double_as_longlong(io.gate_tol)) return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform: addresses below stay scalar
  const long long tile = (long long)blockIdx.x * Geo::WAVES + wave;
  const long long s0 = tile * Geo::G;
  if (s0 >= P.nslices) return;                                   // no workgroup barrier below
  const long long rem = (P.nslices - s0) * (N * N / 2);          // valid units of a trailing partial tile
  const int nvalid = rem < Geo::UNITS ? (int)rem : Geo::UNITS;
  double* const wl = lds + wave * Geo::TILE;
  const long long gbase = s0 * (N * N);
  // uniform 64-bit bases + one 32-bit byte offset per lane: global_load/store with an SGPR base
  const char* const inb = reinterpret_cast<const char*>(io.in + gbase);
  const char* const auxb = reinterpret_cast<const char*>(io.aux_in + gbase);
  char* const outb = reinterpret_cast<char*>(io.out + gbase);
  char* const auxo = reinterpret_cast<char*>(io.aux_out + gbase);
  const unsigned lb = (unsigned)lane * 16u;

  // ---- loads (all in flight at once), Q fragments behind them --------------------------------------
  double2 v[Geo::EPT];
  double2 c1v[MULP ? Geo::EPT : 1];
#pragma unroll
  for (int k = 0; k < Geo::EPT; ++k) {
    const int u = lane + 64 * k;
    if (u < nvalid) {
      v[k] = *reinterpret_cast<const double2*>(inb + (lb + 1024u * k));
      if (MULP) c1v[MULP ? k : 0] = *reinterpret_cast<const double2*>(auxb + (lb + 1024u * k));
    } else {
      v[k] = make_double2(1.0, 1.0);
      if (MULP) c1v[MULP ? k : 0] = make_double2(0.0, 0.0);
    }
  }
  QFrag<N> qf;
  qf.load(P.Qf, lane);

  // ---- park the tile in LDS (linear image of the global layout); J.v: x = c1 * v on the way ------------
#pragma unroll
  for (int k = 0; k < Geo::EPT; ++k) {
    const int u = lane + 64 * k;
    if (MULP) { v[k].x *= c1v[MULP ? k : 0].x; v[k].y *= c1v[MULP ? k : 0].y; }
    if (Geo::UNITS % 64 == 0 || u < Geo::UNITS) *reinterpret_cast<double2*>(wl + 2 * u) = v[k];
  }
  wave_lds_fence();
  // ---- prologue x = w^theta in place (rolled: one copy of the power routine; every lane stays active) --
  if (POWP) {
    const PowLane PT = pow_lane_init(lane);
#pragma unroll 1
    for (int k = 0; k < Geo::EPT; ++k) {
      const int u = lane + 64 * k;
      const bool in_tile = Geo::UNITS % 64 == 0 || u < Geo::UNITS;
      const int lo = in_tile ? 2 * u : 0;
      const double2 x2 = *reinterpret_cast<const double2*>(wl + lo);
      const double xin[2] = {in_tile ? x2.x : 1.0, in_tile ? x2.y : 1.0};
      double xw[2];
      pow_fast_n<true, 2>(xin, P.theta, PT, xw);
      if (in_tile) {
        *reinterpret_cast<double2*>(wl + lo) = make_double2(xw[0], xw[1]);
        if (LIN && u < nvalid)                                    // c1 = w^(theta-1)
          *reinterpret_cast<double2*>(auxo + (lb + 1024u * k)) = make_double2(xw[0] / xin[0], xw[1] / xin[1]);
      }
    }
    wave_lds_fence();
  }

  const int li = lane & 15, lk = lane >> 4;
  // ---- contraction over the fastest axis: column c = (slice, e) at wl + c * N, rows contiguous --------
  {
    double* const p0 = wl + li * N + lk;
#pragma unroll
    for (int ct = 0; ct < Geo::NCT; ++ct) ctile<N, 1>(p0 + ct * 16 * N, qf);
  }
  wave_lds_fence();
  // ---- contraction over the second axis: column c = (slice g, f) at wl + g N^2 + f, row stride N -----
  {
    QFrag<N> qe;
    qe.load(P.Qe, lane);
#pragma unroll
    for (int ct = 0; ct < Geo::NCT; ++ct) {
      const int c = 16 * ct + li;
      const int g = c / N, f = c - g * N;
      ctile<N, N>(wl + g * (N * N) + f + lk * N, qe);
    }
  }
  wave_lds_fence();
  // ---- stream out -------------------------------------------------------------------------------------
#pragma unroll
  for (int k = 0; k < Geo::EPT; ++k) {
    const int u = lane + 64 * k;
    if (u < nvalid) *reinterpret_cast<double2*>(outb + (lb + 1024u * k)) = *reinterpret_cast<const double2*>(wl + 2 * u);
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
  // aggregator scale a3 (current state), index = out_idx[o] + x * a3x + y * a3y + rest_idx[pos]
  const double* a3;
  const int* out_idx;
  const int* rest_idx;
  int a3x, a3y;
  int minus_identity;
};

struct LineIO {
  const double* in;
  double* out;
  const double* aux_in;     // J.v: c2
  double* aux_out;          // T + linearise: c2 = beta u / S
  const double* old;        // T: w (residual);  J.v: v
  unsigned long long* resid;
  const unsigned long long* gate;
  double gate_tol;
  double* dotp;             // J.v with minus_identity: per-block partial sums <out, v>, <out, out>: [2][ntiles]
};

template <int N> struct LineGeo {
  static constexpr int B = N <= 20 ? 256 : 512;                   // threads per workgroup
  static constexpr int W = B / 64;
  static constexpr int UNITS = N * N * LINE_R / 2;                // double2 units per tile
  static constexpr int EPT = (UNITS + B - 1) / B;
  static constexpr int LX = N * LINE_R;                           // LDS stride of X (doubles); Y's is LINE_R
  static constexpr int BPC = N <= 16 ? 4 : (N == 20 ? 3 : (N == 24 ? 2 : 1));   // workgroups per CU (LDS)
  static_assert(N % W == 0, "column tiles split evenly over the waves");
};

template <int N, int MODE>
__global__ void __launch_bounds__(LineGeo<N>::B, LineGeo<N>::BPC * LineGeo<N>::B / 256)
line_kernel(const LineDesc P, const LineIO io) {
  using Geo = LineGeo<N>;
  constexpr int B = Geo::B;
  constexpr bool CES = MODE == L_TLAST || MODE == L_TLAST_LIN;
  constexpr bool LINE = MODE == L_TLAST_LIN;
  constexpr bool MULE = MODE == L_JLAST;
  constexpr bool PARTIAL = Geo::UNITS % B != 0;
  extern __shared__ double lds[];
  __shared__ double red[16];
  if (io.gate != nullptr) {
    const unsigned long long g = *io.gate;
    if (g <= (unsigned long long)__double_as_longlong(io.gate_tol)) return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned t = (unsigned)xcd_remap((long long)blockIdx.x, P.ntiles);      // ntiles < 2^31 (host check)
  const unsigned o = t / (unsigned)P.nchunks;
  const int chunk = (int)(t - o * (unsigned)P.nchunks);
  const int c2 = tid & 7;                                        // this thread's double2 inside the 16-double row
  const long long pos = (long long)chunk * LINE_R + 2 * c2;     // position in the contiguous remainder
  const bool cok = pos < P.lrest;                                // trailing partial chunk
  // unit u = tid + k B: row = u >> 3 = (x, y), global offset = row * lrest + pos (X, Y adjacent axes)
  // uniform 64-bit tile base + 32-bit byte offsets (the host checks that a tile spans < 4 GB)
  const long long tbase = (long long)o * (N * N) * P.lrest + (long long)chunk * LINE_R;
  const unsigned b0 = ((unsigned)(tid >> 3) * (unsigned)P.lrest + 2u * c2) * 8u;
  const unsigned bstep = (unsigned)(B / 8) * (unsigned)P.lrest * 8u;
  const char* const inb = reinterpret_cast<const char*>(io.in + tbase);
  char* const outb = reinterpret_cast<char*>(io.out + tbase);
  const char* const oldb = reinterpret_cast<const char*>(io.old + tbase);
  const char* const auxb = reinterpret_cast<const char*>(io.aux_in + tbase);
  char* const auxo = reinterpret_cast<char*>(io.aux_out + tbase);

  double2 v[Geo::EPT];
#pragma unroll
  for (int k = 0; k < Geo::EPT; ++k) {
    const bool ok = cok && (!PARTIAL || tid + k * B < Geo::UNITS);
    v[k] = ok ? *reinterpret_cast<const double2*>(inb + (b0 + k * bstep)) : make_double2(0.0, 0.0);
  }
  QFrag<N> q;
  q.load(P.Qx, lane);
#pragma unroll
  for (int k = 0; k < Geo::EPT; ++k)
    if (!PARTIAL || tid + k * B < Geo::UNITS) *reinterpret_cast<double2*>(lds + 2 * (tid + k * B)) = v[k];
  // the residual's w / the J.v's v: in flight across the contractions
  constexpr bool EARLY_OLD = CES || MULE;
  double2 oldv[EARLY_OLD ? Geo::EPT : 1];
  const bool need_old = CES ? (io.resid != nullptr) : (MULE && P.minus_identity);
  if (EARLY_OLD && need_old) {
#pragma unroll
    for (int k = 0; k < Geo::EPT; ++k) {
      const bool ok = cok && (!PARTIAL || tid + k * B < Geo::UNITS);
      oldv[EARLY_OLD ? k : 0] = ok ? *reinterpret_cast<const double2*>(oldb + (b0 + k * bstep)) : make_double2(0.0, 0.0);
    }
  }
  __syncthreads();

  const int li = lane & 15, lk = lane >> 4;
  // ---- contraction over X: column = (y, r) = LDS offset, row stride LX ---------------------------------
  {
    double* const p0 = lds + li + lk * Geo::LX;
#pragma unroll
    for (int j = 0; j < N / Geo::W; ++j) ctile<N, Geo::LX>(p0 + (wave + j * Geo::W) * 16, q);
  }
  q.load(P.Qy, lane);
  __syncthreads();
  // ---- contraction over Y: column = (x, r) at x * LX + r, row stride 16 --------------------------------
  {
    double* const p0 = lds + li + lk * LINE_R;
#pragma unroll
    for (int j = 0; j < N / Geo::W; ++j) ctile<N, LINE_R>(p0 + (wave + j * Geo::W) * Geo::LX, q);
  }
  __syncthreads();

  // ---- epilogue and store ---------------------------------------------------------------------------------
  double rmax = 0.0, dot_yv = 0.0, dot_yy = 0.0;
  if (CES) {
    const PowLane PT = pow_lane_init(lane);
    // a3 index of this thread's two elements: outer part + remainder part (host tables), (x, y) part per unit
    const unsigned ia3a = cok ? (unsigned)(P.out_idx[o] + P.rest_idx[pos]) : 0u;
    const unsigned ia3b = cok ? (unsigned)(P.out_idx[o] + P.rest_idx[pos + 1]) : 0u;
    const char* const a3b = reinterpret_cast<const char*>(P.a3);
    // one unit of the aggregator: Tw = 1 + beta (a3 S)^(1/theta), c2 = beta u / S, |Tw - w|, store.
    // FULL = false: the straight-line power (flags lanes it cannot serve); FULL = true: the full-range routine.
    auto unit = [&](const int k, const double2 oldk, auto full_tag) -> bool {
      constexpr bool FULL = decltype(full_tag)::value;
      const int u = tid + k * B;
      const bool rowok = !PARTIAL || u < Geo::UNITS;
      const bool ok = cok && rowok;
      const int row = rowok ? (u >> 3) : 0;
      const int x = row / N, y = row - x * N;
      const unsigned ixy = __umul24((unsigned)x, (unsigned)P.a3x) + __umul24((unsigned)y, (unsigned)P.a3y);   // strides < 2^24 (host check)
      const double2 sv = rowok ? *reinterpret_cast<const double2*>(lds + 2 * u) : make_double2(1.0, 1.0);
      double ks[2], uu[2], eh[2];
      // unconditional gathers (index 0 when masked): no branch, so the loads of later units move up
      const double k0 = *reinterpret_cast<const double*>(a3b + (ok ? (ia3a + ixy) * 8u : 0u));
      const double k1 = *reinterpret_cast<const double*>(a3b + (ok ? (ia3b + ixy) * 8u : 0u));
      ks[0] = ok ? k0 * sv.x : 1.0;
      ks[1] = ok ? k1 * sv.y : 1.0;
      bool rare = false;
      if (FULL) { uu[0] = pow_full<false>(ks[0], P.inv_theta, PT); uu[1] = pow_full<false>(ks[1], P.inv_theta, PT); }
      else rare = pow_fast_try<false, 2>(ks, P.inv_theta, PT, uu, eh);
      const double2 y2 = make_double2(1.0 + P.beta * uu[0], 1.0 + P.beta * uu[1]);
      if (ok) {
        if (LINE) *reinterpret_cast<double2*>(auxo + (b0 + k * bstep)) =
            make_double2(P.beta * uu[0] / sv.x, P.beta * uu[1] / sv.y);
        if (need_old) {
          double r0 = fabs(y2.x - oldk.x), r1 = fabs(y2.y - oldk.y);
          if (!(r0 == r0)) r0 = __longlong_as_double(0x7ff0000000000000LL);   // NaN -> +inf
          if (!(r1 == r1)) r1 = __longlong_as_double(0x7ff0000000000000LL);
          rmax = fmax(rmax, fmax(r0, r1));
        }
        *reinterpret_cast<double2*>(outb + (b0 + k * bstep)) = y2;
      }
      return rare;
    };
    bool rare = false;
#pragma unroll
    for (int k = 0; k < Geo::EPT; ++k) rare |= unit(k, oldv[EARLY_OLD ? k : 0], std::false_type{});
    // x <= 0, NaN, Inf, subnormal or out-of-range inputs anywhere in the wave: redo its units with the full
    // routine (rolled, one copy; every lane stays active for the table gathers)
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(rare) != 0ULL, 0)) {
      rmax = 0.0;
#pragma unroll 1
      for (int k = 0; k < Geo::EPT; ++k) {
        const bool ok = cok && (!PARTIAL || tid + k * B < Geo::UNITS);
        const double2 oldk = (ok && need_old) ? *reinterpret_cast<const double2*>(oldb + (b0 + k * bstep)) : make_double2(0.0, 0.0);
        unit(k, oldk, std::true_type{});
      }
    }
  } else {
    double2 c2v[MULE ? Geo::EPT : 1];
    if (MULE) {
#pragma unroll
      for (int k = 0; k < Geo::EPT; ++k) {
        const bool ok = cok && (!PARTIAL || tid + k * B < Geo::UNITS);
        c2v[MULE ? k : 0] = ok ? *reinterpret_cast<const double2*>(auxb + (b0 + k * bstep)) : make_double2(0.0, 0.0);
      }
    }
#pragma unroll
    for (int k = 0; k < Geo::EPT; ++k) {
      const int u = tid + k * B;
      const bool ok = cok && (!PARTIAL || u < Geo::UNITS);
      if (ok) {
        double2 y2 = *reinterpret_cast<const double2*>(lds + 2 * u);
        if (MULE) {
          y2.x *= c2v[MULE ? k : 0].x; y2.y *= c2v[MULE ? k : 0].y;
          if (P.minus_identity) {
            const double2 ov = oldv[EARLY_OLD ? k : 0];
            y2.x -= ov.x; y2.y -= ov.y;
            dot_yv = fma(y2.x, ov.x, dot_yv); dot_yv = fma(y2.y, ov.y, dot_yv);
            dot_yy = fma(y2.x, y2.x, dot_yy); dot_yy = fma(y2.y, y2.y, dot_yy);
          }
        }
        *reinterpret_cast<double2*>(outb + (b0 + k * bstep)) = y2;
      }
    }
  }

  if (MULE && io.dotp != nullptr) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) { dot_yv += __shfl_xor(dot_yv, s); dot_yy += __shfl_xor(dot_yy, s); }
    if (lane == 0) { red[wave] = dot_yv; red[8 + wave] = dot_yy; }
    __syncthreads();
    if (tid == 0) {
      double a = 0.0, b = 0.0;
      for (int w = 0; w < Geo::W; ++w) { a += red[w]; b += red[8 + w]; }
      io.dotp[blockIdx.x] = a;
      io.dotp[P.ntiles + blockIdx.x] = b;
    }
  }
  if (CES && io.resid != nullptr) {
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

typedef void (*slice_fn)(const SliceDesc, const SliceIO);
typedef void (*line_fn)(const LineDesc, const LineIO);

template <int N> inline slice_fn slice_variant_n(int mode) {
  switch (mode) {
    case S_TFIRST: return (slice_fn)slice_kernel<N, S_TFIRST>;
    case S_TFIRST_LIN: return (slice_fn)slice_kernel<N, S_TFIRST_LIN>;
    case S_JFIRST: return (slice_fn)slice_kernel<N, S_JFIRST>;
    default: return nullptr;
  }
}
inline slice_fn slice_variant(int n, int mode) {
  switch (n) {
    case 16: return slice_variant_n<16>(mode);
    case 20: return slice_variant_n<20>(mode);
    case 24: return slice_variant_n<24>(mode);
    case 32: return slice_variant_n<32>(mode);
    default: return nullptr;
  }
}
template <int N> inline line_fn line_variant_n(int mode) {
  switch (mode) {
    case L_MID: return (line_fn)line_kernel<N, L_MID>;
    case L_TLAST: return (line_fn)line_kernel<N, L_TLAST>;
    case L_TLAST_LIN: return (line_fn)line_kernel<N, L_TLAST_LIN>;
    case L_JLAST: return (line_fn)line_kernel<N, L_JLAST>;
    default: return nullptr;
  }
}
inline line_fn line_variant(int n, int mode) {
  switch (n) {
    case 16: return line_variant_n<16>(mode);
    case 20: return line_variant_n<20>(mode);
    case 24: return line_variant_n<24>(mode);
    case 32: return line_variant_n<32>(mode);
    default: return nullptr;
  }
}
inline int slice_tile_slices(int n) { return n == 16 ? SliceGeo<16>::G : n == 20 ? SliceGeo<20>::G : n == 24 ? SliceGeo<24>::G : SliceGeo<32>::G; }
inline size_t slice_lds_bytes(int n) { return (size_t)slice_tile_slices(n) * n * n * 8 * 4; }
inline int line_block(int n) { return n <= 20 ? 256 : 512; }
inline size_t line_lds_bytes(int n) { return (size_t)n * n * LINE_R * 8; }

}  // namespace sdfs
