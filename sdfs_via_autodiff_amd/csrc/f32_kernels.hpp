// f32_kernels.hpp -- the Jacobian-vector product of the pair plan with an fp32 LDS tile and fp32 MFMA (round 4).
//
// BASELINE config 5 ("fp32/bf16 mixed precision ON MFMA with fp64 residual accumulation").  The reference is fp64 only
// (code/solvers.py:9-11); what can run at reduced precision without touching the fixed point is the INNER solve of the
// Newton step, the matrix-free J.v of code/solvers.py:87 inside BiCGSTAB (:91-93): an inexact Newton method only needs
// its linear system solved to the inner tolerance, while the outer residual T(x) - x, the iterate and every reduction
// stay fp64.  Rounds 1-3 stored the Krylov vectors, c1, c2 and the J.v intermediates as floats (opts.krylov_f32 = 1)
// but still widened every tile to fp64 in LDS and contracted it with v_mfma_f64.  Here (opts.krylov_f32 = 3):
//   * the tile is parked in LDS as floats -- half the LDS bytes, and a line tile of 25.6 KB admits six workgroups per CU
//     where the fp64 tile (51.2 KB) admits three;
//   * the contractions run on v_mfma_f32_16x16x4_f32 (32 cycles per 16 x 16 x 4 block against 64 for the fp64 shape;
//     lane maps probed on hardware, tools/probes/mfma_f32_probe.hip), accumulating in fp32;
//   * extents that are not a multiple of 16 (n = 20, 24) take a second row tile whose A fragment is zero beyond row n
//     (20: 2 x 5 MFMAs of 32 cycles = 320 per 16 columns against 5 x 64 + 5 x 20 = 420 in fp64; 16: 160 against 320);
//   * the dot products of the last pass (<t, s>, <t, t> of BiCGSTAB) are summed in fp64, as everywhere.
// Rounding: one stored float per stream as before, plus fp32 products and sums inside the three passes: ~1e-6 relative
// on J.v, against inner tolerances of 1e-4 .. 1e-6 -- tools/mixed_precision_sweep.py measures what that costs.
#pragma once
#include "fast_kernels.hpp"

namespace sdfs {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int N> struct MShape32 {
  static_assert(N == 16 || N == 20 || N == 24 || N == 32, "pair plan extents");
  static constexpr int NT = (N + 15) / 16;       // row tiles of 16
  static constexpr int KT = N / 4;               // k-steps of 4
};

// A operand of v_mfma_f32_16x16x4_f32: lane l holds A[row l & 15][k l >> 4]; rows beyond N are zero
template <int N> struct QFrag32 {
  float a[MShape32<N>::NT][MShape32<N>::KT];
  __device__ __forceinline__ void load(const double* __restrict__ Q, int lane) {
    using S = MShape32<N>;
    const int li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int t = 0; t < S::NT; ++t) {
      const int row = 16 * t + li;
#pragma unroll
      for (int kk = 0; kk < S::KT; ++kk) a[t][kk] = row < N ? (float)Q[(row < N ? row : 0) * N + 4 * kk + lk] : 0.f;
    }
  }
};

// One column tile in place in an fp32 LDS tile: y[:, 16 columns] = Q x[:, 16 columns].  `col` points at the tile's
// element (row 0, this lane's column li); RS = row stride in floats.  B operand: lane l reads row 4 kk + (l >> 4);
// D: lane l, register i holds row 4 (l >> 4) + i of its column.  Every row is read before any is written.
template <int N, int RS>
__device__ __forceinline__ void ctile32(float* __restrict__ col, const int lk, const QFrag32<N>& q) {
  using S = MShape32<N>;
  float b[S::KT];
  const float* const pr = col + lk * RS;
#pragma unroll
  for (int kk = 0; kk < S::KT; ++kk) b[kk] = pr[4 * kk * RS];
  v4f acc[S::NT];
#pragma unroll
  for (int t = 0; t < S::NT; ++t) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kk = 0; kk < S::KT; ++kk) {
#pragma unroll
    for (int t = 0; t < S::NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(q.a[t][kk], b[kk], acc[t], 0, 0, 0);
  }
  float* const pw = col + 4 * lk * RS;
#pragma unroll
  for (int t = 0; t < S::NT; ++t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row0 = 16 * t + i;                       // + 4 lk
      if (16 * t + 12 + i < N || row0 + 4 * lk < N) pw[row0 * RS] = acc[t][i];      // (first test: compile time, whole tile inside)
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// slice32_kernel: first pass of J.v on the two fastest axes, x = c1 * v in fp32.  A wave owns G consecutive n x n slices
// in a private LDS region of floats, row stride N + 2 (16 rows read at once then hit 32 different banks).
template <int N> struct Slice32Geo {
  static constexpr int G = N == 32 ? 2 : 4;
  static constexpr int RS = N + 2;
  static constexpr int LTILE = G * N * RS;               // floats
  static constexpr int TILE = G * N * N;
  static constexpr int UNITS4 = TILE / 4;                // float4 units
  static constexpr int EPT4 = (UNITS4 + 63) / 64;
  static constexpr int NCT = (G * N + 15) / 16;
  static constexpr int CPAD = NCT * 16 - G * N;          // repeated columns of the last column tile (see SliceGeo)
  static constexpr int WAVES = 4;
  static_assert(CPAD <= 8, "the repeated columns lie inside the last column tile");
};
inline size_t slice32_lds_bytes(int n) { return (size_t)(n == 32 ? 2 : 4) * n * (n + 2) * 4 * 4; }
inline int slice32_tile_slices(int n) { return n == 32 ? 2 : 4; }

template <int N>
__global__ void __launch_bounds__(256, 4)
slice32_kernel(const SliceDesc P, const SliceIO io) {
  using Geo = Slice32Geo<N>;
  extern __shared__ double lds_[];
  float* const lds = reinterpret_cast<float*>(lds_);
  if (io.gate != nullptr) {
    const unsigned long long g = *io.gate;
    if (g <= (unsigned long long)__double_as_longlong(io.gate_tol)) return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long tile = (long long)blockIdx.x * Geo::WAVES + wave;
  const long long s0 = tile * Geo::G;
  if (s0 >= P.nslices) return;
  const long long rem = (P.nslices - s0) * (N * N / 4);
  const int nvalid4 = rem < Geo::UNITS4 ? (int)rem : Geo::UNITS4;
  float* const wl = lds + wave * Geo::LTILE;
  auto lofs = [](const int e) -> int { return (e / N) * Geo::RS + (e % N); };
  const long long gbase = s0 * (N * N);
  const unsigned lb = (unsigned)lane * 16u;
  const char* const inb = reinterpret_cast<const char*>(reinterpret_cast<const float*>(io.in) + gbase);
  const char* const auxb = reinterpret_cast<const char*>(reinterpret_cast<const float*>(io.aux_in) + gbase);
  v4f v[Geo::EPT4], c1v[Geo::EPT4];
#pragma unroll
  for (int k = 0; k < Geo::EPT4; ++k) {
    // (a unit beyond the tile, or in a slice a trailing partial tile does not have, re-reads unit 0's piece: finite
    // values that stay inside slices which are never stored)
    const unsigned off = (lane + 64 * k < nvalid4) ? lb + 1024u * k : lb;
    v[k] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(inb + off));
    c1v[k] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(auxb + off));
  }
  QFrag32<N> qf;
  qf.load(P.Qf, lane);
#pragma unroll
  for (int k = 0; k < Geo::EPT4; ++k) {
    const int u = lane + 64 * k;
    if (Geo::UNITS4 % 64 == 0 || u < Geo::UNITS4) {
      const v4f x = v[k] * c1v[k];
      float* const p = wl + lofs(4 * u);                 // rows are 8-byte aligned (N + 2 is even): two 8-byte stores
      *reinterpret_cast<v2f*>(p) = (v2f){x.x, x.y};
      *reinterpret_cast<v2f*>(p + 2) = (v2f){x.z, x.w};
    }
  }
  wave_lds_fence();
  const int li = lane & 15, lk = lane >> 4;
  // contraction over the fastest axis: column c = LDS row c (its N elements contiguous: "row stride" 1)
#pragma unroll
  for (int ct = 0; ct < Geo::NCT; ++ct) {
    const bool rep = Geo::CPAD > 0 && ct == Geo::NCT - 1;
    ctile32<N, 1>(wl + (ct * 16 + li - ((rep && li >= 16 - Geo::CPAD) ? Geo::CPAD : 0)) * Geo::RS, lk, qf);
    __builtin_amdgcn_sched_barrier(0);
  }
  wave_lds_fence();
  {
    QFrag32<N> qe;
    qe.load(P.Qe, lane);
#pragma unroll
    for (int ct = 0; ct < Geo::NCT; ++ct) {
      const bool rep = Geo::CPAD > 0 && ct == Geo::NCT - 1;
      const int c = 16 * ct + li - ((rep && li >= 16 - Geo::CPAD) ? Geo::CPAD : 0);
      const int g = c / N, f = c - g * N;
      ctile32<N, Geo::RS>(wl + g * (N * Geo::RS) + f, lk, qe);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  wave_lds_fence();
  char* const outb = reinterpret_cast<char*>(reinterpret_cast<float*>(io.out) + gbase);
#pragma unroll
  for (int k = 0; k < Geo::EPT4; ++k) {
    const int u = lane + 64 * k;
    if (u < nvalid4) {
      const float* const p = wl + lofs(4 * u);
      const v2f a = *reinterpret_cast<const v2f*>(p), b = *reinterpret_cast<const v2f*>(p + 2);
      *reinterpret_cast<v4f*>(outb + (lb + 1024u * k)) = (v4f){a.x, a.y, b.x, b.y};      // (default policy: the next pass reads it, pass_kernel.hpp SDFS_NT)
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// line32_kernel: middle / last pass of J.v on a slower pair, whole chunks only.  Tile = all n x n (x, y) rows of one chunk
// of R floats of the contiguous remainder behind the pair: R = 32 (one 128-byte line per row, 51 KB of LDS at n = 20:
// three workgroups per CU) where the remainder is a multiple of 32 elements, else R = 16 (64-byte rows, 25.6 KB: six
// workgroups per CU) -- at GCY 20^6 the pair (z, z_pi) takes 32, the pair (h_z, h_c), whose remainder is 400, takes 16.
// LDS: row (x, y) at (x * N + y) * R floats; unit u = tid + k B is float4 (u % (R / 4)) of row u / (R / 4).
template <int N, int R> struct Line32Geo {
  static_assert(R == 16 || R == 32, "row lengths of the fp32 line tiles");
  static constexpr int B = 256;
  static constexpr int W = B / 64;
  static constexpr int Q4 = R / 4;                       // float4 units per row
  static constexpr int UNITS4 = N * N * Q4;
  static constexpr int EPT4 = (UNITS4 + B - 1) / B;
  static constexpr int LX = N * R;                       // floats between two x
  static constexpr int LDS_BYTES = N * N * R * 4;
  static constexpr int BPC = 163840 / LDS_BYTES > 8 ? 8 : 163840 / LDS_BYTES;       // workgroups per CU (LDS)
  static constexpr int OCC = BPC >= 6 ? 6 : (BPC < 1 ? 1 : BPC);                    // waves per SIMD the register budget is set for
};
inline size_t line32_lds_bytes(int n, int r) { return (size_t)n * n * r * 4; }

// DOT3: the last pass also sums <out, io.dot_with> (fused BiCGSTAB iteration, krylov_kernels.hpp).  A variant of its own:
// the third side stream takes the 64-byte-row kernel from 52 to 112 VGPRs and from six to four workgroups per CU (229 ->
// 300 us at GCY 20^6), which the other application of an iteration -- no third sum -- need not pay.
template <int N, int MODE, int R, bool DOT3 = false>
__global__ void __launch_bounds__(256, ((MODE == L_JLAST && Line32Geo<N, R>::OCC > 4) ? 4 : Line32Geo<N, R>::OCC))   // (the last pass holds two side streams)
line32_kernel(const LineDesc P, const LineIO io) {
  using Geo = Line32Geo<N, R>;
  static_assert(MODE == L_MID || MODE == L_JLAST, "J.v roles");
  constexpr bool MUL = MODE == L_JLAST;
  constexpr int B = Geo::B, EPT4 = Geo::EPT4, Q4 = Geo::Q4;
  constexpr bool PART4 = Geo::UNITS4 % B != 0;
  extern __shared__ double lds_[];
  float* const lds = reinterpret_cast<float*>(lds_);
  __shared__ double red4[16];
  if (io.gate != nullptr) {
    const unsigned long long g = *io.gate;
    if (g <= (unsigned long long)__double_as_longlong(io.gate_tol)) return;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  // tiles: chunks of R elements (LineDesc counts chunks of 16: nchunks * 16 / R of them per outer index)
  const unsigned cpo = (unsigned)P.nchunks * LINE_R / R;
  const long long ntiles = P.nouter * (long long)cpo;
  const unsigned t = (unsigned)xcd_remap((long long)blockIdx.x, ntiles);
  const unsigned o = t / cpo;
  const int chunk = (int)(t - o * cpo);
  const long long tbase = (long long)o * (N * N) * P.lrest + (long long)chunk * R;
  const unsigned b0 = ((unsigned)(tid / Q4) * (unsigned)P.lrest + 4u * (tid % Q4)) * 4u;
  const unsigned bstep = (unsigned)(B / Q4) * (unsigned)P.lrest * 4u;
  const char* const inb = reinterpret_cast<const char*>(reinterpret_cast<const float*>(io.in) + tbase);
  const char* const auxb = reinterpret_cast<const char*>(reinterpret_cast<const float*>(io.aux_in) + tbase);
  const char* const oldb = reinterpret_cast<const char*>(reinterpret_cast<const float*>(io.old) + tbase);
  char* const outb = reinterpret_cast<char*>(reinterpret_cast<float*>(io.out) + tbase);
  const bool need_old = MUL && P.minus_identity;
  // (non-temporal only where a row is a whole 128-byte line: with 64-byte rows the other half of the line belongs to the
  // neighbouring tile, and an nt load does not keep it in L2 for that tile -- measured: 1.65 GB fetched for 1.02)
  auto ldrow = [](const char* p) -> v4f {
    if (R == 32) return __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
    return *reinterpret_cast<const v4f*>(p);
  };
  v4f v[EPT4];
#pragma unroll
  for (int k = 0; k < EPT4; ++k) {
    const bool rowok = !PART4 || tid + k * B < Geo::UNITS4;
    v[k] = ldrow(inb + (rowok ? b0 + k * bstep : b0));
  }
  QFrag32<N> q;
  q.load(P.Qx, lane);
#pragma unroll
  for (int k = 0; k < EPT4; ++k) {
    const int u = tid + k * B;
    if (!PART4 || u < Geo::UNITS4) *reinterpret_cast<v4f*>(lds + 4 * u) = v[k];
  }
  // side streams of the J.v epilogue: v (for "- v" and the dots) travels across the contractions, c2 is fetched behind
  // them where both would not fit the register budget (R = 32: thirteen quads per stream)
  constexpr bool C2_EARLY = EPT4 <= 8;
  v4f c2v[MUL ? EPT4 : 1], oldv[MUL ? EPT4 : 1];
  static_assert(!DOT3 || MUL, "the third sum belongs to the last pass");
  constexpr bool dot3 = DOT3;
  const char* const rwb = reinterpret_cast<const char*>(reinterpret_cast<const float*>(DOT3 ? io.dot_with : nullptr) + tbase);
  v4f rwv[(DOT3 && C2_EARLY) ? EPT4 : 1];
  if (MUL) {
#pragma unroll
    for (int k = 0; k < EPT4; ++k) {
      const bool rowok = !PART4 || tid + k * B < Geo::UNITS4;
      if (C2_EARLY) c2v[MUL ? k : 0] = ldrow(auxb + (rowok ? b0 + k * bstep : b0));
      if (need_old) oldv[MUL ? k : 0] = ldrow(oldb + (rowok ? b0 + k * bstep : b0));
      if (C2_EARLY && dot3) rwv[(DOT3 && C2_EARLY) ? k : 0] = ldrow(rwb + (rowok ? b0 + k * bstep : b0));
    }
  }
  __syncthreads();
  // contraction over X: column (y, r) at y * R + r, row stride LX; N * R / 16 column tiles go round the four waves
  constexpr int NCT = N * R / 16;
#pragma unroll
  for (int j = 0; j < NCT / Geo::W; ++j) { ctile32<N, Geo::LX>(lds + (wave + j * Geo::W) * 16 + li, lk, q); __builtin_amdgcn_sched_barrier(0); }
  q.load(P.Qy, lane);
  __syncthreads();
  // contraction over Y: column (x, r) at x * LX + r, row stride R
#pragma unroll
  for (int j = 0; j < NCT / Geo::W; ++j) {
    const int c = (wave + j * Geo::W) * 16;              // first column of the tile: x = c / R, r = c % R (+ li)
    ctile32<N, R>(lds + (c / R) * Geo::LX + (c % R) + li, lk, q);
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
  if (MUL && !C2_EARLY) {
#pragma unroll
    for (int k = 0; k < EPT4; ++k) {
      const bool rowok = !PART4 || tid + k * B < Geo::UNITS4;
      c2v[MUL ? k : 0] = ldrow(auxb + (rowok ? b0 + k * bstep : b0));
    }
  }
  // third sum of the fused BiCGSTAB iteration (krylov_kernels.hpp): <out, dot_with>.  Its stream was fetched with the other
  // side streams where all three fit the register budget (the 64-byte-row tiles: fetching it behind the contractions cost
  // the last pass 229 -> 298 us at GCY 20^6); on the 128-byte-row tiles it is read unit by unit here
  double dot_yv = 0.0, dot_yy = 0.0, dot_yr = 0.0;
#pragma unroll
  for (int k = 0; k < EPT4; ++k) {
    const int u = tid + k * B;
    if (!PART4 || u < Geo::UNITS4) {
      v4f y = *reinterpret_cast<const v4f*>(lds + 4 * u);
      if (MUL) {
        y = y * c2v[MUL ? k : 0];
        if (need_old) {
          const v4f ov = oldv[MUL ? k : 0];
          y = y - ov;
#pragma unroll
          for (int j = 0; j < 4; ++j) {                   // what is stored is what the dots refer to; sums in fp64
            dot_yv = fma((double)y[j], (double)ov[j], dot_yv);
            dot_yy = fma((double)y[j], (double)y[j], dot_yy);
          }
        }
        if (dot3) {
          const v4f rw = C2_EARLY ? rwv[(DOT3 && C2_EARLY) ? k : 0] : ldrow(rwb + (b0 + k * bstep));
#pragma unroll
          for (int j = 0; j < 4; ++j) dot_yr = fma((double)y[j], (double)rw[j], dot_yr);
        }
      }
      *reinterpret_cast<v4f*>(outb + (b0 + k * bstep)) = y;
    }
  }
  if (MUL && io.dotp != nullptr) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) { dot_yv += __shfl_xor(dot_yv, s); dot_yy += __shfl_xor(dot_yy, s); dot_yr += __shfl_xor(dot_yr, s); }
    if (lane == 0) { red4[wave] = dot_yv; red4[4 + wave] = dot_yy; red4[8 + wave] = dot_yr; }
    __syncthreads();
    if (tid == 0) {
      double a = 0.0, b = 0.0, c = 0.0;
      for (int w = 0; w < Geo::W; ++w) { a += red4[w]; b += red4[4 + w]; c += red4[8 + w]; }
      io.dotp[blockIdx.x] = a;
      io.dotp[gridDim.x + blockIdx.x] = b;
      if (dot3) io.dotp[2 * gridDim.x + blockIdx.x] = c;
    }
  }
}

#ifndef SDFS_NO_VARIANT_TABLES
inline slice_fn slice32_variant(int n) {
  switch (n) {
    case 16: return (slice_fn)slice32_kernel<16>;
    case 20: return (slice_fn)slice32_kernel<20>;
    case 24: return (slice_fn)slice32_kernel<24>;
    case 32: return (slice_fn)slice32_kernel<32>;
    default: return nullptr;
  }
}
// r: row length in floats (line32_row_floats)
template <int N, int R> inline line_fn line32_variant_nr(int mode, bool dot3) {
  if (mode == L_JLAST) return dot3 ? (line_fn)line32_kernel<N, L_JLAST, R, true> : (line_fn)line32_kernel<N, L_JLAST, R, false>;
  return mode == L_MID ? (line_fn)line32_kernel<N, L_MID, R> : nullptr;
}
template <int N> inline line_fn line32_variant_n(int mode, int r, bool dot3) {
  if constexpr (N <= 24) { if (r == 32) return line32_variant_nr<N, 32>(mode, dot3); }      // (32 x 32 x 32 floats would be 128 KB)
  return r == 16 ? line32_variant_nr<N, 16>(mode, dot3) : nullptr;
}
// dot3: the last pass with the third sum <out, LineIO::dot_with>
inline line_fn line32_variant(int n, int mode, int r, bool dot3 = false) {
  switch (n) {
    case 16: return line32_variant_n<16>(mode, r, dot3);
    case 20: return line32_variant_n<20>(mode, r, dot3);
    case 24: return line32_variant_n<24>(mode, r, dot3);
    case 32: return line32_variant_n<32>(mode, r, dot3);
    default: return nullptr;
  }
}
#endif
// row length of the fp32 line tiles of a pass: one whole 128-byte line where the remainder allows it
inline int line32_row_floats(int n, long long lrest) { return (n <= 24 && lrest % 32 == 0) ? 32 : 16; }

}  // namespace sdfs
