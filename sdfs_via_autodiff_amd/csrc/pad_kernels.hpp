// pad_kernels.hpp -- the pair plan for the grids between the plans: extents the compile-time pair kernels do not exist
// for (10^6 ... 15^6, 17^6 ... 19^6, 25^6, ragged shapes up to 32 per axis), too many points for the latency-tuned
// small-grid kernels.
//
// Same factorisation as fast_kernels.hpp (code/gcy/discrete/gcy_wc_ratio.py:134-238: H is a Kronecker product, so an
// application is three passes, one per adjacent axis pair), with RUN-TIME extents nx, ny <= NT on compile-time NT x NT
// MFMA tiles (NT = 16, 24, 32, chosen per pass): global addresses follow the real strides -- HBM traffic is the grid's,
// not the padded one's -- the LDS image is padded to NT x NT with zeros and the matrices are zero-padded NT x NT copies
// (padded rows and columns contribute exact zeros, ctile<NT, .> runs unmasked).
//   pad_slice_kernel  the two fastest axes: a tile is G = 4 / 2 / 1 consecutive slices of nx * ny contiguous doubles per wave,
//                     elements in memory order (8-byte requests, 512 contiguous bytes per wave request);
//   pad_line_kernel   a slower pair: a tile is all (x, y) rows of 16 consecutive positions behind the pair (one
//                     128-byte line per row when the remainder is a multiple of 16 doubles; rows of a ragged
//                     remainder start on any 8-byte boundary, so every request is one double).
// One tile per wave (slices) / workgroup (lines), no look-ahead: several workgroups per CU cover each other's phases.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fast_kernels.hpp"

// workgroups per CU of the slice form by tile width (tools/README.md: build-time probes)
#ifndef SDFS_PAD_SOCC16
#define SDFS_PAD_SOCC16 3
#endif
#ifndef SDFS_PAD_SOCC24
#define SDFS_PAD_SOCC24 3      // (round 4, 21^6: 558 -> 483 us at three, 732 at four -- spills)
#endif
#ifndef SDFS_PAD_SOCC32
#define SDFS_PAD_SOCC32 3      // (25^6: 2400 -> 2004 us at three, 2700 at four)
#endif
#ifndef SDFS_PAD_SOCC16_J
#define SDFS_PAD_SOCC16_J 3    // the J.v first pass (no power) on 16-wide tiles
#endif
// workgroups per CU of the line form where the tile would allow more than the register budget of the last pass does
#ifndef SDFS_PAD_LOCC_MID16
#define SDFS_PAD_LOCC_MID16 3
#endif
#ifndef SDFS_PAD_LOCC_MID24
#define SDFS_PAD_LOCC_MID24 3
#endif
#ifndef SDFS_PAD_LOCC_MID32
#define SDFS_PAD_LOCC_MID32 1
#endif
#ifndef SDFS_PAD_LOCC_LAST32
#define SDFS_PAD_LOCC_LAST32 1
#endif

namespace sdfs {

struct PadDesc {
  int nx, ny;               // extents of the contracted pair (X slower)
  unsigned mxy, my;         // ceil(2^20 / (nx ny)), ceil(2^20 / ny): e / d == (e * m) >> 20 for e < 4096 (and d >= 4)
  long long nslices;        // slice form: slices of nx * ny contiguous doubles
  long long lrest;          // line form: contiguous doubles behind Y
  int nchunks;              //            ceil(lrest / 16)
  long long nouter;         //            product of the extents before X
  long long ntiles;         //            nouter * nchunks
  const double* Qx;         // NT x NT zero-padded matrices
  const double* Qy;
  double theta, inv_theta, beta;
  const double* a3;         // aggregator scale: index = out_idx[o] + x * a3x + y * a3y + rest_idx[pos]
  const int* out_idx;
  const int* rest_idx;
  int a3x, a3y;
  int minus_identity;
};

// slice form: G slices per wave tile, LDS row stride NT + 2 (conflict-free columns, see SliceGeo)
template <int NT> struct PadSliceGeo {
  static constexpr int G = NT == 16 ? 4 : (NT == 20 ? 4 : (NT == 24 ? 2 : 1));      // (static LDS of the four wave tiles stays below 64 KB)
  static constexpr int RS = NT + 2;
  static constexpr int LT = G * NT * RS;            // doubles of LDS per wave tile
  static constexpr int EPL = 2 * ((G * NT * NT + 127) / 128);      // elements per lane of a full tile, rounded up to even (the power runs on pairs)
  static constexpr int OCC = NT == 16 ? SDFS_PAD_SOCC16 : (NT == 24 ? SDFS_PAD_SOCC24 : (NT == 32 ? SDFS_PAD_SOCC32 : 2));      // workgroups per CU the register budget is set for (20-wide: 56 KB of LDS per workgroup allow two)
  static_assert(EPL % 2 == 0, "the power runs on pairs");
  static_assert((G * NT) % 16 == 0, "whole column tiles");
};

// PAIR (round 4): 16-byte requests.  A wave tile is G * nx * ny contiguous doubles from an even element offset whenever
// that count is even (G = 4 / 2: always; G = 1: nx * ny even), so lane l takes the double2 units l, l + 64, ... of the
// tile -- half the load / store instructions of the element-wise form, which stays for odd 32-wide slices.  A unit may
// straddle a row or a slice: each of its two elements has its own LDS offset, as before.
template <int MODE, int NT, bool PAIR>
__global__ void __launch_bounds__(256, ((MODE == S_JFIRST && NT == 16) ? SDFS_PAD_SOCC16_J : PadSliceGeo<NT>::OCC))
pad_slice_kernel(const PadDesc P, const SliceIO io) {
  using Geo = PadSliceGeo<NT>;
  constexpr int PAD_G = Geo::G, PAD_RS = Geo::RS, PAD_LT = Geo::LT, PAD_EPL = Geo::EPL;
  constexpr bool POWP = MODE == S_TFIRST || MODE == S_TFIRST_LIN;
  constexpr bool LIN = MODE == S_TFIRST_LIN;
  constexpr bool MULP = MODE == S_JFIRST;
  __shared__ __attribute__((aligned(16))) double lds[4 * PAD_LT];
  if (io.gate != nullptr) {
    const unsigned long long g = *io.gate;
    if (g <= (unsigned long long)__double_as_longlong(io.gate_tol)) return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (io.zero != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *io.zero = 0ULL;
  const long long tile = (long long)blockIdx.x * 4 + wave;
  const long long s0 = tile * PAD_G;
  if (s0 >= P.nslices) return;                                   // no workgroup barrier below
  const int nxy = P.nx * P.ny;
  const long long left = P.nslices - s0;
  const int nval = (int)(left < PAD_G ? left : PAD_G) * nxy;     // elements of this tile (PAIR: even)
  const long long gbase = s0 * nxy;
  double* const wl = lds + wave * PAD_LT;
  // element k of this lane within the tile
  auto EI = [&](const int k) -> int { return PAIR ? 2 * (lane + 64 * (k >> 1)) + (k & 1) : lane + 64 * k; };
  // ---- loads, in memory order ------------------------------------------------------------------------------------
  double v[PAD_EPL], c1v[MULP ? PAD_EPL : 1];
  if (PAIR) {
#pragma unroll
    for (int k = 0; k < PAD_EPL; k += 2) {
      const int e = EI(k);
      const bool ok = e < nval;
      const v2d a = ok ? *reinterpret_cast<const v2d*>(io.in + gbase + e) : (v2d){POWP ? 1.0 : 0.0, POWP ? 1.0 : 0.0};
      v[k] = a.x; v[k + 1] = a.y;
      if (MULP) {
        const v2d c = ok ? *reinterpret_cast<const v2d*>(io.aux_in + gbase + e) : (v2d){0.0, 0.0};
        c1v[MULP ? k : 0] = c.x; c1v[MULP ? k + 1 : 0] = c.y;
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < PAD_EPL; ++k) {
      const int e = EI(k);
      v[k] = e < nval ? io.in[gbase + e] : (POWP ? 1.0 : 0.0);
      if (MULP) c1v[MULP ? k : 0] = e < nval ? io.aux_in[gbase + e] : 0.0;
    }
  }
  // ---- zero image, element -> (slice, x, y) -> LDS offset -------------------------------------------------------
#pragma unroll
  for (int k = 0; k < (PAD_LT + 63) / 64; ++k)
    if (PAD_LT % 64 == 0 || lane + 64 * k < PAD_LT) wl[lane + 64 * k] = 0.0;
  int lo[PAD_EPL];
#pragma unroll
  for (int k = 0; k < PAD_EPL; ++k) {
    const unsigned e = (unsigned)EI(k);
    const unsigned s = (e * P.mxy) >> 20, r = e - s * (unsigned)nxy;
    const unsigned x = (r * P.my) >> 20, y = r - x * (unsigned)P.ny;
    lo[k] = (int)((s * (unsigned)NT + x) * (unsigned)PAD_RS + y);
  }
  QFrag<NT> qf;
  qf.load(P.Qy, lane);
  wave_lds_fence();
  // ---- x = w^theta (c1 = w^(theta-1)) or v c1 on the registers, then parked -------------------------------------
  if (POWP) {
    PowK<true> PT;
    PT.init(P.theta, lane);
#pragma unroll
    for (int k = 0; k < PAD_EPL; k += 2) {
      // (uniform: no lane's pair lies inside the tile -- 25 x 25 on 32 x 32 fills ten of the 16 units: 25^6 2004 -> 1835 us.
      // Only there: on the 20- / 24-wide tiles the exits cost the unrolled routines their interleaving, 21^6 483 -> 918 us)
      if (NT == 32 && (PAIR ? 128 * (k >> 1) : 64 * k) >= nval) break;
      const double xin[2] = {v[k], v[k + 1]};                     // (masked lanes were loaded as 1)
      double xw[2];
      PT.template run<2>(xin, xw);
      if (LIN) {
        if (PAIR) {
          if (EI(k) < nval) *reinterpret_cast<v2d*>(io.aux_out + gbase + EI(k)) = (v2d){xw[0] / xin[0], xw[1] / xin[1]};
        } else {
          if (EI(k) < nval) io.aux_out[gbase + EI(k)] = xw[0] / xin[0];
          if (EI(k + 1) < nval) io.aux_out[gbase + EI(k + 1)] = xw[1] / xin[1];
        }
      }
      v[k] = xw[0]; v[k + 1] = xw[1];
    }
  }
#pragma unroll
  for (int k = 0; k < PAD_EPL; ++k) {
    if (MULP) v[k] *= c1v[MULP ? k : 0];
    if (EI(k) < nval) wl[lo[k]] = v[k];
  }
  wave_lds_fence();
  const int li = lane & 15, lk = lane >> 4;
  constexpr int NCT = PAD_G * NT / 16;                           // column tiles of either contraction
  // ---- contraction over the fastest axis: column c = (slice, x) at wl + c * RS, rows contiguous -----------------
  {
    double* const p0 = wl + li * PAD_RS + lk;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) ctile<NT, 1>(p0 + ct * 16 * PAD_RS, qf);
  }
  wave_lds_fence();
  // ---- contraction over the second axis: column c = (slice g, y) at wl + g NT RS + y, row stride RS -------------
  {
    QFrag<NT> qe;
    qe.load(P.Qx, lane);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const int c = 16 * ct + li;
      const int g = c / NT, f = c - g * NT;
      ctile<NT, PAD_RS>(wl + g * (NT * PAD_RS) + f + lk * PAD_RS, qe);
    }
  }
  wave_lds_fence();
  if (PAIR) {
#pragma unroll
    for (int k = 0; k < PAD_EPL; k += 2)
      if (EI(k) < nval) *reinterpret_cast<v2d*>(io.out + gbase + EI(k)) = (v2d){wl[lo[k]], wl[lo[k + 1]]};
  } else {
#pragma unroll
    for (int k = 0; k < PAD_EPL; ++k)
      if (EI(k) < nval) io.out[gbase + EI(k)] = wl[lo[k]];
  }
}

// Line form.  A tile is all (x, y) rows of R consecutive positions behind the pair (R = 16: a 128-byte line per row;
// R = 8, the wide tiles: 64-byte rows hold 4.7 of the 5.4 TB/s of whole lines as a memory skeleton -- tile_copy_probe,
// round 3 -- and halve the LDS of a tile: 37 / 64 KB at 24 / 32 wide).  Unit u = tid + B k is double2 c2 = tid % (R / 2)
// of row u / (R / 2) = (x, y) = (row / NT, row % NT); a thread keeps its two positions.  LDS: NT * NT rows of R doubles.
template <int NT, int R> struct PadLineGeo {
  static constexpr int B = NT <= 24 ? 256 : 512;
  static constexpr int W = B / 64;
  static constexpr int UPR = R / 2;                              // double2 units per row
  static constexpr int UNITS = NT * NT * UPR;
  static constexpr int EPT = (UNITS + B - 1) / B;
  static constexpr bool PARTIAL = UNITS % B != 0;
  static constexpr int LX = NT * R;
  static constexpr int NCT = NT * R / 16;                        // column tiles of either contraction
  static constexpr int OCC = B == 512 ? 1 : ((size_t)NT * NT * R * 8 <= 26 * 1024 ? 4 : ((size_t)NT * NT * R * 8 <= 40 * 1024 ? 3 : 2));      // (512 threads: 256 VGPRs, no spills in the last pass)
  static_assert(R == 8 || R == 16, "row lengths");
  static_assert((NT * R) % 16 == 0, "whole column tiles");
};

// EVEN (round 4): the remainder behind the pair is even, so every row of a tile starts on a 16-byte boundary and a unit is
// one 16-byte request in every stream (both of its positions are inside the remainder or neither is).
template <int MODE, int NT, int R> constexpr int pad_line_occ() {
  return (MODE == L_MID && NT == 16) ? SDFS_PAD_LOCC_MID16 : ((MODE == L_MID && NT == 24) ? SDFS_PAD_LOCC_MID24 :
         (NT == 32 ? (MODE == L_MID ? SDFS_PAD_LOCC_MID32 : SDFS_PAD_LOCC_LAST32) : PadLineGeo<NT, R>::OCC));
}
template <int MODE, int NT, int R, bool EVEN>
__global__ void __launch_bounds__((PadLineGeo<NT, R>::B), (pad_line_occ<MODE, NT, R>() * PadLineGeo<NT, R>::B / 256))
pad_line_kernel(const PadDesc P, const LineIO io) {
  using Geo = PadLineGeo<NT, R>;
  constexpr int UPR = Geo::UPR;
  constexpr bool CES = MODE == L_TLAST || MODE == L_TLAST_LIN;
  constexpr bool LINE = MODE == L_TLAST_LIN;
  constexpr bool MULE = MODE == L_JLAST;
  constexpr int EPT = Geo::EPT, B = Geo::B;
  extern __shared__ double lds[];
  __shared__ double red[32];
  if (io.gate != nullptr) {
    const unsigned long long g = *io.gate;
    if (g <= (unsigned long long)__double_as_longlong(io.gate_tol)) return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const unsigned t = (unsigned)xcd_remap((long long)blockIdx.x, P.ntiles);
  const unsigned o = t / (unsigned)P.nchunks;
  const int chunk = (int)(t - o * (unsigned)P.nchunks);
  const int c2 = tid & (UPR - 1);
  const long long pos = (long long)chunk * R + 2 * c2;
  const bool pk0 = pos < P.lrest, pk1 = pos + 1 < P.lrest;
  const long long tbase = (long long)o * (P.nx * P.ny) * P.lrest + (long long)chunk * R + 2 * c2;
  const bool need_old = CES ? io.resid != nullptr : (MULE && P.minus_identity);
  // unit k of this thread: row -> (x, y), validity, element offset against tbase
  auto unit_of = [&](const int k, int& x, int& y, bool& rok, long long& off) {
    const int row = (tid + B * k) / UPR;
    x = row / NT; y = row - x * NT;
    rok = x < P.nx && y < P.ny && (!Geo::PARTIAL || tid + B * k < Geo::UNITS);
    off = ((long long)x * P.ny + y) * P.lrest;
  };
  auto ld2 = [](const double* const q, const bool a, const bool b) -> double2 {
    if (EVEN) return a ? *reinterpret_cast<const double2*>(q) : make_double2(0.0, 0.0);
    return make_double2(a ? q[0] : 0.0, b ? q[1] : 0.0);
  };
  auto st2 = [](double* const q, const bool a, const bool b, const double x, const double y) {
    if (EVEN) { if (a) *reinterpret_cast<double2*>(q) = make_double2(x, y); }
    else { if (a) q[0] = x; if (b) q[1] = y; }
  };
  // ---- loads: the tile and the side stream that crosses the contractions -----------------------------------------
  const double* const inb = io.in + tbase;
  const double* const oldb = io.old + tbase;
  double2 s1[(CES || MULE) ? EPT : 1];
  {
    double2 v[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      int x, y; bool rok; long long off;
      unit_of(k, x, y, rok, off);
      v[k] = ld2(inb + off, pk0 && rok, pk1 && rok);
      if ((CES || MULE) && need_old) s1[(CES || MULE) ? k : 0] = ld2(oldb + off, pk0 && rok, pk1 && rok);
    }
#pragma unroll
    for (int k = 0; k < EPT; ++k)
      if (!Geo::PARTIAL || tid + B * k < Geo::UNITS) *reinterpret_cast<double2*>(lds + 2 * (tid + B * k)) = v[k];     // (every slot has its owner: no zero fill)
  }
  {
    QFrag<NT> q;
    q.load(P.Qx, lane);
    __syncthreads();
    // contraction over X: column = (y, r) = LDS offset y R + r, row stride NT R; column tile j = offsets 16 j ...
    {
      double* const p0 = lds + li + lk * Geo::LX;
#pragma unroll
      for (int j = 0; j < (Geo::NCT + Geo::W - 1) / Geo::W; ++j)
        if (Geo::NCT % Geo::W == 0 || wave + j * Geo::W < Geo::NCT) ctile<NT, Geo::LX>(p0 + (wave + j * Geo::W) * 16, q);
    }
    q.load(P.Qy, lane);
    __syncthreads();
    // contraction over Y: column = (x, r) at x LX + r, row stride R; a column tile is 16 / R values of x
    {
      constexpr int XPT = 16 / R;
      double* const p0 = lds + (li / R) * Geo::LX + (li % R) + lk * R;
#pragma unroll
      for (int j = 0; j < (Geo::NCT + Geo::W - 1) / Geo::W; ++j)
        if (Geo::NCT % Geo::W == 0 || wave + j * Geo::W < Geo::NCT) ctile<NT, R>(p0 + (wave + j * Geo::W) * XPT * Geo::LX, q);
    }
    __syncthreads();
  }
  // ---- epilogue --------------------------------------------------------------------------------------------------
  double* const outb = io.out + tbase;
  double rmax = 0.0, dot_yv = 0.0, dot_yy = 0.0;
  bool rnan = false;
  if (CES) {
    PowK<false> PT;
    PT.init(P.inv_theta, lane);
    const unsigned io0 = (unsigned)P.out_idx[o];
    const unsigned ia = pk0 ? io0 + (unsigned)P.rest_idx[pos] : 0u;
    const unsigned ib = pk1 ? io0 + (unsigned)P.rest_idx[pos + 1] : 0u;
    double2 s2[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      int x, y; bool rok; long long off;
      unit_of(k, x, y, rok, off);
      const unsigned ixy = (unsigned)x * (unsigned)P.a3x + (unsigned)y * (unsigned)P.a3y;
      s2[k] = make_double2(P.a3[(pk0 && rok) ? ia + ixy : 0u], P.a3[(pk1 && rok) ? ib + ixy : 0u]);
    }
    double* const auxo = io.aux_out + tbase;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      int x, y; bool rok; long long off;
      unit_of(k, x, y, rok, off);
      const bool a = pk0 && rok, b = pk1 && rok;
      const double2 sv = *reinterpret_cast<const double2*>(lds + 2 * ((!Geo::PARTIAL || tid + B * k < Geo::UNITS) ? tid + B * k : tid));
      // Tw = 1 + beta (a3 S)^(1/theta), c2 = beta u / S, |Tw - w|; every lane runs the power (its table gathers need the
      // whole wave), masked lanes feed it 1
      const double ks[2] = {a ? s2[k].x * sv.x : 1.0, b ? s2[k].y * sv.y : 1.0};
      double uu[2];
      PT.template run<2>(ks, uu);
      const double y0 = 1.0 + P.beta * uu[0], y1 = 1.0 + P.beta * uu[1];
      if (a && need_old) { const double r0 = fabs(y0 - s1[k].x); rnan |= (r0 != r0); rmax = fmax(rmax, r0); }
      if (b && need_old) { const double r1 = fabs(y1 - s1[k].y); rnan |= (r1 != r1); rmax = fmax(rmax, r1); }
      if (LINE) st2(auxo + off, a, b, P.beta * uu[0] / sv.x, P.beta * uu[1] / sv.y);
      st2(outb + off, a, b, y0, y1);
    }
  } else {
    const double* const auxb = io.aux_in + tbase;
    double2 s2[MULE ? EPT : 1];
    if (MULE) {
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        int x, y; bool rok; long long off;
        unit_of(k, x, y, rok, off);
        s2[MULE ? k : 0] = ld2(auxb + off, pk0 && rok, pk1 && rok);
      }
    }
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      int x, y; bool rok; long long off;
      unit_of(k, x, y, rok, off);
      const bool a = pk0 && rok, b = pk1 && rok;
      double2 y2 = *reinterpret_cast<const double2*>(lds + 2 * ((!Geo::PARTIAL || tid + B * k < Geo::UNITS) ? tid + B * k : tid));
      if (MULE) {
        y2.x *= s2[MULE ? k : 0].x; y2.y *= s2[MULE ? k : 0].y;
        if (P.minus_identity) {
          const double2 o2 = s1[MULE ? k : 0];
          y2.x -= o2.x; y2.y -= o2.y;
          if (a) { dot_yv = fma(y2.x, o2.x, dot_yv); dot_yy = fma(y2.x, y2.x, dot_yy); }
          if (b) { dot_yv = fma(y2.y, o2.y, dot_yv); dot_yy = fma(y2.y, y2.y, dot_yy); }
        }
      }
      st2(outb + off, a, b, y2.x, y2.y);
    }
  }
  // ---- per-workgroup reductions ------------------------------------------------------------------------------------
  if (MULE && io.dotp != nullptr) {
    dot_yv = wave_sum_f64(dot_yv); dot_yy = wave_sum_f64(dot_yy);
    if (lane == 0) { red[wave] = dot_yv; red[16 + wave] = dot_yy; }
    __syncthreads();
    if (tid == 0) {
      double sa = 0.0, sb = 0.0;
      for (int w = 0; w < Geo::W; ++w) { sa += red[w]; sb += red[16 + w]; }
      io.dotp[blockIdx.x] = sa;
      io.dotp[gridDim.x + blockIdx.x] = sb;
    }
  }
  if (CES && io.resid != nullptr) {
    if (rnan) rmax = __longlong_as_double(0x7ff0000000000000LL);                // NaN -> +inf
    rmax = wave_max_f64(rmax);
    if (lane == 0) red[wave] = rmax;
    __syncthreads();
    if (tid == 0) {
      double r = red[0];
      for (int w = 1; w < Geo::W; ++w) r = fmax(r, red[w]);
      atomicMax(io.resid, (unsigned long long)__double_as_longlong(r));
    }
  }
}

typedef void (*pad_slice_fn)(const PadDesc, const SliceIO);
typedef void (*pad_line_fn)(const PadDesc, const LineIO);
#ifndef SDFS_NO_VARIANT_TABLES
template <int NT, bool PAIR> inline pad_slice_fn pad_slice_variant_n(int mode) {
  switch (mode) {
    case S_TFIRST: return pad_slice_kernel<S_TFIRST, NT, PAIR>;
    case S_TFIRST_LIN: return pad_slice_kernel<S_TFIRST_LIN, NT, PAIR>;
    case S_JFIRST: return pad_slice_kernel<S_JFIRST, NT, PAIR>;
    default: return nullptr;
  }
}
template <int NT, int R, bool EVEN> inline pad_line_fn pad_line_variant_ne(int mode) {
  switch (mode) {
    case L_MID: return pad_line_kernel<L_MID, NT, R, EVEN>;
    case L_TLAST: return pad_line_kernel<L_TLAST, NT, R, EVEN>;
    case L_TLAST_LIN: return pad_line_kernel<L_TLAST_LIN, NT, R, EVEN>;
    case L_JLAST: return pad_line_kernel<L_JLAST, NT, R, EVEN>;
    default: return nullptr;
  }
}
template <int NT, int R> inline pad_line_fn pad_line_variant_n(int mode, bool even) {
  return even ? pad_line_variant_ne<NT, R, true>(mode) : pad_line_variant_ne<NT, R, false>(mode);
}
// (nxy = nx * ny of the slice pass: the 32-wide tiles hold one slice and pair their requests when it is even)
inline pad_slice_fn pad_slice_variant(int nt, int mode, int nxy) {
  return nt == 16 ? pad_slice_variant_n<16, true>(mode) : (nt == 20 ? pad_slice_variant_n<20, true>(mode) : (nt == 24 ? pad_slice_variant_n<24, true>(mode) :
         (nt == 32 ? (nxy % 2 == 0 ? pad_slice_variant_n<32, true>(mode) : pad_slice_variant_n<32, false>(mode)) : nullptr)));
}
// (rows of 16 doubles on the 16-wide tiles, of 8 on the wide ones)
// (even: the remainder behind the pair is even -- 16-byte requests)
inline pad_line_fn pad_line_variant(int nt, int mode, bool even) {
  return nt == 16 ? pad_line_variant_n<16, 16>(mode, even) : (nt == 20 ? pad_line_variant_n<20, 8>(mode, even) : (nt == 24 ? pad_line_variant_n<24, 8>(mode, even) :
         (nt == 32 ? pad_line_variant_n<32, 8>(mode, even) : nullptr)));
}
#endif
inline int pad_slice_g(int nt) { return nt == 16 ? PadSliceGeo<16>::G : (nt == 20 ? PadSliceGeo<20>::G : (nt == 24 ? PadSliceGeo<24>::G : PadSliceGeo<32>::G)); }
inline int pad_line_r(int nt) { return nt == 16 ? 16 : 8; }
inline int pad_line_block(int nt) { return nt <= 24 ? 256 : 512; }
inline size_t pad_line_lds(int nt) { return (size_t)nt * nt * pad_line_r(nt) * 8; }

}  // namespace sdfs
