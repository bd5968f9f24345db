// pad_kernels.hpp -- the pair plan for 6-D grids between the plans: every extent <= 16 (10^6 ... 15^6, ragged shapes),
// too many points for the latency-tuned small-grid kernels, no extent the compile-time pair kernels exist for.
//
// Same factorisation as fast_kernels.hpp (code/gcy/discrete/gcy_wc_ratio.py:134-238: H is a Kronecker product, so an
// application is three passes, one per adjacent axis pair), with RUN-TIME extents nx, ny <= 16 on compile-time 16 x 16
// MFMA tiles: global addresses follow the real strides -- HBM traffic is the grid's, not the padded one's -- the LDS
// image is padded to 16 x 16 with zeros and the matrices are the zero-padded 16 x 16 copies the small-grid plan keeps
// (padded rows and columns contribute exact zeros, ctile<16, .> runs unmasked).
//   pad_slice_kernel  the two fastest axes: a tile is G = 4 consecutive slices of nx * ny contiguous doubles per wave,
//                     elements in memory order (8-byte requests, 512 contiguous bytes per wave request);
//   pad_line_kernel   a slower pair: a tile is all (x, y) rows of 16 consecutive positions behind the pair (one
//                     128-byte line per row when the remainder is a multiple of 16 doubles; rows of a ragged
//                     remainder start on any 8-byte boundary, so every request is one double).
// One tile per wave (slices) / workgroup (lines), no look-ahead: several workgroups per CU cover each other's phases.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fast_kernels.hpp"

namespace sdfs {

struct PadDesc {
  int nx, ny;               // extents of the contracted pair (X slower)
  unsigned mxy, my;         // ceil(2^20 / (nx ny)), ceil(2^20 / ny): e / d == (e * m) >> 20 for e < 4096
  long long nslices;        // slice form: slices of nx * ny contiguous doubles
  long long lrest;          // line form: contiguous doubles behind Y
  int nchunks;              //            ceil(lrest / 16)
  long long nouter;         //            product of the extents before X
  long long ntiles;         //            nouter * nchunks
  const double* Qx;         // 16 x 16 zero-padded matrices
  const double* Qy;
  double theta, inv_theta, beta;
  const double* a3;         // aggregator scale: index = out_idx[o] + x * a3x + y * a3y + rest_idx[pos]
  const int* out_idx;
  const int* rest_idx;
  int a3x, a3y;
  int minus_identity;
};

constexpr int PAD_G = 4;                        // slices per wave tile
constexpr int PAD_RS = 18;                      // LDS row stride (conflict-free columns, see SliceGeo)
constexpr int PAD_LT = PAD_G * 16 * PAD_RS;     // doubles of LDS per wave tile
constexpr int PAD_EPL = PAD_G * 256 / 64;       // elements per lane of a full tile

template <int MODE>
__global__ void __launch_bounds__(256, 3)
pad_slice_kernel(const PadDesc P, const SliceIO io) {
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
  const int nval = (int)(left < PAD_G ? left : PAD_G) * nxy;     // elements of this tile
  const long long gbase = s0 * nxy;
  double* const wl = lds + wave * PAD_LT;
  // ---- loads, in memory order ------------------------------------------------------------------------------------
  double v[PAD_EPL], c1v[MULP ? PAD_EPL : 1];
#pragma unroll
  for (int k = 0; k < PAD_EPL; ++k) {
    const int e = lane + 64 * k;
    v[k] = e < nval ? io.in[gbase + e] : (POWP ? 1.0 : 0.0);
    if (MULP) c1v[MULP ? k : 0] = e < nval ? io.aux_in[gbase + e] : 0.0;
  }
  // ---- zero image, element -> (slice, x, y) -> LDS offset -------------------------------------------------------
#pragma unroll
  for (int k = 0; k < PAD_LT / 64; ++k) wl[lane + 64 * k] = 0.0;
  int lo[PAD_EPL];
#pragma unroll
  for (int k = 0; k < PAD_EPL; ++k) {
    const unsigned e = (unsigned)(lane + 64 * k);
    const unsigned s = (e * P.mxy) >> 20, r = e - s * (unsigned)nxy;
    const unsigned x = (r * P.my) >> 20, y = r - x * (unsigned)P.ny;
    lo[k] = (int)((s * 16u + x) * (unsigned)PAD_RS + y);
  }
  QFrag<16> qf;
  qf.load(P.Qy, lane);
  wave_lds_fence();
  // ---- x = w^theta (c1 = w^(theta-1)) or v c1 on the registers, then parked -------------------------------------
  if (POWP) {
    const PowLane PT = pow_lane_init(lane);
#pragma unroll
    for (int k = 0; k < PAD_EPL; k += 2) {
      const double xin[2] = {v[k], v[k + 1]};                     // (masked lanes were loaded as 1)
      double xw[2];
      pow_fast_n<true, 2>(xin, P.theta, PT, xw);
      if (LIN) {
        if (lane + 64 * k < nval) io.aux_out[gbase + lane + 64 * k] = xw[0] / xin[0];
        if (lane + 64 * (k + 1) < nval) io.aux_out[gbase + lane + 64 * (k + 1)] = xw[1] / xin[1];
      }
      v[k] = xw[0]; v[k + 1] = xw[1];
    }
  }
#pragma unroll
  for (int k = 0; k < PAD_EPL; ++k) {
    if (MULP) v[k] *= c1v[MULP ? k : 0];
    if (lane + 64 * k < nval) wl[lo[k]] = v[k];
  }
  wave_lds_fence();
  const int li = lane & 15, lk = lane >> 4;
  // ---- contraction over the fastest axis: column c = (slice, x) at wl + c * RS, rows contiguous -----------------
  {
    double* const p0 = wl + li * PAD_RS + lk;
#pragma unroll
    for (int ct = 0; ct < PAD_G; ++ct) ctile<16, 1>(p0 + ct * 16 * PAD_RS, qf);
  }
  wave_lds_fence();
  // ---- contraction over the second axis: column c = (slice g, y) at wl + g 16 RS + y, row stride RS -------------
  {
    QFrag<16> qe;
    qe.load(P.Qx, lane);
#pragma unroll
    for (int g = 0; g < PAD_G; ++g) ctile<16, PAD_RS>(wl + g * (16 * PAD_RS) + li + lk * PAD_RS, qe);
  }
  wave_lds_fence();
#pragma unroll
  for (int k = 0; k < PAD_EPL; ++k)
    if (lane + 64 * k < nval) io.out[gbase + lane + 64 * k] = wl[lo[k]];
}

// Line form.  Unit u = tid + 256 k (k < 8) is the double2 c2 = tid & 7 of row u >> 3 = (x, y) = ((tid >> 7) + 2 k,
// (tid >> 3) & 15): a thread keeps its y and its two positions, x advances by 2 per unit.
template <int MODE>
__global__ void __launch_bounds__(256, 3)
pad_line_kernel(const PadDesc P, const LineIO io) {
  constexpr bool CES = MODE == L_TLAST || MODE == L_TLAST_LIN;
  constexpr bool LINE = MODE == L_TLAST_LIN;
  constexpr bool MULE = MODE == L_JLAST;
  constexpr int EPT = 8;
  __shared__ __attribute__((aligned(16))) double lds[16 * 16 * LINE_R];
  __shared__ double red[16];
  if (io.gate != nullptr) {
    const unsigned long long g = *io.gate;
    if (g <= (unsigned long long)__double_as_longlong(io.gate_tol)) return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const unsigned t = (unsigned)xcd_remap((long long)blockIdx.x, P.ntiles);
  const unsigned o = t / (unsigned)P.nchunks;
  const int chunk = (int)(t - o * (unsigned)P.nchunks);
  const int c2 = tid & 7, y = (tid >> 3) & 15, x0 = tid >> 7;
  const long long pos = (long long)chunk * LINE_R + 2 * c2;
  const bool yok = y < P.ny;
  const bool ok0 = yok && pos < P.lrest, ok1 = yok && pos + 1 < P.lrest;
  const long long tbase = (long long)o * (P.nx * P.ny) * P.lrest + (long long)chunk * LINE_R;
  const long long off0 = ((long long)x0 * P.ny + y) * P.lrest + 2 * c2;       // element offset of unit 0 against tbase
  const long long ostep = 2LL * P.ny * P.lrest;
  const bool need_old = CES ? io.resid != nullptr : (MULE && P.minus_identity);
  // ---- loads: the tile and the side stream that crosses the contractions -----------------------------------------
  const double* const inb = io.in + tbase + off0;
  const double* const oldb = io.old + tbase + off0;
  double2 v[EPT], s1[(CES || MULE) ? EPT : 1];
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const bool xok = x0 + 2 * k < P.nx;
    v[k] = make_double2((ok0 && xok) ? inb[k * ostep] : 0.0, (ok1 && xok) ? inb[k * ostep + 1] : 0.0);
    if ((CES || MULE) && need_old)
      s1[(CES || MULE) ? k : 0] = make_double2((ok0 && xok) ? oldb[k * ostep] : 0.0, (ok1 && xok) ? oldb[k * ostep + 1] : 0.0);
  }
#pragma unroll
  for (int k = 0; k < EPT; ++k) *reinterpret_cast<double2*>(lds + 2 * (tid + 256 * k)) = v[k];     // (every slot has its owner: no zero fill)
  {
    QFrag<16> q;
    q.load(P.Qx, lane);
    __syncthreads();
    // contraction over X: column = (y, r) = LDS offset, row stride 16 * 16
    {
      double* const p0 = lds + li + lk * (16 * LINE_R);
#pragma unroll
      for (int j = 0; j < 4; ++j) ctile<16, 16 * LINE_R>(p0 + (wave + j * 4) * 16, q);
    }
    q.load(P.Qy, lane);
    __syncthreads();
    // contraction over Y: column = (x, r) at x * 256 + r, row stride 16
    {
      double* const p0 = lds + li + lk * LINE_R;
#pragma unroll
      for (int j = 0; j < 4; ++j) ctile<16, LINE_R>(p0 + (wave + j * 4) * (16 * LINE_R), q);
    }
    __syncthreads();
  }
  // ---- epilogue --------------------------------------------------------------------------------------------------
  double* const outb = io.out + tbase + off0;
  double rmax = 0.0, dot_yv = 0.0, dot_yy = 0.0;
  bool rnan = false;
  if (CES) {
    const PowLane PT = pow_lane_init(lane);
    const unsigned io0 = (unsigned)P.out_idx[o];
    const unsigned ia = ok0 ? io0 + (unsigned)P.rest_idx[pos] + (unsigned)y * (unsigned)P.a3y : 0u;
    const unsigned ib = ok1 ? io0 + (unsigned)P.rest_idx[pos + 1] + (unsigned)y * (unsigned)P.a3y : 0u;
    double2 s2[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const bool xok = x0 + 2 * k < P.nx;
      const unsigned ix = (unsigned)(x0 + 2 * k) * (unsigned)P.a3x;
      s2[k] = make_double2(P.a3[(ok0 && xok) ? ia + ix : 0u], P.a3[(ok1 && xok) ? ib + ix : 0u]);
    }
    double* const auxo = io.aux_out + tbase + off0;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const bool xok = x0 + 2 * k < P.nx;
      const bool a = ok0 && xok, b = ok1 && xok;
      const double2 sv = *reinterpret_cast<const double2*>(lds + 2 * (tid + 256 * k));
      // Tw = 1 + beta (a3 S)^(1/theta), c2 = beta u / S, |Tw - w|; every lane runs the power (its table gathers need the
      // whole wave), masked lanes feed it 1
      const double ks[2] = {a ? s2[k].x * sv.x : 1.0, b ? s2[k].y * sv.y : 1.0};
      double uu[2];
      pow_fast_n<false, 2>(ks, P.inv_theta, PT, uu);
      const double y0 = 1.0 + P.beta * uu[0], y1 = 1.0 + P.beta * uu[1];
      if (a) {
        if (LINE) auxo[k * ostep] = P.beta * uu[0] / sv.x;
        if (need_old) { const double r0 = fabs(y0 - s1[k].x); rnan |= (r0 != r0); rmax = fmax(rmax, r0); }
        outb[k * ostep] = y0;
      }
      if (b) {
        if (LINE) auxo[k * ostep + 1] = P.beta * uu[1] / sv.y;
        if (need_old) { const double r1 = fabs(y1 - s1[k].y); rnan |= (r1 != r1); rmax = fmax(rmax, r1); }
        outb[k * ostep + 1] = y1;
      }
    }
  } else {
    const double* const auxb = io.aux_in + tbase + off0;
    double2 s2[MULE ? EPT : 1];
    if (MULE) {
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const bool xok = x0 + 2 * k < P.nx;
        s2[MULE ? k : 0] = make_double2((ok0 && xok) ? auxb[k * ostep] : 0.0, (ok1 && xok) ? auxb[k * ostep + 1] : 0.0);
      }
    }
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const bool xok = x0 + 2 * k < P.nx;
      const bool a = ok0 && xok, b = ok1 && xok;
      double2 y2 = *reinterpret_cast<const double2*>(lds + 2 * (tid + 256 * k));
      if (MULE) {
        y2.x *= s2[MULE ? k : 0].x; y2.y *= s2[MULE ? k : 0].y;
        if (P.minus_identity) {
          const double2 o2 = s1[MULE ? k : 0];
          y2.x -= o2.x; y2.y -= o2.y;
          if (a) { dot_yv = fma(y2.x, o2.x, dot_yv); dot_yy = fma(y2.x, y2.x, dot_yy); }
          if (b) { dot_yv = fma(y2.y, o2.y, dot_yv); dot_yy = fma(y2.y, y2.y, dot_yy); }
        }
      }
      if (a) outb[k * ostep] = y2.x;
      if (b) outb[k * ostep + 1] = y2.y;
    }
  }
  // ---- per-workgroup reductions ------------------------------------------------------------------------------------
  if (MULE && io.dotp != nullptr) {
    dot_yv = wave_sum_f64(dot_yv); dot_yy = wave_sum_f64(dot_yy);
    if (lane == 0) { red[wave] = dot_yv; red[8 + wave] = dot_yy; }
    __syncthreads();
    if (tid == 0) {
      io.dotp[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
      io.dotp[gridDim.x + blockIdx.x] = (red[8] + red[9]) + (red[10] + red[11]);
    }
  }
  if (CES && io.resid != nullptr) {
    if (rnan) rmax = __longlong_as_double(0x7ff0000000000000LL);                // NaN -> +inf
    rmax = wave_max_f64(rmax);
    if (lane == 0) red[wave] = rmax;
    __syncthreads();
    if (tid == 0) atomicMax(io.resid, (unsigned long long)__double_as_longlong(fmax(fmax(red[0], red[1]), fmax(red[2], red[3]))));
  }
}

typedef void (*pad_slice_fn)(const PadDesc, const SliceIO);
typedef void (*pad_line_fn)(const PadDesc, const LineIO);
#ifndef SDFS_NO_VARIANT_TABLES
inline pad_slice_fn pad_slice_variant(int mode) {
  switch (mode) {
    case S_TFIRST: return pad_slice_kernel<S_TFIRST>;
    case S_TFIRST_LIN: return pad_slice_kernel<S_TFIRST_LIN>;
    case S_JFIRST: return pad_slice_kernel<S_JFIRST>;
    default: return nullptr;
  }
}
inline pad_line_fn pad_line_variant(int mode) {
  switch (mode) {
    case L_MID: return pad_line_kernel<L_MID>;
    case L_TLAST: return pad_line_kernel<L_TLAST>;
    case L_TLAST_LIN: return pad_line_kernel<L_TLAST_LIN>;
    case L_JLAST: return pad_line_kernel<L_JLAST>;
    default: return nullptr;
  }
}
#endif

}  // namespace sdfs
