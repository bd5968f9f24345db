// krylov_kernels.hpp -- BiCGSTAB vector updates folded into the first pass of J.v (round 4).
//
// The inner solve of the Newton step (jax.scipy.sparse.linalg.bicgstab at code/solvers.py:91-93 on the matrix-free
// J.v of :87) is HBM-bound on large grids: at GCY 20^6 an iteration moves 34 grid streams -- two J.v applications of nine
// each and sixteen of fused BLAS-1 -- at 0.71 of the HBM peak (bench.py, secondary.gcy20_newton_1e-8.roofline).  What is
// left is the stream count.  Two of the BLAS-1 kernels produce exactly the vector the next J.v application starts from:
//     p = r + beta (p - omega q)     then  q = (J - I) p        (k_bicg_update_p, 4 streams + J.v's 3 in its first pass)
//     s = r - alpha q (in r), <s,s>  then  t = (J - I) s        (k_bicg_s, 3 streams + 3)
// slice_jfused_kernel does either update on the registers of J.v's first pass -- the slice pass over the two fastest axes
// (fast_kernels.hpp), x = c1 * v -- and writes the updated vector back beside the pass's own output: 6 and 5 streams
// instead of 7 and 6.  The third stream saved is <rhat, q>, summed by the last pass of the first J.v application (which
// holds q and reads p for its "- v" anyway: line_stream_kernel, LineIO::dot_with) instead of a kernel of its own:
// 31 streams per iteration instead of 34, the same arithmetic (the sums are added in another order).
// fp64 Krylov storage on the compile-time pair plan; the other plans and the fp32 forms keep the separate kernels.
#pragma once
#include "f32_kernels.hpp"
#include "stream_kernels.hpp"
#include "vec_kernels.hpp"

namespace sdfs {

enum { JF_P = 0, JF_S = 1 };

struct JFusedIO {
  double* upd;              // JF_P: p;  JF_S: r (becomes s) -- read, updated in place, and the vector J.v is applied to
  const double* a;          // JF_P: r
  const double* q;
  const double* c1;
  double* out;              // the pass's output (the intermediate of the J.v application)
  const double* sc;         // scalar block of the recurrence (vec_kernels.hpp): beta, omega / alpha
  double* dot;              // JF_S: per-wave-tile partial sums of <s, s>
  const unsigned long long* gate;
};

template <int N, int KIND>
__global__ void __launch_bounds__(256, 3)
slice_jfused_kernel(const SliceDesc P, const JFusedIO io) {
  using Geo = SliceGeo<N>;
  constexpr int BU = 2;                                            // units per batch (two batches in flight)
  constexpr int NB = (Geo::EPT + BU - 1) / BU;
  extern __shared__ double lds[];
  SDFS_GATED(io.gate);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long tile = (long long)blockIdx.x * Geo::WAVES + wave;
  const long long s0 = tile * Geo::G;
  if (s0 >= P.nslices) return;                                     // no workgroup barrier below
  const long long rem = (P.nslices - s0) * (N * N / 2);
  const int nvalid = rem < Geo::UNITS ? (int)rem : Geo::UNITS;
  double* const wl = lds + wave * Geo::LTILE;
  auto lofs = [](const int e) -> int { return Geo::RS == N ? e : (e / N) * Geo::RS + (e % N); };
  const long long gbase = s0 * (N * N);
  const unsigned lb = (unsigned)lane * 16u;
  char* const updb = reinterpret_cast<char*>(io.upd + gbase);
  const char* const ab = reinterpret_cast<const char*>(io.a + gbase);
  const char* const qb = reinterpret_cast<const char*>(io.q + gbase);
  const char* const cb = reinterpret_cast<const char*>(io.c1 + gbase);
  const double c_a = KIND == JF_P ? io.sc[SC_BETA] : io.sc[SC_ALPHA];
  const double c_b = KIND == JF_P ? io.sc[SC_OMEGA] : 0.0;
  struct Unit { v2d u, a, q, c; };
  auto load = [&](Unit (&B)[BU], const int b) {
#pragma unroll
    for (int j = 0; j < BU; ++j) {
      const int k = b * BU + j;
      // (a unit beyond the tile, or in a slice a trailing tile does not have, re-reads unit 0's piece: finite values
      // that are neither stored nor parked into a slice that is)
      const unsigned off = (k < Geo::EPT && lane + 64 * k < nvalid) ? lb + 1024u * k : lb;
      B[j].u = *reinterpret_cast<const v2d*>(updb + off);
      if (KIND == JF_P) B[j].a = ldg_stream(ab + off);
      B[j].q = *reinterpret_cast<const v2d*>(qb + off);
      B[j].c = *reinterpret_cast<const v2d*>(cb + off);
    }
  };
  double ss = 0.0;
  auto work = [&](const Unit (&B)[BU], const int b) {
#pragma unroll
    for (int j = 0; j < BU; ++j) {
      const int k = b * BU + j;
      if (k >= Geo::EPT) continue;
      const int u = lane + 64 * k;
      v2d nv;
      if (KIND == JF_P) {                                          // p = r + beta (p - omega q)          (k_bicg_update_p)
        nv.x = B[j].a.x + c_a * (B[j].u.x - c_b * B[j].q.x);
        nv.y = B[j].a.y + c_a * (B[j].u.y - c_b * B[j].q.y);
      } else {                                                     // s = r - alpha q, <s, s>             (k_bicg_s)
        nv.x = B[j].u.x - c_a * B[j].q.x;
        nv.y = B[j].u.y - c_a * B[j].q.y;
      }
      if (u < nvalid) {
        *reinterpret_cast<v2d*>(updb + (lb + 1024u * k)) = nv;
        if (KIND == JF_S) { ss += nv.x * nv.x; ss += nv.y * nv.y; }
      }
      if (Geo::UNITS % 64 == 0 || u < Geo::UNITS) *reinterpret_cast<v2d*>(wl + lofs(2 * u)) = (v2d){nv.x * B[j].c.x, nv.y * B[j].c.y};
    }
  };
  {
    Unit B0[BU], B1[BU];
    load(B0, 0);
#pragma unroll
    for (int b = 0; b < NB; b += 2) {
      if (b + 1 < NB) load(B1, b + 1);
      work(B0, b);
      if (b + 2 < NB) load(B0, b + 2);
      if (b + 1 < NB) work(B1, b + 1);
    }
  }
  if (KIND == JF_S) {
    ss = wave_sum(ss);
    if (lane == 0) io.dot[tile] = ss;
  }
  QFrag<N> qf;
  qf.load(P.Qf, lane);
  wave_lds_fence();
  const int li = lane & 15, lk = lane >> 4;
  {
    double* const p0 = wl + li * Geo::RS + lk;
#pragma unroll
    for (int ct = 0; ct < Geo::NCT; ++ct) ctile<N, 1>(p0 + ct * 16 * Geo::RS, qf);
  }
  wave_lds_fence();
  {
    QFrag<N> qe;
    qe.load(P.Qe, lane);
#pragma unroll
    for (int ct = 0; ct < Geo::NCT; ++ct) {
      const int c = 16 * ct + li;
      const int g = c / N, f = c - g * N;
      ctile<N, Geo::RS>(wl + g * (N * Geo::RS) + f + lk * Geo::RS, qe);
    }
  }
  wave_lds_fence();
  char* const outb = reinterpret_cast<char*>(io.out + gbase);
#pragma unroll
  for (int k = 0; k < Geo::EPT; ++k) {
    const int u = lane + 64 * k;
    if (u < nvalid) *reinterpret_cast<v2d*>(outb + (lb + 1024u * k)) = *reinterpret_cast<const v2d*>(wl + lofs(2 * u));
  }
}

// The same on the fp32-MFMA first pass (opts.krylov_f32 = 3, f32_kernels.hpp): Krylov vectors, c1 and the pass's output are
// floats; the update is formed in fp64 and rounded to the stored float, <s, s> is summed in fp64 from the rounded values --
// what k_bicg_update_p<float> / k_bicg_s<float> do -- and the product with c1 is the fp32 one of slice32_kernel.  Here the
// separate BLAS-1 kernels are 47 % of an iteration (they run at 4.4-4.5 TB/s, the fused passes at the first pass's 5.7).
struct JFused32IO {
  float* upd; const float* a; const float* q; const float* c1; float* out;
  const double* sc; double* dot; const unsigned long long* gate;
};

template <int N, int KIND>
__global__ void __launch_bounds__(256, 4)
slice32_jfused_kernel(const SliceDesc P, const JFused32IO io) {
  using Geo = Slice32Geo<N>;
  constexpr int BU = 2;                                            // float4 units per batch (two batches in flight)
  constexpr int NB = (Geo::EPT4 + BU - 1) / BU;
  extern __shared__ double lds_[];
  float* const lds = reinterpret_cast<float*>(lds_);
  SDFS_GATED(io.gate);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long tile = (long long)blockIdx.x * Geo::WAVES + wave;
  const long long s0 = tile * Geo::G;
  if (s0 >= P.nslices) return;                                     // no workgroup barrier below
  const long long rem = (P.nslices - s0) * (N * N / 4);
  const int nvalid4 = rem < Geo::UNITS4 ? (int)rem : Geo::UNITS4;
  float* const wl = lds + wave * Geo::LTILE;
  auto lofs = [](const int e) -> int { return (e / N) * Geo::RS + (e % N); };
  const long long gbase = s0 * (N * N);
  const unsigned lb = (unsigned)lane * 16u;
  char* const updb = reinterpret_cast<char*>(io.upd + gbase);
  const char* const ab = reinterpret_cast<const char*>(io.a + gbase);
  const char* const qb = reinterpret_cast<const char*>(io.q + gbase);
  const char* const cb = reinterpret_cast<const char*>(io.c1 + gbase);
  const double c_a = KIND == JF_P ? io.sc[SC_BETA] : io.sc[SC_ALPHA];
  const double c_b = KIND == JF_P ? io.sc[SC_OMEGA] : 0.0;
  struct Unit { v4f u, a, q, c; };
  auto load = [&](Unit (&B)[BU], const int b) {
#pragma unroll
    for (int j = 0; j < BU; ++j) {
      const int k = b * BU + j;
      const unsigned off = (k < Geo::EPT4 && lane + 64 * k < nvalid4) ? lb + 1024u * k : lb;      // (as slice32_kernel: unit 0 again)
      B[j].u = *reinterpret_cast<const v4f*>(updb + off);
      if (KIND == JF_P) B[j].a = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(ab + off));
      B[j].q = *reinterpret_cast<const v4f*>(qb + off);
      B[j].c = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(cb + off));
    }
  };
  double ss = 0.0;
  auto work = [&](const Unit (&B)[BU], const int b) {
#pragma unroll
    for (int j = 0; j < BU; ++j) {
      const int k = b * BU + j;
      if (k >= Geo::EPT4) continue;
      const int u = lane + 64 * k;
      v4f nv;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double d = KIND == JF_P ? (double)B[j].a[e] + c_a * ((double)B[j].u[e] - c_b * (double)B[j].q[e])     // p = r + beta (p - omega q)
                                      : (double)B[j].u[e] - c_a * (double)B[j].q[e];                               // s = r - alpha q
        nv[e] = (float)d;
      }
      if (u < nvalid4) {
        *reinterpret_cast<v4f*>(updb + (lb + 1024u * k)) = nv;
        if (KIND == JF_S) {
#pragma unroll
          for (int e = 0; e < 4; ++e) ss = fma((double)nv[e], (double)nv[e], ss);
        }
      }
      if (Geo::UNITS4 % 64 == 0 || u < Geo::UNITS4) {
        const v4f x = nv * B[j].c;
        float* const pl = wl + lofs(4 * u);
        *reinterpret_cast<v2f*>(pl) = (v2f){x.x, x.y};
        *reinterpret_cast<v2f*>(pl + 2) = (v2f){x.z, x.w};
      }
    }
  };
  {
    Unit B0[BU], B1[BU];
    load(B0, 0);
#pragma unroll
    for (int b = 0; b < NB; b += 2) {
      if (b + 1 < NB) load(B1, b + 1);
      work(B0, b);
      if (b + 2 < NB) load(B0, b + 2);
      if (b + 1 < NB) work(B1, b + 1);
    }
  }
  if (KIND == JF_S) {
    ss = wave_sum(ss);
    if (lane == 0) io.dot[tile] = ss;
  }
  QFrag32<N> qf;
  qf.load(P.Qf, lane);
  wave_lds_fence();
  const int li = lane & 15, lk = lane >> 4;
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
  char* const outb = reinterpret_cast<char*>(io.out + gbase);
#pragma unroll
  for (int k = 0; k < Geo::EPT4; ++k) {
    const int u = lane + 64 * k;
    if (u < nvalid4) {
      const float* const pl = wl + lofs(4 * u);
      const v2f a = *reinterpret_cast<const v2f*>(pl), b = *reinterpret_cast<const v2f*>(pl + 2);
      *reinterpret_cast<v4f*>(outb + (lb + 1024u * k)) = (v4f){a.x, a.y, b.x, b.y};
    }
  }
}

typedef void (*jfused32_fn)(const SliceDesc, const JFused32IO);
// ---------------------------------------------------------------------------------------------------
// line32_stream_mid_kernel: the middle pass of J.v on the fp32-MFMA tiles (f32_kernels.hpp, line32_kernel<N, L_MID, R>) as
// a persistent workgroup walking tiles with the next tile's loads in flight during the contractions -- the form that took
// the fp64 middle pass from 0.59 to 0.72 of the HBM peak (stream_kernels.hpp, line_stream_kernel<..., PERSIST>).  The
// registers a tile arrives in are free again once it is parked, so the next tile's loads are issued right there; a thread
// parks and stores the same LDS units, so no barrier separates a tile's stores from the next tile's park.
template <int N, int R, int WPC>
__global__ void __launch_bounds__(256, WPC)
line32_stream_mid_kernel(const LineDesc P, const LineIO io) {
  using Geo = Line32Geo<N, R>;
  constexpr int B = Geo::B, EPT4 = Geo::EPT4, Q4 = Geo::Q4;
  constexpr bool PART4 = Geo::UNITS4 % B != 0;
  extern __shared__ double lds_[];
  float* const lds = reinterpret_cast<float*>(lds_);
  __shared__ unsigned tk[2];
  SDFS_GATED(io.gate);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const unsigned cpo = (unsigned)P.nchunks * LINE_R / R;        // tiles per outer index (whole chunks of R floats)
  const long long ntiles = P.nouter * (long long)cpo;
  const unsigned b0 = ((unsigned)(tid / Q4) * (unsigned)P.lrest + 4u * (tid % Q4)) * 4u;
  const unsigned bstep = (unsigned)(B / Q4) * (unsigned)P.lrest * 4u;
  const TicketWalk W(ntiles, blockIdx.x, io.sched);
  unsigned cur, nxt;
  int par = 0;
  if (tid == 0) { const unsigned t0 = W.draw(); tk[0] = t0; tk[1] = t0 != NO_TILE ? W.draw() : NO_TILE; }
  __syncthreads();
  cur = tk[0]; nxt = tk[1];
  __syncthreads();
  QFrag32<N> qx, qy;
  qx.load(P.Qx, lane);
  qy.load(P.Qy, lane);
  auto tile_base = [&](const unsigned t) -> long long {
    const unsigned o = t / cpo;
    return (long long)o * (N * N) * P.lrest + (long long)(t - o * cpo) * R;
  };
  v4f v[EPT4];
  auto load_tile = [&](const unsigned t) {
    const char* const inb = reinterpret_cast<const char*>(reinterpret_cast<const float*>(io.in) + tile_base(t));
#pragma unroll
    for (int k = 0; k < EPT4; ++k) {
      const bool rowok = !PART4 || tid + k * B < Geo::UNITS4;
      const char* const q = inb + (rowok ? b0 + k * bstep : b0);
      v[k] = R == 32 ? __builtin_nontemporal_load(reinterpret_cast<const v4f*>(q)) : *reinterpret_cast<const v4f*>(q);      // (line32_kernel: nt on whole lines only)
    }
  };
  if (cur != NO_TILE) {
    load_tile(cur);
    for (;;) {
#pragma unroll
      for (int k = 0; k < EPT4; ++k) {
        const int u = tid + k * B;
        if (!PART4 || u < Geo::UNITS4) *reinterpret_cast<v4f*>(lds + 4 * u) = v[k];
      }
      const bool has_next = nxt != NO_TILE;                     // uniform over the workgroup
      const long long tbase = tile_base(cur);
      if (has_next) load_tile(nxt);
      unsigned nn = NO_TILE;                                     // (published before the third barrier: by then it has returned)
      if (tid == 0 && has_next) nn = W.draw();
      __syncthreads();
      constexpr int NCT = N * R / 16;
#pragma unroll
      for (int j = 0; j < NCT / Geo::W; ++j) { ctile32<N, Geo::LX>(lds + (wave + j * Geo::W) * 16 + li, lk, qx); __builtin_amdgcn_sched_barrier(0); }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < NCT / Geo::W; ++j) {
        const int c = (wave + j * Geo::W) * 16;
        ctile32<N, R>(lds + (c / R) * Geo::LX + (c % R) + li, lk, qy);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (tid == 0) tk[par] = nn;
      __syncthreads();
      char* const outb = reinterpret_cast<char*>(reinterpret_cast<float*>(io.out) + tbase);
#pragma unroll
      for (int k = 0; k < EPT4; ++k) {
        const int u = tid + k * B;
        if (!PART4 || u < Geo::UNITS4) *reinterpret_cast<v4f*>(outb + (b0 + k * bstep)) = *reinterpret_cast<const v4f*>(lds + 4 * u);
      }
      if (!has_next) break;
      cur = nxt;
      nxt = tk[par];
      par ^= 1;
    }
  }
  if (tid == 0) ticket_walk_done(io.sched, gridDim.x);
}

// workgroups per CU of the persistent fp32 middle pass (LDS: 51 KB tiles at 20 x 20 x 32 floats)
template <int N, int R> struct Line32Stream { static constexpr int WPC = Line32Geo<N, R>::BPC >= 3 ? 3 : (Line32Geo<N, R>::BPC < 1 ? 1 : Line32Geo<N, R>::BPC); };
inline int line32_stream_wpc(int n, int r) {
  switch (n) {
    case 16: return r == 32 ? Line32Stream<16, 32>::WPC : Line32Stream<16, 16>::WPC;
    case 20: return r == 32 ? Line32Stream<20, 32>::WPC : Line32Stream<20, 16>::WPC;
    case 24: return r == 32 ? Line32Stream<24, 32>::WPC : Line32Stream<24, 16>::WPC;
    default: return Line32Stream<32, 16>::WPC;
  }
}
#ifndef SDFS_NO_VARIANT_TABLES
template <int N> inline line_fn line32_stream_mid_variant_n(int r) {
  if constexpr (N <= 24) { if (r == 32) return (line_fn)line32_stream_mid_kernel<N, 32, Line32Stream<N, 32>::WPC>; }
  return r == 16 ? (line_fn)line32_stream_mid_kernel<N, 16, Line32Stream<N, 16>::WPC> : nullptr;
}
inline line_fn line32_stream_mid_variant(int n, int r) {
  switch (n) {
    case 16: return line32_stream_mid_variant_n<16>(r);
    case 20: return line32_stream_mid_variant_n<20>(r);
    case 24: return line32_stream_mid_variant_n<24>(r);
    case 32: return line32_stream_mid_variant_n<32>(r);
    default: return nullptr;
  }
}
#endif

typedef void (*jfused_fn)(const SliceDesc, const JFusedIO);
#ifndef SDFS_NO_VARIANT_TABLES
template <int N> inline jfused_fn slice_jfused_variant_n(int kind) {
  return kind == JF_P ? (jfused_fn)slice_jfused_kernel<N, JF_P> : (jfused_fn)slice_jfused_kernel<N, JF_S>;
}
template <int N> inline jfused32_fn slice32_jfused_variant_n(int kind) {
  return kind == JF_P ? (jfused32_fn)slice32_jfused_kernel<N, JF_P> : (jfused32_fn)slice32_jfused_kernel<N, JF_S>;
}
inline jfused32_fn slice32_jfused_variant(int n, int kind) {
  switch (n) {
    case 16: return slice32_jfused_variant_n<16>(kind);
    case 20: return slice32_jfused_variant_n<20>(kind);
    case 24: return slice32_jfused_variant_n<24>(kind);
    case 32: return slice32_jfused_variant_n<32>(kind);
    default: return nullptr;
  }
}
inline jfused_fn slice_jfused_variant(int n, int kind) {
  switch (n) {
    case 16: return slice_jfused_variant_n<16>(kind);
    case 20: return slice_jfused_variant_n<20>(kind);
    case 24: return slice_jfused_variant_n<24>(kind);
    case 32: return slice_jfused_variant_n<32>(kind);
    default: return nullptr;
  }
}
#endif

}  // namespace sdfs
