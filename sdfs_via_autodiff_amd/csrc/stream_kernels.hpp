// stream_kernels.hpp -- forms of the pair plan's line kernel (fast_kernels.hpp) that keep global loads in flight
// while a tile is in its fp64 phases.
//
// Why: the one-tile-per-workgroup kernels issue a tile's loads, wait, compute, store -- nothing of theirs is in
// flight during the powers and the MFMAs, and the memory skeleton of the same tile walks loses 15-25 % of its rate
// as soon as a compute-sized gap sits between its loads and its stores (tools/probes/tile_copy_probe.hip, `sleep`
// rows of profiles/round2_probes.txt: 5.6 -> 4.4-4.6 TB/s); a load takes 5-7 us once the memory system is loaded.
//   PERSIST  a workgroup walks tiles drawn from ticket counters; right after parking tile t in LDS it issues every
//            load of tile t' and only waits for them when t is stored.  The Q fragments of both contractions stay in
//            registers for the whole launch (loads return in order on gfx950: a Q load issued behind the prefetch
//            would wait for all of it).  Pays for the middle pass (two contractions between load and store).
//   OLDPF    the last pass's HBM side stream (w for the residual) is loaded for the whole tile before the
//            contractions instead of a window of four units ahead of the power.
// Measured one kernel at a time by tools/probes/kernel_bench.hip (profiles/round3_kernel_bench.txt).  The same
// treatment of the slice kernel (one wave walking wave tiles, next tile in registers) lost 5 %: that pass is bound
// by the issue of its power routine, and the 52 extra live registers cost more than the prefetch gains.
#pragma once
#include "fast_kernels.hpp"

#ifndef SDFS_STREAM_STAMP      // tools/probes/kernel_bench.hip defines it: s_memtime stamps of one workgroup's phases
#define SDFS_STREAM_STAMP(i)
#define SDFS_STREAM_STAMP_NEXT
#define SDFS_STREAM_STAMP_DECL
#endif

#ifndef SDFS_LAST_SB           // units of the last pass's epilogue between two scheduling barriers (build-time probe)
#define SDFS_LAST_SB 1
#endif

namespace sdfs {


// Tile scheduler of a persistent launch: tickets, one counter per XCD.  The tiles are cut into eight contiguous
// ranges; the workgroups (slice form: waves) of XCD x (= blockIdx.x & 7 under round-robin dispatch; only locality
// depends on that) draw the tiles of range x in order, so tiles that are neighbours in memory are in flight at
// about the same time (as under dispatch order; with a static stride per workgroup the walk drifts apart and
// 128-byte rows lose their DRAM-page neighbours: 0.234 against 0.213 ms for the middle pass at 64 KB row spacing)
// and the launch ends within one tile time on every CU.  The last workgroup out clears the scheduler words, so no
// memset sits between launches.
constexpr unsigned NO_TILE = 0xffffffffu;
constexpr int TK_SUB = 4;                      // ticket words per XCD
constexpr int TK_STRIDE = 16;                  // unsigneds between two words: a 64-byte line each
constexpr int TK_WORDS = (8 * TK_SUB + 1) * TK_STRIDE;      // scheduler buffer of a launch (the last line: workgroups done)
struct TicketWalk {
  unsigned start, count;
  unsigned* ctr;
  // A returning atomic on one word retires every ~110 ns when hundreds of waves draw from it (measured: 40000
  // tickets over 8 words took 0.55 ms), so every XCD's range is cut once more into TK_SUB sub-ranges with a word
  // each, and a ticket can stand for `batch` consecutive tiles.
  __device__ __forceinline__ TicketWalk(long long nunits, unsigned block, unsigned* sched) {
    const unsigned q = (unsigned)(nunits >> 3), r = (unsigned)(nunits & 7);
    const unsigned x = block & 7u, sub = (block >> 3) & (TK_SUB - 1);
    const unsigned xs = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    const unsigned xc = q + (x < r ? 1u : 0u);
    const unsigned q2 = xc / TK_SUB, r2 = xc % TK_SUB;
    start = xs + ((sub < r2) ? sub * (q2 + 1) : r2 * (q2 + 1) + (sub - r2) * q2);
    count = q2 + (sub < r2 ? 1u : 0u);
    ctr = sched + (x * TK_SUB + sub) * TK_STRIDE;
  }
  // one lane calls; NO_TILE once the sub-range is drained
  __device__ __forceinline__ unsigned draw() const {
    const unsigned i = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return i < count ? start + i : NO_TILE;
  }
};
// called by one lane per workgroup (slice form: per wave) after its last draw has returned
__device__ __forceinline__ void ticket_walk_done(unsigned* sched, unsigned narrivals) {
  const unsigned d = __hip_atomic_fetch_add(&sched[8 * TK_SUB * TK_STRIDE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (d == narrivals - 1) {
    for (int i = 0; i <= 8 * TK_SUB; ++i) __hip_atomic_store(&sched[i * TK_STRIDE], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// tile t of a line pass -> (outer index, chunk) and the element offset of its first row
__device__ __forceinline__ long long line_tile_base(const LineDesc& P, const long long t, const int nn, unsigned& o, int& chunk) {
  o = (unsigned)t / (unsigned)P.nchunks;
  chunk = (int)((unsigned)t - o * (unsigned)P.nchunks);
  return (long long)o * nn * P.lrest + (long long)chunk * LINE_R;
}
// all loads of one line tile (whole chunks): unit u = tid + k B at byte offset b0 + k bstep of the tile's base
template <int EPT, int B, int UNITS>
__device__ __forceinline__ void line_tile_load(v2d (&v)[EPT], const double* base, const int tid, const unsigned b0, const unsigned bstep) {
  const char* const inb = reinterpret_cast<const char*>(base);
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const bool rowok = UNITS % B == 0 || tid + k * B < UNITS;
    v[k] = ldg_stream(inb + (rowok ? b0 + k * bstep : b0));
  }
}

// ---------------------------------------------------------------------------------------------------
// slice_walk_kernel: slice_kernel's plain first pass of T (S_TFIRST) and its plain contraction (S_MID) with a wave
// WALKING wave tiles (round 4).  What bounds slice_kernel at GCY 20^6 is neither its instruction count (a power routine
// 16 % shorter bought 2.7 %), nor occupancy (16 waves per CU instead of 12: nothing), nor its LDS bank conflicts
// (padded rows: nothing) -- profiles/round4_kernel_bench.txt -- but the bytes it keeps in flight: a wave has loads
// outstanding only while it waits for its own tile (~45 % of its time), so a CU holds ~70 KB in flight against the
// ~135 KB that 5.4 TB/s at 6 us per load need.  Here the registers a tile arrives in are free again as soon as the tile
// is parked in LDS (the power runs on them first), so the NEXT tile's 13 loads are issued right there and travel while
// the wave runs its two contractions and stores the result -- no second register set (round 3's form issued them before
// the power, next to the current tile: 168 VGPRs, 2 spills, slower).  Both Q fragment sets stay in registers for the
// launch (loads return in order: a fragment load behind the prefetch would wait for all of it).  Tiles are drawn from
// the per-XCD ticket counters, one ticket per wave and tile, the ticket of the tile after next during the power.
// OCC = waves per SIMD the register budget is set for; QRELOAD: the Q fragments are fetched per tile, in front of the
// prefetch (their L2 round trip is exposed once per tile, but 40 registers are free during the power).
// POWLDS: the tile is parked raw, the next tile's loads are issued at once (they travel under the power as well) and the
// power runs in place in LDS, unit by unit in a rolled loop (one copy of the routine; write + read + write per unit instead
// of one write, but no unit waits in registers while the routine runs).
// POW1 (with POWLDS): one element at a time through the routine (half the temporaries; three waves per SIMD interleave).
template <int N, int MODE, int OCC = 3, bool QRELOAD = false, bool POWLDS = false, bool POW1 = false>
__global__ void __launch_bounds__(256, OCC)
slice_walk_kernel(const SliceDesc P, const SliceIO io) {
  using Geo = SliceGeo<N>;
  static_assert(MODE == S_TFIRST || MODE == S_MID, "plain first pass of T / plain contraction");
  constexpr bool POWP = MODE == S_TFIRST;
  extern __shared__ double lds[];
  if (io.gate != nullptr) {
    const unsigned long long g = *io.gate;
    if (g <= (unsigned long long)__double_as_longlong(io.gate_tol)) return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (io.zero != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *io.zero = 0ULL;
  const long long ntiles = P.nslices / Geo::G;
  const TicketWalk W(ntiles, blockIdx.x, io.sched);
  double* const wl = lds + wave * Geo::LTILE;
  auto lofs = [](const int e) -> int { return Geo::RS == N ? e : (e / N) * Geo::RS + (e % N); };
  const unsigned lb = (unsigned)lane * 16u;
  const int li = lane & 15, lk = lane >> 4;
  auto draw = [&]() -> unsigned {
    unsigned t = 0u;
    if (lane == 0) t = W.draw();
    return (unsigned)__builtin_amdgcn_readfirstlane((int)t);
  };
  // whole wave tiles only (the host launches this form when nslices is a multiple of G): every offset is an immediate
  auto load_tile = [&](v2d (&v)[Geo::EPT], const unsigned t) {
    const char* const inb = reinterpret_cast<const char*>(io.in + (long long)t * Geo::TILE);
#pragma unroll
    for (int k = 0; k < Geo::EPT; ++k) {
      const bool in_tile = Geo::UNITS % 64 == 0 || lane + 64 * k < Geo::UNITS;      // (compile time except in the last unit)
      v[k] = ldg_stream(inb + (in_tile ? lb + 1024u * k : lb));    // (a lane past the tile re-reads unit 0's piece: positive, finite)
    }
  };
  unsigned cur = draw();
  unsigned nxt = cur != NO_TILE ? draw() : NO_TILE;
  if (cur != NO_TILE) {
    QFrag<N> qf, qe;
    if (!QRELOAD) { qf.load(P.Qf, lane); qe.load(P.Qe, lane); }
    PowK<true> PT;
    if (POWP) PT.init(P.theta, lane);
    v2d v[Geo::EPT];
    load_tile(v, cur);
    for (;;) {
      unsigned nn = NO_TILE;
      if (nxt != NO_TILE) nn = draw();               // the tile after next (returns under the power)
      // ---- power on the registers, park ------------------------------------------------------------------------
#pragma unroll
      for (int k = 0; k < Geo::EPT; ++k) {
        const int u = lane + 64 * k;
        if (POWP && !POWLDS) {
          const double xin[2] = {v[k].x, v[k].y};
          double xw[2];
          PT.template run<2>(xin, xw);
          v[k] = (v2d){xw[0], xw[1]};
        }
        if (Geo::UNITS % 64 == 0 || u < Geo::UNITS) *reinterpret_cast<v2d*>(wl + lofs(2 * u)) = v[k];
      }
      char* const outb = reinterpret_cast<char*>(io.out + (long long)cur * Geo::TILE);
      // ---- the next tile's loads: the registers are free, the requests travel under the contractions ----------------
      if (QRELOAD) { qf.load(P.Qf, lane); qe.load(P.Qe, lane); }
      if (nxt != NO_TILE) load_tile(v, nxt);
      if (POWP && POWLDS) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        for (int k = 0; k < Geo::EPT; ++k) {
          const int u = lane + 64 * k;
          const bool in_tile = Geo::UNITS % 64 == 0 || u < Geo::UNITS;
          const int lo = in_tile ? lofs(2 * u) : 0;
          const v2d x2 = *reinterpret_cast<const v2d*>(wl + lo);
          const double xin[2] = {in_tile ? x2.x : 1.0, in_tile ? x2.y : 1.0};
          double xw[2];
          if (POW1) {
            const double xa[1] = {xin[0]}, xb[1] = {xin[1]};
            double ya[1], yb[1];
            PT.template run<1>(xa, ya);
            __builtin_amdgcn_sched_barrier(0);
            PT.template run<1>(xb, yb);
            xw[0] = ya[0]; xw[1] = yb[0];
          } else {
            PT.template run<2>(xin, xw);
          }
          if (in_tile) *reinterpret_cast<v2d*>(wl + lo) = (v2d){xw[0], xw[1]};
        }
      }
      wave_lds_fence();
      {
        double* const p0 = wl + li * Geo::RS + lk;
#pragma unroll
        for (int ct = 0; ct < Geo::NCT; ++ct) { ctile<N, 1>(p0 + ct * 16 * Geo::RS, qf); __builtin_amdgcn_sched_barrier(0); }
      }
      wave_lds_fence();
      {
#pragma unroll
        for (int ct = 0; ct < Geo::NCT; ++ct) {
          const int c = 16 * ct + li;
          const int g = c / N, f = c - g * N;
          ctile<N, Geo::RS>(wl + g * (N * Geo::RS) + f + lk * Geo::RS, qe);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      wave_lds_fence();
#pragma unroll
      for (int k = 0; k < Geo::EPT; ++k) {
        const int u = lane + 64 * k;
        if (Geo::UNITS % 64 == 0 || u < Geo::UNITS) stg_stream<NT_SLICE>(outb + (lb + 1024u * k), *reinterpret_cast<const v2d*>(wl + lofs(2 * u)));
        if (k % 4 == 3) __builtin_amdgcn_sched_barrier(0);      // four units on their way out at a time (register budget)
      }
      wave_lds_fence();
      if (nxt == NO_TILE) break;
      cur = nxt; nxt = nn;
    }
  }
  if (lane == 0) ticket_walk_done(io.sched, gridDim.x * Geo::WAVES);
}

// ---------------------------------------------------------------------------------------------------
// line_stream_kernel: line_kernel's forms on whole chunks (lrest % 16 == 0), one workgroup walking tiles.
// WPC = workgroups per CU the register budget is set for (3: 168 VGPRs, 2: 256).
// OLDPF: the epilogue's HBM side stream (T: w for the residual; J.v: v) is loaded for the WHOLE tile before the
// contractions instead of a window of four units ahead of the power -- a window is a few hundred cycles deep, an HBM
// load under load takes microseconds (phase stamps: the epilogue took 10 us per tile against 3.3 us of issue).  The
// epilogue is then fully unrolled (static register indices).  (Issuing the next tile's loads unit by unit as the
// epilogue retires side-stream registers, to save the second register set, did not work: the loads of a loaded
// memory system take 5-7 us, the park at the top of the next tile waited 7 us for them.)
// B = threads per workgroup (a multiple of 64; the column tiles of a contraction go round the B / 64 waves).
// PERSIST = false: one tile per workgroup (grid = ntiles, XCD-contiguous order), no next-tile prefetch, one Q fragment
// set at a time -- the OLDPF side-stream load on its own, at a register budget that admits three workgroups per CU.
// A3F (one tile per workgroup): the aggregator's scale from the two small tables of LineDesc::f1 / f2 -- 20 doubles per
// tile through LDS and one 16-byte piece per thread -- instead of two gathers and their index arithmetic per unit.
// IN32 / OUT32 (round 4, opts.t_f32): the tile arrives as / leaves as scaled floats -- 16-byte units of four, row u >> 2,
// float4 u & 3 of its 16 positions -- and is converted at the park / at the store; everything between (fp64 LDS tile,
// contractions, epilogue) is the fp64 form's.  IN32 in the last pass of T multiplies by 2^-k (t32_scale_of) at the park.
template <int N, int MODE, int WPC, bool OLDPF = false, int B = LineGeo<N>::B, bool PERSIST = true, bool A3F = false, bool IN32 = false,
          bool OUT32 = false>
__global__ void __launch_bounds__(B, WPC * B / 256)
line_stream_kernel(const LineDesc P, const LineIO io) {
  using Geo = LineGeo<N>;
  constexpr int NW = B / 64;
  constexpr int EPT = (Geo::UNITS + B - 1) / B;
  constexpr bool CES = MODE == L_TLAST || MODE == L_TLAST_LIN;
  constexpr bool LINE = MODE == L_TLAST_LIN;
  constexpr bool MULE = MODE == L_JLAST;
  constexpr bool PARTIAL = Geo::UNITS % B != 0;
  static_assert(MODE != L_TFUSED, "the fused end + start form stays with line_kernel");
  static_assert(!OUT32 || MODE == L_MID, "fp32 output: the middle pass");
  static_assert(!IN32 || MODE == L_MID || MODE == L_TLAST, "fp32 input: the passes of a plain T application");
  constexpr int EPT4 = Geo::EPT4;
  constexpr bool PART4 = Geo::UNITS4 % B != 0;
  static_assert(!A3F || CES, "two-table a3: the last pass of T");
  extern __shared__ double lds[];
  __shared__ double red[16];
  __shared__ double sF1[A3F ? 64 : 1];       // two halves: a persistent workgroup fills the next tile's while the last one is read
  if (io.gate != nullptr) {
    const unsigned long long g = *io.gate;
    if (g <= (unsigned long long)__double_as_longlong(io.gate_tol)) return;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const int c2 = tid & 7;
  const unsigned b0 = ((unsigned)(tid >> 3) * (unsigned)P.lrest + 2u * c2) * 8u;
  const unsigned bstep = (unsigned)(B / 8) * (unsigned)P.lrest * 8u;
  const bool andp = CES && !LINE && io.and_r != nullptr;     // uniform: Anderson's push rides on this pass
  const bool need_old = CES ? (io.resid != nullptr || andp) : (MULE && P.minus_identity);
  const char* const a3b = reinterpret_cast<const char*>(P.a3);
  const unsigned a3x = (unsigned)P.a3x, a3y = (unsigned)P.a3y;
  const TicketWalk W(P.ntiles, blockIdx.x, io.sched);
  __shared__ unsigned tk[3];           // tk[0], tk[1]: the first two tiles; then tk[par] = the ticket drawn during the tile before last
  unsigned cur, nxt = NO_TILE;
  int par = 0;
  if (PERSIST) {
    if (tid == 0) { const unsigned t0 = W.draw(); tk[0] = t0; tk[1] = t0 != NO_TILE ? W.draw() : NO_TILE; }
    __syncthreads();
    cur = tk[0]; nxt = tk[1];
    __syncthreads();
  } else {
    cur = (unsigned)xcd_remap((long long)blockIdx.x, P.ntiles);
  }
  QFrag<N> qx, qy_;
  qx.load(P.Qx, lane);
  if (PERSIST) qy_.load(P.Qy, lane);
  QFrag<N>& qy = PERSIST ? qy_ : qx;
  PowK<false> PT;
  if (CES) PT.init(P.inv_theta, lane);
  SDFS_STREAM_STAMP_DECL;
  double rmax = 0.0, dot_yv = 0.0, dot_yy = 0.0, dot_yr = 0.0;
  bool rnan = false;
  const bool dot3 = MULE && io.dot_with != nullptr;             // uniform

  v2d v[IN32 ? 1 : EPT];
  float4 vf[IN32 ? EPT4 : 1];
  // float units: element offset of unit 0 against the tile base, and between two units of a thread
  const unsigned e0 = (unsigned)(tid >> 2) * (unsigned)P.lrest + 4u * (tid & 3);
  const unsigned estep = (unsigned)(B / 4) * (unsigned)P.lrest;
  auto load32 = [&](const long long base) {
    const char* const inb = reinterpret_cast<const char*>(reinterpret_cast<const float*>(io.in) + base);
#pragma unroll
    for (int k = 0; k < EPT4; ++k) {
      const bool rowok = !PART4 || tid + k * B < Geo::UNITS4;
      vf[IN32 ? k : 0] = *reinterpret_cast<const float4*>(inb + (rowok ? e0 + k * estep : e0) * 4u);
    }
  };
  double unscale32 = 1.0;
  if (IN32 && CES) { const PowLane PL0 = pow_lane_init(lane); unscale32 = t32_scale_of(P.t32_ref > 0.0 ? P.t32_ref : io.old[P.ref_off], P.theta, PL0, true); }
  v2d wv[OLDPF ? EPT : 1];
  if (cur != NO_TILE) {
    {
      unsigned o0; int ch0;
      const long long base0 = line_tile_base(P, cur, N * N, o0, ch0);
      if constexpr (IN32) load32(base0);
      else line_tile_load<EPT, B, Geo::UNITS>(v, io.in + base0, tid, b0, bstep);
    }
    for (;;) {
      SDFS_STREAM_STAMP(0);
      if constexpr (IN32) {
#pragma unroll
        for (int k = 0; k < EPT4; ++k) {
          const int u = tid + k * B;
          if (!PART4 || u < Geo::UNITS4) {
            const float4 f = vf[IN32 ? k : 0];
            *reinterpret_cast<v2d*>(lds + 4 * u) = (v2d){(double)f.x * unscale32, (double)f.y * unscale32};
            *reinterpret_cast<v2d*>(lds + 4 * u + 2) = (v2d){(double)f.z * unscale32, (double)f.w * unscale32};
          }
        }
      } else {
#pragma unroll
        for (int k = 0; k < EPT; ++k)
          if (!PARTIAL || tid + k * B < Geo::UNITS) *reinterpret_cast<v2d*>(lds + 2 * (tid + k * B)) = v[k];
      }
      const bool has_next = PERSIST && nxt != NO_TILE;            // uniform over the workgroup
      unsigned o; int chunk;
      const long long tbase = line_tile_base(P, cur, N * N, o, chunk);
      v2d f2v = (v2d){1.0, 1.0};
      if constexpr (A3F) {
        if (tid < N) sF1[32 * par + tid] = P.f1[(long long)o * N + tid];      // (read behind this tile's barriers)
        f2v = *reinterpret_cast<const v2d*>(P.f2 + (long long)o * P.lrest + (long long)chunk * LINE_R + 2 * c2);
      }
      if constexpr (OLDPF) { if (need_old) line_tile_load<EPT, B, Geo::UNITS>(wv, io.old + tbase, tid, b0, bstep); }
      unsigned o1; int ch1;
      const long long nbase = line_tile_base(P, has_next ? nxt : cur, N * N, o1, ch1);
      if (has_next) {
        if constexpr (IN32) load32(nbase);
        else line_tile_load<EPT, B, Geo::UNITS>(v, io.in + nbase, tid, b0, bstep);
      }
      unsigned nn = NO_TILE;                                         // (published before the third barrier: by then it has returned)
      if (PERSIST && tid == 0 && has_next) nn = W.draw();
      SDFS_STREAM_STAMP(1);
      __syncthreads();
      SDFS_STREAM_STAMP(2);
      {
        double* const p0 = lds + li + lk * Geo::LX;
#pragma unroll
        for (int j = 0; j < (N + NW - 1) / NW; ++j) { if (N % NW == 0 || wave + j * NW < N) ctile<N, Geo::LX>(p0 + (wave + j * NW) * 16, qx); __builtin_amdgcn_sched_barrier(0); }
      }
      SDFS_STREAM_STAMP(3);
      if (!PERSIST) qx.load(P.Qy, lane);
      __syncthreads();
      SDFS_STREAM_STAMP(4);
      {
        double* const p0 = lds + li + lk * LINE_R;
#pragma unroll
        for (int j = 0; j < (N + NW - 1) / NW; ++j) { if (N % NW == 0 || wave + j * NW < N) ctile<N, LINE_R>(p0 + (wave + j * NW) * Geo::LX, qy); __builtin_amdgcn_sched_barrier(0); }
      }
      SDFS_STREAM_STAMP(5);
      if (PERSIST && tid == 0) tk[par] = nn;
      __syncthreads();
      SDFS_STREAM_STAMP(6);
      char* const outb = reinterpret_cast<char*>(io.out + tbase);
      if constexpr (OUT32) {
        char* const outb4 = reinterpret_cast<char*>(reinterpret_cast<float*>(io.out) + tbase);
#pragma unroll
        for (int k = 0; k < EPT4; ++k) {
          const int u = tid + k * B;
          if (!PART4 || u < Geo::UNITS4) {
            const v2d a = *reinterpret_cast<const v2d*>(lds + 4 * u), b = *reinterpret_cast<const v2d*>(lds + 4 * u + 2);
            *reinterpret_cast<float4*>(outb4 + (e0 + k * estep) * 4u) = make_float4((float)a.x, (float)a.y, (float)b.x, (float)b.y);
          }
        }
      } else if (!CES && !MULE) {
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
          const int u = tid + k * B;
          if (!PARTIAL || u < Geo::UNITS) stg_stream<NT_MID>(outb + (b0 + (unsigned)k * bstep), *reinterpret_cast<const v2d*>(lds + 2 * u));
          if (k % 4 == 3) __builtin_amdgcn_sched_barrier(0);      // four units on their way out at a time (register budget)
        }
      } else {
        // epilogue, unit by unit; side streams a window of LOOK units ahead (they queue behind the prefetch, which
        // has had both contractions to land)
        constexpr int LOOK = 4;
        const char* const oldb = reinterpret_cast<const char*>(io.old + tbase);
        const char* const auxb = reinterpret_cast<const char*>(io.aux_in + tbase);
        char* const auxo = reinterpret_cast<char*>(io.aux_out + tbase);
        unsigned ia3a = 0u, ia3b = 0u;
        if (CES && !A3F) {
          const long long pos = (long long)chunk * LINE_R + 2 * c2;
          ia3a = (unsigned)(P.out_idx[o] + P.rest_idx[pos]);
          ia3b = (unsigned)(P.out_idx[o] + P.rest_idx[pos + 1]);
        }
        const char* const dwb = reinterpret_cast<const char*>(io.dot_with + tbase);
        double2 sw[LOOK], cw[LOOK], rw[LOOK];
        auto issue = [&](const int k, double2& s1, double2& s2, double2& s3) {
          const int u = tid + k * B;
          const bool rowok = k < EPT && (!PARTIAL || u < Geo::UNITS);
          const unsigned offc = rowok ? b0 + (unsigned)k * bstep : b0;
          if (!OLDPF && need_old) s1 = ldg_stream2(oldb + offc);
          if (MULE && dot3) s3 = ldg_stream2(dwb + offc);
          if (CES && A3F) {
            const int row = rowok ? (u >> 3) : 0;
            const double fx = sF1[32 * par + row / N];
            s2 = make_double2(fx * f2v.x, fx * f2v.y);
          } else if (CES) {
            const int row = rowok ? (u >> 3) : 0;
            const int x = row / N, y = row - x * N;
            const unsigned ixy = __umul24((unsigned)x, a3x) + __umul24((unsigned)y, a3y);
            s2 = make_double2(*reinterpret_cast<const double*>(a3b + (ia3a + ixy) * 8u),
                              *reinterpret_cast<const double*>(a3b + (ia3b + ixy) * 8u));
          } else {
            s2 = *reinterpret_cast<const double2*>(auxb + offc);
          }
        };
        auto unit = [&](const int k, const double2 s1, const double2 s2, const double2 s3) {
          const int u = tid + k * B;
          const bool rowok = !PARTIAL || u < Geo::UNITS;
          const unsigned off = b0 + (unsigned)k * bstep;
          const double2 sv = *reinterpret_cast<const double2*>(lds + 2 * (rowok ? u : tid));
          if (CES) {
            const double ks[2] = {s2.x * sv.x, s2.y * sv.y};
            double uu[2];
            PT.template run<2>(ks, uu);
            const double2 y2 = make_double2(1.0 + P.beta * uu[0], 1.0 + P.beta * uu[1]);
            if (rowok) {
              if (LINE) *reinterpret_cast<double2*>(auxo + off) = make_double2(P.beta * uu[0] / sv.x, P.beta * uu[1] / sv.y);
              if (need_old) {
                const double r0 = fabs(y2.x - s1.x), r1 = fabs(y2.y - s1.y);
                rnan |= (r0 != r0) | (r1 != r1);
                rmax = fmax(rmax, fmax(r0, r1));
              }
              if (andp) {                                          // k_and_push_lite on the registers of this unit
                const double2 rr2 = make_double2(y2.x - s1.x, y2.y - s1.y);
                *reinterpret_cast<double2*>(reinterpret_cast<char*>(io.and_r + tbase) + off) = rr2;
                *reinterpret_cast<double2*>(reinterpret_cast<char*>(io.and_y + tbase) + off) =
                    make_double2(fma(io.and_beta, rr2.x, s1.x), fma(io.and_beta, rr2.y, s1.y));
                dot_yy = fma(rr2.x, rr2.x, dot_yy); dot_yy = fma(rr2.y, rr2.y, dot_yy);
              }
              // (GCY 16^6, 134 MB per grid: the next application's first pass 0.0620 -> 0.0562 ms when T w was stored cacheably,
              // this pass 0.088 -> 0.089 ms; at 512 MB per grid nothing survives and the non-temporal store wins: same-box A/B)
              if (P.cached_out) *reinterpret_cast<double2*>(outb + off) = y2;
              else stg_stream2<NT_LAST>(outb + off, y2);
            }
          } else if (rowok) {
            double2 y2 = make_double2(sv.x * s2.x, sv.y * s2.y);
            if (P.minus_identity) {
              y2.x -= s1.x; y2.y -= s1.y;
              dot_yv = fma(y2.x, s1.x, dot_yv); dot_yv = fma(y2.y, s1.y, dot_yv);
              dot_yy = fma(y2.x, y2.x, dot_yy); dot_yy = fma(y2.y, y2.y, dot_yy);
            }
            if (dot3) { dot_yr = fma(y2.x, s3.x, dot_yr); dot_yr = fma(y2.y, s3.y, dot_yr); }
            *reinterpret_cast<double2*>(outb + off) = y2;
          }
        };
#pragma unroll
        for (int j = 0; j < LOOK; ++j) issue(j, sw[j], cw[j], rw[j]);
        if (OLDPF) {
#pragma unroll
          for (int k = 0; k < EPT; ++k) {
            const v2d wk = wv[OLDPF ? k : 0];
            unit(k, make_double2(wk.x, wk.y), cw[k % LOOK], rw[k % LOOK]);
            if (k + LOOK < EPT) issue(k + LOOK, sw[k % LOOK], cw[k % LOOK], rw[k % LOOK]);
            if (k % SDFS_LAST_SB == SDFS_LAST_SB - 1) __builtin_amdgcn_sched_barrier(0);      // (units the scheduler may interleave)
          }
        } else {
          int kk = 0;
#pragma unroll 1
          for (; kk + LOOK <= EPT; kk += LOOK) {
#pragma unroll
            for (int j = 0; j < LOOK; ++j) {
              unit(kk + j, sw[j], cw[j], rw[j]);
              issue(kk + j + LOOK, sw[j], cw[j], rw[j]);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
#pragma unroll
          for (int j = 0; j < EPT % LOOK; ++j) unit(EPT - EPT % LOOK + j, sw[j], cw[j], rw[j]);
        }
      }
      SDFS_STREAM_STAMP(7);
      SDFS_STREAM_STAMP_NEXT;
      if (!has_next) break;
      cur = nxt;
      nxt = tk[par];
      par ^= 1;
    }
  }
  if (PERSIST && tid == 0) ticket_walk_done(io.sched, gridDim.x);
  if (CES && !LINE) {
    if (andp) {                                                    // (uniform; `red` is free until the residual's reduction below)
#pragma unroll
      for (int s = 32; s > 0; s >>= 1) dot_yy += __shfl_xor(dot_yy, s);
      if (lane == 0) red[8 + wave] = dot_yy;
      __syncthreads();
      if (tid == 0) {
        double a = 0.0;
        for (int w = 0; w < NW; ++w) a += red[8 + w];
        io.and_dot[blockIdx.x] = a;
      }
    }
  }
  if (MULE && io.dotp != nullptr) {
    __shared__ double red3[8];
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) { dot_yv += __shfl_xor(dot_yv, s); dot_yy += __shfl_xor(dot_yy, s); dot_yr += __shfl_xor(dot_yr, s); }
    if (lane == 0) { red[wave] = dot_yv; red[8 + wave] = dot_yy; red3[wave] = dot_yr; }
    __syncthreads();
    if (tid == 0) {
      double a = 0.0, b = 0.0, c = 0.0;
      for (int w = 0; w < NW; ++w) { a += red[w]; b += red[8 + w]; c += red3[w]; }
      io.dotp[blockIdx.x] = a;
      io.dotp[gridDim.x + blockIdx.x] = b;
      if (dot3) io.dotp[2 * gridDim.x + blockIdx.x] = c;
    }
  }
  if (CES && io.resid != nullptr) {
    if (rnan) rmax = __longlong_as_double(0x7ff0000000000000LL);
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) rmax = fmax(rmax, __shfl_xor(rmax, s));
    if (lane == 0) red[wave] = rmax;
    __syncthreads();
    if (tid == 0) {
      double r = red[0];
      for (int w = 1; w < NW; ++w) r = fmax(r, red[w]);
      atomicMax(io.resid, (unsigned long long)__double_as_longlong(r));
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// line_tlast32_kernel: the last pass of T on an fp32 intermediate (opts.t_f32, BASELINE config 5; first pass:
// slice_kernel<N, S_TFIRST32>, middle pass: line_kernel's fp32 L_MID form).  The tile arrives as scaled floats in
// 16-byte units of four (row u >> 2, float4 u & 3 of its 16 elements), is contracted in fp64, multiplied by 2^-k
// (t32_scale_of) and goes through the aggregator; w (for the residual) and T w are fp64 streams, two 16-byte
// pieces per unit.  One tile per workgroup, the residual's w loaded for the whole tile before the contractions.
template <int N, int WPC>
__global__ void __launch_bounds__(LineGeo<N>::B, WPC * LineGeo<N>::B / 256)
line_tlast32_kernel(const LineDesc P, const LineIO io) {
  using Geo = LineGeo<N>;
  constexpr int B = Geo::B;
  constexpr int NW = B / 64;
  constexpr int EPT4 = Geo::EPT4;
  constexpr bool PART4 = Geo::UNITS4 % B != 0;
  extern __shared__ double lds[];
  __shared__ double red[16];
  if (io.gate != nullptr) {
    const unsigned long long g = *io.gate;
    if (g <= (unsigned long long)__double_as_longlong(io.gate_tol)) return;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const unsigned t = (unsigned)xcd_remap((long long)blockIdx.x, P.ntiles);
  unsigned o; int chunk;
  const long long tbase = line_tile_base(P, t, N * N, o, chunk);
  const unsigned e0 = (unsigned)(tid >> 2) * (unsigned)P.lrest + 4u * (tid & 3);      // element offset of unit 0 against the tile base
  const unsigned estep = (unsigned)(B / 4) * (unsigned)P.lrest;
  const char* const inb = reinterpret_cast<const char*>(reinterpret_cast<const float*>(io.in) + tbase);
  const char* const oldb = reinterpret_cast<const char*>(io.old + tbase);
  char* const outb = reinterpret_cast<char*>(io.out + tbase);
  const bool need_old = io.resid != nullptr;
  float4 v[EPT4];
#pragma unroll
  for (int k = 0; k < EPT4; ++k) {
    const bool rowok = !PART4 || tid + k * B < Geo::UNITS4;
    v[k] = *reinterpret_cast<const float4*>(inb + (rowok ? e0 + k * estep : e0) * 4u);
  }
  QFrag<N> q;
  q.load(P.Qx, lane);
#pragma unroll
  for (int k = 0; k < EPT4; ++k) {
    const int u = tid + k * B;
    if (!PART4 || u < Geo::UNITS4) {
      *reinterpret_cast<v2d*>(lds + 4 * u) = (v2d){(double)v[k].x, (double)v[k].y};
      *reinterpret_cast<v2d*>(lds + 4 * u + 2) = (v2d){(double)v[k].z, (double)v[k].w};
    }
  }
  v2d wv[EPT4][2];
  if (need_old) {
#pragma unroll
    for (int k = 0; k < EPT4; ++k) {
      const bool rowok = !PART4 || tid + k * B < Geo::UNITS4;
      const char* const pk = oldb + (size_t)(rowok ? e0 + k * estep : e0) * 8u;
      wv[k][0] = *reinterpret_cast<const v2d*>(pk);
      wv[k][1] = *reinterpret_cast<const v2d*>(pk + 16);
    }
  }
  const PowLane PT = pow_lane_init(lane);
  const double unscale = t32_scale_of(P.t32_ref > 0.0 ? P.t32_ref : io.old[P.ref_off], P.theta, PT, true);
  PowK<false> PK;
  PK.init(P.inv_theta, lane);
  __syncthreads();
  {
    double* const p0 = lds + li + lk * Geo::LX;
#pragma unroll
    for (int j = 0; j < (N + NW - 1) / NW; ++j) { if (N % NW == 0 || wave + j * NW < N) ctile<N, Geo::LX>(p0 + (wave + j * NW) * 16, q); __builtin_amdgcn_sched_barrier(0); }
  }
  q.load(P.Qy, lane);
  __syncthreads();
  {
    double* const p0 = lds + li + lk * LINE_R;
#pragma unroll
    for (int j = 0; j < (N + NW - 1) / NW; ++j) { if (N % NW == 0 || wave + j * NW < N) ctile<N, LINE_R>(p0 + (wave + j * NW) * Geo::LX, q); __builtin_amdgcn_sched_barrier(0); }
  }
  __syncthreads();
  const char* const a3b = reinterpret_cast<const char*>(P.a3);
  const unsigned a3x = (unsigned)P.a3x, a3y = (unsigned)P.a3y;
  unsigned ia3[4];
  {
    const long long pos = (long long)chunk * LINE_R + 4 * (tid & 3);
#pragma unroll
    for (int j = 0; j < 4; ++j) ia3[j] = (unsigned)(P.out_idx[o] + P.rest_idx[pos + j]);
  }
  double rmax = 0.0;
  bool rnan = false;
#pragma unroll
  for (int k = 0; k < EPT4; ++k) {
    const int u = tid + k * B;
    const bool rowok = !PART4 || u < Geo::UNITS4;
    const int row = rowok ? (u >> 2) : 0;
    const int x = row / N, y = row - x * N;
    const unsigned ixy = __umul24((unsigned)x, a3x) + __umul24((unsigned)y, a3y);
    double a3v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a3v[j] = *reinterpret_cast<const double*>(a3b + (ia3[j] + ixy) * 8u);
    const v2d s0 = *reinterpret_cast<const v2d*>(lds + 4 * (rowok ? u : tid)), s1 = *reinterpret_cast<const v2d*>(lds + 4 * (rowok ? u : tid) + 2);
    const double ksa[2] = {a3v[0] * (s0.x * unscale), a3v[1] * (s0.y * unscale)};
    const double ksb[2] = {a3v[2] * (s1.x * unscale), a3v[3] * (s1.y * unscale)};
    double ua[2], ub[2];
    PK.template run<2>(ksa, ua);
    PK.template run<2>(ksb, ub);
    const v2d ya = (v2d){1.0 + P.beta * ua[0], 1.0 + P.beta * ua[1]}, yb = (v2d){1.0 + P.beta * ub[0], 1.0 + P.beta * ub[1]};
    if (rowok) {
      if (need_old) {
        const double r0 = fabs(ya.x - wv[k][0].x), r1 = fabs(ya.y - wv[k][0].y), r2 = fabs(yb.x - wv[k][1].x), r3 = fabs(yb.y - wv[k][1].y);
        rnan |= (r0 != r0) | (r1 != r1) | (r2 != r2) | (r3 != r3);
        rmax = fmax(fmax(rmax, fmax(r0, r1)), fmax(r2, r3));
      }
      char* const pk = outb + (size_t)(e0 + k * estep) * 8u;
      *reinterpret_cast<v2d*>(pk) = ya;
      *reinterpret_cast<v2d*>(pk + 16) = yb;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (io.resid != nullptr) {
    if (rnan) rmax = __longlong_as_double(0x7ff0000000000000LL);
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) rmax = fmax(rmax, __shfl_xor(rmax, s));
    if (lane == 0) red[wave] = rmax;
    __syncthreads();
    if (tid == 0) {
      double r = red[0];
      for (int w = 1; w < NW; ++w) r = fmax(r, red[w]);
      atomicMax(io.resid, (unsigned long long)__double_as_longlong(r));
    }
  }
}

#ifndef SDFS_NO_VARIANT_TABLES
// whole chunks only (lrest % 16 == 0); fp64 streams
template <int N> struct StreamGeo {
  static constexpr int B = LineGeo<N>::B;
  static constexpr int WPC_MID = N == 32 ? 1 : 2;                       // persistent middle pass: workgroups per CU launched
  static constexpr int WPC_MID32 = N <= 20 ? 3 : WPC_MID;               // ... with fp32 streams: the tile is in flight in half the registers
  static constexpr int WPC_LAST = N == 16 ? 3 : LineGeo<N>::BPC;        // one tile per workgroup: register budget of the last pass
};
template <int N> inline line_fn line_stream_variant_a3f_n(int mode) {
  using G = StreamGeo<N>;
  switch (mode) {
    case L_TLAST: return (line_fn)line_stream_kernel<N, L_TLAST, G::WPC_LAST, true, G::B, false, true>;
    case L_TLAST_LIN: return (line_fn)line_stream_kernel<N, L_TLAST_LIN, G::WPC_LAST, true, G::B, false, true>;
    default: return nullptr;
  }
}
template <int N> inline line_fn line_stream_variant_n(int mode) {
  using G = StreamGeo<N>;
  switch (mode) {
    case L_JLAST: return (line_fn)line_stream_kernel<N, L_JLAST, LineGeo<N>::BPC, false, G::B, false>;      // (krylov_kernels.hpp: with <out, dot_with>)
    case L_MID: return (line_fn)line_stream_kernel<N, L_MID, G::WPC_MID, false, G::B, true>;
    case L_TLAST: return (line_fn)line_stream_kernel<N, L_TLAST, G::WPC_LAST, true, G::B, false>;
    case L_TLAST_LIN: return (line_fn)line_stream_kernel<N, L_TLAST_LIN, G::WPC_LAST, true, G::B, false>;
    default: return nullptr;
  }
}
// a3f: LineDesc::f1 / f2 are set (the aggregator's scale factorises for this pass)
inline line_fn line_stream_variant(int n, int mode, bool a3f = false) {
  if (a3f && (mode == L_TLAST || mode == L_TLAST_LIN)) {
    switch (n) {
      case 16: return line_stream_variant_a3f_n<16>(mode);
      case 20: return line_stream_variant_a3f_n<20>(mode);
      case 24: return line_stream_variant_a3f_n<24>(mode);
      case 32: return line_stream_variant_a3f_n<32>(mode);
      default: return nullptr;
    }
  }
  switch (n) {
    case 16: return line_stream_variant_n<16>(mode);
    case 20: return line_stream_variant_n<20>(mode);
    case 24: return line_stream_variant_n<24>(mode);
    case 32: return line_stream_variant_n<32>(mode);
    default: return nullptr;
  }
}
// opts.t_f32: the streamed forms on fp32 intermediates (middle pass: floats in, floats out; last pass of T: floats in)
template <int N> inline line_fn line_stream_t32_variant_n(int mode, bool a3f) {
  using G = StreamGeo<N>;
  if (mode == L_MID) return (line_fn)line_stream_kernel<N, L_MID, G::WPC_MID32, false, G::B, true, false, true, true>;
  if (mode == L_TLAST) return a3f ? (line_fn)line_stream_kernel<N, L_TLAST, G::WPC_LAST, true, G::B, false, true, true, false>
                                  : (line_fn)line_stream_kernel<N, L_TLAST, G::WPC_LAST, true, G::B, false, false, true, false>;
  return nullptr;
}
inline line_fn line_stream_t32_variant(int n, int mode, bool a3f) {
  switch (n) {
    case 16: return line_stream_t32_variant_n<16>(mode, a3f);
    case 20: return line_stream_t32_variant_n<20>(mode, a3f);
    case 24: return line_stream_t32_variant_n<24>(mode, a3f);
    case 32: return line_stream_t32_variant_n<32>(mode, a3f);
    default: return nullptr;
  }
}
inline int line_stream_wpc_mid(int n) { return n == 32 ? 1 : 2; }
inline int line_stream_wpc_mid32(int n) { return n <= 20 ? 3 : line_stream_wpc_mid(n); }
inline line_fn line_tlast32_variant(int n) {
  switch (n) {
    case 16: return (line_fn)line_tlast32_kernel<16, 3>;
    case 20: return (line_fn)line_tlast32_kernel<20, 3>;
    case 24: return (line_fn)line_tlast32_kernel<24, 2>;
    case 32: return (line_fn)line_tlast32_kernel<32, 1>;
    default: return nullptr;
  }
}
#endif

}  // namespace sdfs
