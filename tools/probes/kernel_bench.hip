// Probe: the pair-plan kernels of csrc/fast_kernels.hpp and their persistent prefetching forms
// (csrc/stream_kernels.hpp) on the GCY 20^6 geometry, one kernel at a time, outside the library:
// HIP-event time per launch, algorithmic GB/s, and the maximum relative difference of the candidate's output
// from the baseline kernel's.  Synthetic operands (random row-stochastic matrices, w in [400, 900]).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o kernel_bench kernel_bench.hip && ./kernel_bench [n]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <algorithm>
#include <random>
#include <string>
#include <vector>

#define SDFS_NO_VARIANT_TABLES
// phase stamps of workgroup 8 (wave 0) for its first 12 tiles
__device__ long long g_stamps[12 * 8];
__device__ int g_stamp_it;
// (the counter lives in a register and the enable flag comes through a scalar load: a vector load in a stamp would
// wait, in order, for every prefetch issued before it)
#define SDFS_STREAM_STAMP_DECL int stamp_it = (blockIdx.x == 8 && __builtin_nontemporal_load(&g_stamp_it) == 0) ? 0 : 1000
#define SDFS_STREAM_STAMP(i) do { if (threadIdx.x == 0 && stamp_it < 12) g_stamps[stamp_it * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define SDFS_STREAM_STAMP_NEXT do { stamp_it++; } while (0)
#ifdef KB_OLD_POW
// A/B of the power routines inside one process: a COPY of csrc/ (tools/probes/kb.sh makes it: #pragma once keys on the
// file) compiled into namespace sdfs_old with the general routine of rounds 1-3
#ifndef KB_OLD_POWY
#define KB_OLD_POWY 0
#endif
#ifndef KB_OLD_NT
#define KB_OLD_NT 3
#endif
#define SDFS_POWY KB_OLD_POWY
#define SDFS_NT KB_OLD_NT
#define sdfs sdfs_old
#include "kb_oldpow/stream_kernels.hpp"
#undef sdfs
#undef SDFS_POWY
#undef SDFS_NT
#endif
#include "../../sdfs_via_autodiff_amd/csrc/stream_kernels.hpp"

using namespace sdfs;

// streaming copy: every lane keeps U 16-byte loads in flight; a workgroup walks chunks of 256 * U units (grid-stride)
template <int U, int NT>
__global__ void __launch_bounds__(256) copy_stream_kernel(const double2* __restrict__ in, double2* __restrict__ out, long long units) {
  typedef double v2 __attribute__((ext_vector_type(2)));
  const v2* src = reinterpret_cast<const v2*>(in);
  v2* dst = reinterpret_cast<v2*>(out);
  const long long chunk = 256LL * U;
  for (long long c0 = (long long)blockIdx.x * chunk; c0 < units; c0 += (long long)gridDim.x * chunk) {
    v2 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const long long i = c0 + threadIdx.x + 256LL * k;
      if (i < units) v[k] = (NT & 1) ? __builtin_nontemporal_load(src + i) : src[i];
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const long long i = c0 + threadIdx.x + 256LL * k;
      if (i < units) { if (NT & 2) __builtin_nontemporal_store(v[k], dst + i); else dst[i] = v[k]; }
    }
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static double time_ms(const std::function<void()>& f, int reps = 20) {
  static hipEvent_t e0, e1; static bool init = false;
  if (!init) { hipEventCreate(&e0); hipEventCreate(&e1); init = true; }
  f();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  CK(hipGetLastError());
  return ms / reps;
}

// candidates are registered first and then timed in interleaved rounds (clock ramp, thermal drift and cache state
// hit every candidate alike); the median over the rounds is reported
struct Cand { std::string name; std::function<void()> launch; std::function<double()> check; double bytes; std::vector<double> ms; };
static std::vector<Cand> g_cands;
// one launch with stamps: phase durations (s_memtime ticks at 2.09 GHz -> us) of workgroup 8's tiles
static void dump_stamps(const char* name, const std::function<void()>& launch) {
  int zero = 0;
  static long long zeros[96];
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zeros, sizeof zeros));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_it), &zero, sizeof zero));
  launch();
  CK(hipDeviceSynchronize());
  long long st[96]; int nit = 12;
  CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof st));
  printf("%s: phases per tile, us (park+issue | barrier | contract X | barrier | contract Y | barrier | epilogue | tile total)\n", name);
  for (int t = 0; t < nit && st[t * 8 + 7] != 0; ++t) {
    printf("   tile %2d:", t);
    for (int i = 0; i < 7; ++i) printf(" %6.2f", (st[t * 8 + i + 1] - st[t * 8 + i]) / 2090.0);
    printf("  | %6.2f", (st[t * 8 + 7] - st[t * 8]) / 2090.0);
    if (t > 0) printf("  (gap %5.2f)", (st[t * 8] - st[(t - 1) * 8 + 7]) / 2090.0);
    printf("\n");
  }
  int big = 1 << 20;
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_it), &big, sizeof big));
}
static void run_all(int rounds) {
  for (int i = 0; i < 300; ++i) g_cands[i % g_cands.size()].launch();      // spin-up
  CK(hipDeviceSynchronize());
  for (int r = 0; r < rounds; ++r)
    for (auto& c : g_cands) c.ms.push_back(time_ms(c.launch));
  for (auto& c : g_cands) {
    std::sort(c.ms.begin(), c.ms.end());
    const double med = c.ms[c.ms.size() / 2];
    printf("%-56s %.4f ms (min %.4f max %.4f)  %.2f TB/s  maxrel %.2e\n", c.name.c_str(), med, c.ms.front(), c.ms.back(), c.bytes / med / 1e9,
           c.check ? c.check() : 0.0);
  }
  g_cands.clear();
}

__global__ void max_rel_diff(const double* a, const double* b, long long n, double* out) {
  double m = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double d = fabs(a[i] - b[i]) / fmax(fabs(b[i]), 1e-300);
    m = fmax(m, d != d ? 1e300 : d);
  }
  for (int s = 32; s > 0; s >>= 1) m = fmax(m, __shfl_xor(m, s));
  if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long*)out, (unsigned long long)__double_as_longlong(m));
}
static double diff(const double* a, const double* b, long long n) {
  static double* d = nullptr;
  if (!d) CK(hipMalloc(&d, 8));
  CK(hipMemset(d, 0, 8));
  max_rel_diff<<<1024, 256>>>(a, b, n, d);
  double h; CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
  return h;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 20;
  if (n != 20 && n != 16) { printf("n = 16 or 20\n"); return 1; }
  const long long N = (long long)n * n * n * n * n * n;
  int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  printf("grid %d^6 = %lld points, %d CUs\n", n, N, ncu);
  std::mt19937_64 rng(1);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  std::vector<double> hw((size_t)N);
  for (auto& x : hw) x = 400.0 + 500.0 * U(rng);
  auto mkQ = [&]() {
    std::vector<double> q((size_t)n * n);
    for (int i = 0; i < n; ++i) { double s = 0; for (int j = 0; j < n; ++j) { q[i * n + j] = U(rng) + (i == j ? 3.0 : 0.0); s += q[i * n + j]; } for (int j = 0; j < n; ++j) q[i * n + j] /= s; }
    return q;
  };
  double *w, *tmpA, *tmpB, *outA, *outB, *Q[4], *a3;
  CK(hipMalloc(&w, N * 8)); CK(hipMalloc(&tmpA, N * 8)); CK(hipMalloc(&tmpB, N * 8)); CK(hipMalloc(&outA, N * 8)); CK(hipMalloc(&outB, N * 8));
  CK(hipMemcpy(w, hw.data(), N * 8, hipMemcpyHostToDevice));
  for (int i = 0; i < 4; ++i) { auto q = mkQ(); CK(hipMalloc(&Q[i], q.size() * 8)); CK(hipMemcpy(Q[i], q.data(), q.size() * 8, hipMemcpyHostToDevice)); }
  const long long n2 = (long long)n * n, n4 = n2 * n2;
  // a3[b, c, e, a] as in GCY: strides a:1, e:n, c:n^2, b:n^3
  // ... as a product F1[c, a] * F2[e, b], the form the GCY tables have (the last pass's two-table variants read F1 / F2)
  std::vector<double> hF1((size_t)n2), hF2((size_t)n2), ha3((size_t)n4);
  for (auto& x : hF1) x = 0.7 + 0.5 * U(rng);
  for (auto& x : hF2) x = 0.7 + 0.5 * U(rng);
  for (int b = 0; b < n; ++b) for (int c = 0; c < n; ++c) for (int e = 0; e < n; ++e) for (int a = 0; a < n; ++a)
    ha3[(size_t)(((b * n + c) * n + e) * n + a)] = hF1[(size_t)(c * n + a)] * hF2[(size_t)(e * n + b)];
  CK(hipMalloc(&a3, n4 * 8)); CK(hipMemcpy(a3, ha3.data(), n4 * 8, hipMemcpyHostToDevice));
  const double theta = -36.03, beta = 0.998;
  unsigned long long* resid; CK(hipMalloc(&resid, 8));
  unsigned* sched; CK(hipMalloc(&sched, TK_WORDS * 4)); CK(hipMemset(sched, 0, TK_WORDS * 4));

  // ---- pass 1: slices over the two fastest axes ------------------------------------------------------
  SliceDesc sd; memset(&sd, 0, sizeof sd);
  sd.nslices = N / n2; sd.Qf = Q[0]; sd.Qe = Q[1]; sd.theta = theta;
  const size_t slds = slice_lds_bytes(n, S_MID);
  const long long swt = (sd.nslices + slice_tile_slices(n, S_MID) - 1) / slice_tile_slices(n, S_MID);
  const unsigned sgrid0 = (unsigned)((swt + 3) / 4);
  typedef std::function<void()> Launch;
  // registers a candidate; `ref` (if given) is the launch whose output `refbuf` the candidate's `out` is compared with
  auto add = [&](const std::string& name, Launch l, double bytes, double* out, Launch ref, double* refbuf) {
    Cand c; c.name = name; c.bytes = bytes; c.launch = l;
    if (ref) c.check = [=]() { ref(); l(); return diff(out, refbuf, N); };
    g_cands.push_back(c);
  };
  auto slice_launch = [&](slice_fn fn, unsigned grid, const double* in, double* out) -> Launch {
    hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds);
    SliceIO io; memset(&io, 0, sizeof io); io.in = in; io.out = out; io.sched = sched;
    return [=]() { hipLaunchKernelGGL(fn, dim3(grid), dim3(256), slds, 0, sd, io); };
  };
  const double b2 = 16.0 * N, b3 = 24.0 * N;
  auto mk_line = [&](int ax0) {
    LineDesc L; memset(&L, 0, sizeof L);
    long long stride[6]; { long long st = 1; for (int a = 5; a >= 0; --a) { stride[a] = st; st *= n; } }
    L.lrest = stride[ax0 + 1]; L.nchunks = (int)(L.lrest / LINE_R); L.nouter = N / (n2 * L.lrest); L.ntiles = L.nouter * L.nchunks;
    L.Qx = Q[2]; L.Qy = Q[3]; L.inv_theta = 1.0 / theta; L.beta = beta; L.theta = theta; L.cbt = pow(beta, theta);
    const int a3s[6] = {1, (int)(n * n2), (int)n2, 0, n, 0};
    L.a3x = a3s[ax0]; L.a3y = a3s[ax0 + 1];
    std::vector<int> outv((size_t)L.nouter), restv((size_t)L.lrest);
    for (long long o = 0; o < L.nouter; ++o) { long long r = o; int idx = 0; for (int c = ax0 - 1; c >= 0; --c) { idx += (int)(r % n) * a3s[c]; r /= n; } outv[(size_t)o] = idx; }
    for (long long q = 0; q < L.lrest; ++q) { long long r = q; int idx = 0; for (int c = 5; c > ax0 + 1; --c) { idx += (int)(r % n) * a3s[c]; r /= n; } restv[(size_t)q] = idx; }
    int *od, *rd; CK(hipMalloc(&od, outv.size() * 4)); CK(hipMalloc(&rd, restv.size() * 4));
    CK(hipMemcpy(od, outv.data(), outv.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(rd, restv.data(), restv.size() * 4, hipMemcpyHostToDevice));
    L.a3 = a3; L.out_idx = od; L.rest_idx = rd;
    if (ax0 == 2) {                    // pair (c, d): outer o = a * n + b, x = c, position = e * n + f
      std::vector<double> f1((size_t)(L.nouter * n)), f2((size_t)(L.nouter * L.lrest));
      for (long long o = 0; o < L.nouter; ++o) {
        const int a_ = (int)(o / n), b_ = (int)(o % n);
        for (int x = 0; x < n; ++x) f1[(size_t)(o * n + x)] = hF1[(size_t)(x * n + a_)];
        for (long long q = 0; q < L.lrest; ++q) f2[(size_t)(o * L.lrest + q)] = hF2[(size_t)((q / n) * n + b_)];
      }
      double *d1, *d2; CK(hipMalloc(&d1, f1.size() * 8)); CK(hipMalloc(&d2, f2.size() * 8));
      CK(hipMemcpy(d1, f1.data(), f1.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d2, f2.data(), f2.size() * 8, hipMemcpyHostToDevice));
      L.f1 = d1; L.f2 = d2;
    }
    return L;
  };
  const size_t llds = line_lds_bytes(n);
  const int blk = line_block(n);
  auto line_launch = [&](line_fn fn, const LineDesc& L, unsigned grid, const double* in, double* out, const double* old, bool res, int blk = 256) -> Launch {
    hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)llds);
    LineIO io; memset(&io, 0, sizeof io); io.in = in; io.out = out; io.old = old; io.resid = res ? resid : nullptr; io.sched = sched;
    return [=]() { hipLaunchKernelGGL(fn, dim3(grid), dim3(blk), llds, 0, L, io); };
  };
  const int rounds = 7;
  { int big = 1 << 20; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_it), &big, sizeof big)); }
  if (n != 20) { printf("only n = 20 is wired up\n"); return 0; }
  const LineDesc L2 = mk_line(0), L3 = mk_line(2);
  const unsigned C = (unsigned)ncu;
  printf("power routine: %s; non-temporal hint on the grid streams: %d (bit 0 loads, bit 1 stores)\n", SDFS_POWY ? "powy (pre-scaled tables, round 4)" : "pow_fast_try (general, rounds 1-3)", SDFS_NT);
#ifdef KB_OLD_POW
  printf("sdfs_old:: kernels: power routine %s, non-temporal hint %d\n", KB_OLD_POWY ? "powy" : "general (rounds 1-3)", KB_OLD_NT);
#endif
  // ---- round 4 (a): what a streaming copy of one grid reaches on this box (the ceiling of a 16 B/point pass) ----
  {
    const double bytes = 16.0 * N;
    const long long units = N / 2;
    auto cp = [&](auto kern, unsigned grid, const char* name) {
      add(name, [=]() { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, (const double2*)w, (double2*)outA, units); }, bytes, nullptr, nullptr, nullptr);
    };
    cp(copy_stream_kernel<4, 0>, 8u * C, "copy: grid-stride, 4 x 16 B per lane in flight, 8 wg/CU");
    cp(copy_stream_kernel<8, 0>, 4u * C, "copy: grid-stride, 8 x 16 B per lane in flight, 4 wg/CU");
    cp(copy_stream_kernel<8, 0>, 8u * C, "copy: grid-stride, 8 x 16 B per lane in flight, 8 wg/CU");
    cp(copy_stream_kernel<13, 0>, 3u * C, "copy: grid-stride, 13 x 16 B per lane (a line tile), 3 wg/CU");
    cp(copy_stream_kernel<16, 0>, 4u * C, "copy: grid-stride, 16 x 16 B per lane in flight, 4 wg/CU");
    cp(copy_stream_kernel<8, 3>, 4u * C, "copy: grid-stride, 8 x 16 B, nontemporal, 4 wg/CU");
    cp(copy_stream_kernel<8, 3>, 8u * C, "copy: grid-stride, 8 x 16 B, nontemporal, 8 wg/CU");
    cp(copy_stream_kernel<16, 3>, 4u * C, "copy: grid-stride, 16 x 16 B, nontemporal, 4 wg/CU");
    {
      const unsigned grid1 = (unsigned)((units + 256 * 8 - 1) / (256 * 8));
      cp(copy_stream_kernel<8, 0>, grid1, "copy: one chunk per workgroup, 8 x 16 B per lane");
      cp(copy_stream_kernel<8, 3>, grid1, "copy: one chunk per workgroup, 8 x 16 B, nontemporal");
    }
    {
      const unsigned grid1 = (unsigned)((units + 256 * 8 - 1) / (256 * 8));
      cp(copy_stream_kernel<8, 1>, grid1, "copy: one chunk per workgroup, 8 x 16 B, nontemporal loads only");
      cp(copy_stream_kernel<8, 2>, grid1, "copy: one chunk per workgroup, 8 x 16 B, nontemporal stores only");
    }
    add("copy: hipMemcpyAsync device to device", [=]() { hipMemcpyAsync(outA, w, (size_t)N * 8, hipMemcpyDeviceToDevice, 0); }, bytes, nullptr, nullptr, nullptr);
    run_all(rounds);
  }
  // S = w^theta contracted over (e, f): the input of the later passes
  slice_launch((slice_fn)slice_kernel<20, S_TFIRST, false>, sgrid0, w, tmpA)();
  CK(hipDeviceSynchronize());
  {
    // ---- pass 1: the library's form, and the LDS-conflict variants (b): padded rows (22 doubles) at 1 / 2 / 4 waves
    // per workgroup (LDS per wave 12.8 -> 14.08 KB: 11 / 10 / 8 waves per CU against 12) -----------------------------
    Launch base = slice_launch((slice_fn)slice_kernel<20, S_TFIRST, false>, sgrid0, w, outA);
    add("slice_kernel<20,TFIRST> 4 waves/wg, rows of 20 (library)", base, b2, nullptr, nullptr, nullptr);
    auto sl = [&](slice_fn fn, int wv, bool pad, int g = 4) -> Launch {
      const size_t lds = (size_t)wv * g * 20 * (pad ? 22 : 20) * 8;
      hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      SliceIO io; memset(&io, 0, sizeof io); io.in = w; io.out = outB; io.sched = sched;
      const long long wt = (sd.nslices + g - 1) / g;
      const unsigned grid = (unsigned)((wt + wv - 1) / wv);
      return [=]() { hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * wv), lds, 0, sd, io); };
    };
#ifdef KB_OLD_POW
    {
      typedef void (*oslice_fn)(const sdfs_old::SliceDesc, const sdfs_old::SliceIO);
      oslice_fn fn = (oslice_fn)sdfs_old::slice_kernel<20, sdfs_old::S_TFIRST, false>;
      hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds);
      sdfs_old::SliceDesc so; memcpy(&so, &sd, sizeof so);
      sdfs_old::SliceIO io; memset(&io, 0, sizeof io); io.in = w; io.out = outB; io.sched = sched;
      add("sdfs_old::slice_kernel<20,TFIRST>", [=]() { hipLaunchKernelGGL(fn, dim3(sgrid0), dim3(256), slds, 0, so, io); }, b2, outB, base, outA);
    }
#endif
    // (d) a wave walking tiles, the next tile's loads issued as soon as the current one is parked
    add("slice_walk<20,TFIRST> (next tile in flight under the contractions) 3 waves/SIMD wgs " + std::to_string(3 * C), slice_launch((slice_fn)slice_walk_kernel<20, S_TFIRST>, 3 * C, w, outB), b2, outB, base, outA);
    add("slice_walk<20,TFIRST> Q fragments per tile, 3 waves/SIMD wgs " + std::to_string(3 * C), slice_launch((slice_fn)slice_walk_kernel<20, S_TFIRST, 3, true>, 3 * C, w, outB), b2, outB, base, outA);
    add("slice_walk<20,TFIRST> power in place in LDS, 3 waves/SIMD wgs " + std::to_string(3 * C), slice_launch((slice_fn)slice_walk_kernel<20, S_TFIRST, 3, false, true>, 3 * C, w, outB), b2, outB, base, outA);
    add("slice_walk<20,TFIRST> power in place in LDS, Q fragments per tile, 3 waves/SIMD wgs " + std::to_string(3 * C), slice_launch((slice_fn)slice_walk_kernel<20, S_TFIRST, 3, true, true>, 3 * C, w, outB), b2, outB, base, outA);
    add("slice_walk<20,TFIRST> power in place in LDS, one element at a time, 3 waves/SIMD wgs " + std::to_string(3 * C), slice_launch((slice_fn)slice_walk_kernel<20, S_TFIRST, 3, false, true, true>, 3 * C, w, outB), b2, outB, base, outA);
    add("slice_walk<20,TFIRST> 2 waves/SIMD wgs " + std::to_string(2 * C), slice_launch((slice_fn)slice_walk_kernel<20, S_TFIRST, 2>, 2 * C, w, outB), b2, outB, base, outA);
    add("slice_walk<20,TFIRST> Q fragments per tile, 2 waves/SIMD wgs " + std::to_string(2 * C), slice_launch((slice_fn)slice_walk_kernel<20, S_TFIRST, 2, true>, 2 * C, w, outB), b2, outB, base, outA);
    // (c) occupancy: three slices per wave tile (9.6 KB of LDS per wave: 16 waves per CU instead of 12; the last of its four
    // column tiles repeats four columns: +6.7 % MFMA work)
    add("slice_kernel<20,TFIRST> 4 waves/wg, 3 slices per wave tile, 4 waves/SIMD", sl((slice_fn)slice_kernel<20, S_TFIRST, false, 4, false, 3, 4>, 4, false, 3), b2, outB, base, outA);
    add("slice_kernel<20,TFIRST> 2 waves/wg, 3 slices per wave tile, 4 waves/SIMD", sl((slice_fn)slice_kernel<20, S_TFIRST, false, 2, false, 3, 4>, 2, false, 3), b2, outB, base, outA);
    add("slice_kernel<20,TFIRST> 1 wave/wg, 3 slices per wave tile, 4 waves/SIMD", sl((slice_fn)slice_kernel<20, S_TFIRST, false, 1, false, 3, 4>, 1, false, 3), b2, outB, base, outA);
    add("slice_kernel<20,TFIRST> 1 wave/wg, 3 slices, rows padded to 22, 4 waves/SIMD", sl((slice_fn)slice_kernel<20, S_TFIRST, false, 1, true, 3, 4>, 1, true, 3), b2, outB, base, outA);
    add("slice_kernel<20,TFIRST> 4 waves/wg, 2 slices per wave tile, 5 waves/SIMD", sl((slice_fn)slice_kernel<20, S_TFIRST, false, 4, false, 2, 5>, 4, false, 2), b2, outB, base, outA);
    add("slice_kernel<20,TFIRST> 1 wave/wg, rows of 20", sl((slice_fn)slice_kernel<20, S_TFIRST, false, 1, false>, 1, false), b2, outB, base, outA);
    add("slice_kernel<20,TFIRST> 2 waves/wg, rows of 20", sl((slice_fn)slice_kernel<20, S_TFIRST, false, 2, false>, 2, false), b2, outB, base, outA);
    add("slice_kernel<20,TFIRST> 1 wave/wg, rows padded to 22", sl((slice_fn)slice_kernel<20, S_TFIRST, false, 1, true>, 1, true), b2, outB, base, outA);
    add("slice_kernel<20,TFIRST> 2 waves/wg, rows padded to 22", sl((slice_fn)slice_kernel<20, S_TFIRST, false, 2, true>, 2, true), b2, outB, base, outA);
    add("slice_kernel<20,TFIRST> 4 waves/wg, rows padded to 22", sl((slice_fn)slice_kernel<20, S_TFIRST, false, 4, true>, 4, true), b2, outB, base, outA);
    Launch basem = slice_launch((slice_fn)slice_kernel<20, S_MID, false>, sgrid0, w, outA);
    add("slice_kernel<20,MID> (no power)", basem, b2, nullptr, nullptr, nullptr);
    add("slice_walk<20,MID> wgs " + std::to_string(3 * C), slice_launch((slice_fn)slice_walk_kernel<20, S_MID>, 3 * C, w, outB), b2, outB, basem, outA);
    add("slice_kernel<20,MID> 1 wave/wg, rows padded to 22 (no power)", sl((slice_fn)slice_kernel<20, S_MID, false, 1, true>, 1, true), b2, outB, basem, outA);
    run_all(rounds);
  }
  {
    // ---- pass 2: the persistent middle pass -----------------------------------------------------------------------
    Launch base = line_launch((line_fn)line_kernel<20, L_MID, false, true, false>, L2, (unsigned)L2.ntiles, tmpA, outA, nullptr, false);
    add("line_kernel<20,MID> pair (a,b), one tile per workgroup", base, b2, nullptr, nullptr, nullptr);
#ifdef KB_OLD_POW
    {
      typedef void (*oline_fn)(const sdfs_old::LineDesc, const sdfs_old::LineIO);
      oline_fn fn = (oline_fn)sdfs_old::line_stream_kernel<20, sdfs_old::L_MID, 2>;
      hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)llds);
      sdfs_old::LineDesc lo; memcpy(&lo, &L2, sizeof lo);
      sdfs_old::LineIO io; memset(&io, 0, sizeof io); io.in = tmpA; io.out = outB; io.sched = sched;
      add("sdfs_old::line_stream<20,MID,2,persistent> pair (a,b) wgs " + std::to_string(2 * C), [=]() { hipLaunchKernelGGL(fn, dim3(2 * C), dim3(256), llds, 0, lo, io); }, b2, outB, base, outA);
    }
#endif
    for (unsigned wg : {3u * C, 2u * C})
      add("line_stream<20,MID,2,persistent> pair (a,b) wgs " + std::to_string(wg), line_launch((line_fn)line_stream_kernel<20, L_MID, 2>, L2, wg, tmpA, outB, nullptr, false), b2, outB, base, outA);
    run_all(rounds);
  }
  {
    // ---- pass 3: the last pass (aggregator + residual) ----------------------------------------------------------------
    Launch base = line_launch((line_fn)line_kernel<20, L_TLAST, false, true, false>, L3, (unsigned)L3.ntiles, tmpA, outA, w, true);
    add("line_kernel<20,TLAST> pair (c,d) (general power, window of four)", base, b3, nullptr, nullptr, nullptr);
    add("line_stream<20,TLAST,3,OLDPF,A3F,nonpersist> pair (c,d) (library)", line_launch((line_fn)line_stream_kernel<20, L_TLAST, 3, true, 256, false, true>, L3, (unsigned)L3.ntiles, tmpA, outB, w, true), b3, outB, base, outA);
#ifdef KB_OLD_POW
    {
      typedef void (*oline_fn)(const sdfs_old::LineDesc, const sdfs_old::LineIO);
      oline_fn fn = (oline_fn)sdfs_old::line_stream_kernel<20, sdfs_old::L_TLAST, 3, true, 256, false, true>;
      hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)llds);
      sdfs_old::LineDesc lo; memcpy(&lo, &L3, sizeof lo);
      sdfs_old::LineIO io; memset(&io, 0, sizeof io); io.in = tmpA; io.out = outB; io.old = w; io.resid = resid; io.sched = sched;
      const unsigned grid = (unsigned)L3.ntiles;
      add("sdfs_old::line_stream<20,TLAST,3,OLDPF,A3F,nonpersist>", [=]() { hipLaunchKernelGGL(fn, dim3(grid), dim3(256), llds, 0, lo, io); }, b3, outB, base, outA);
    }
#endif
    add("line_stream<20,TLAST,3,OLDPF,256,nonpersist> pair (c,d) (a3 gathers)", line_launch((line_fn)line_stream_kernel<20, L_TLAST, 3, true, 256, false>, L3, (unsigned)L3.ntiles, tmpA, outB, w, true), b3, outB, base, outA);
    add("line_stream<20,TLAST,2,OLDPF,A3F,persistent> pair (c,d) wgs " + std::to_string(2 * C), line_launch((line_fn)line_stream_kernel<20, L_TLAST, 2, true, 256, true, true>, L3, 2 * C, tmpA, outB, w, true), b3, outB, base, outA);
    add("line_stream<20,TLAST,2,OLDPF,A3F,512thr,nonpersist> pair (c,d)", line_launch((line_fn)line_stream_kernel<20, L_TLAST, 2, true, 512, false, true>, L3, (unsigned)L3.ntiles, tmpA, outB, w, true, 512), b3, outB, base, outA);
    add("line_stream<20,TLAST,3,OLDPF,A3F,512thr,nonpersist> pair (c,d)", line_launch((line_fn)line_stream_kernel<20, L_TLAST, 3, true, 512, false, true>, L3, (unsigned)L3.ntiles, tmpA, outB, w, true, 512), b3, outB, base, outA);
    add("line_kernel<20,TLAST> pair (c,d), no residual", line_launch((line_fn)line_kernel<20, L_TLAST, false, true, false>, L3, (unsigned)L3.ntiles, tmpA, outA, w, false), b2, nullptr, nullptr, nullptr);
    run_all(rounds);
    dump_stamps("line_stream<20,TLAST,3,OLDPF,A3F,nonpersist>", line_launch((line_fn)line_stream_kernel<20, L_TLAST, 3, true, 256, false, true>, L3, (unsigned)L3.ntiles, tmpA, outB, w, true));
  }
  return 0;
}
