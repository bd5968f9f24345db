// Probe: the MEMORY skeleton of a pass (load a tile -> LDS -> barrier -> [idle "compute" time] -> store),
// for the tile shapes the planner can choose from on the GCY 20^6 grid.  No arithmetic: what is measured is how
// fast a given (rows x run) tile walk with a given number of resident workgroups streams 512 MB in / 512 MB out
// (plus an optional third read stream, the residual's w).  `sleep` adds s_sleep time between the barrier and the
// store phase to mimic the serial fp64 phases of the real kernel without touching the fp64 pipe.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Pat {
  int nrows, m1;            // rows of the tile; row r -> (r / m1) * S0 + (r % m1) * S1
  long long S0, S1;
  int run;                  // doubles per row (even)
  int nouter;
  int oext[4];
  long long ostride[4];
  long long ntiles;
  int third;                // also read a third stream with the tile's pattern
  int sleep;                // s_sleep(127) repetitions between the barrier and the stores (~8k cycles each 127)
};

__device__ __forceinline__ long long xcd_remap(long long b, long long n) {
  const long long q = n >> 3, r = n & 7;
  const long long x = b & 7, k = b >> 3;
  const long long start = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return start + k;
}

template <int EPT>
__global__ void __launch_bounds__(512) tile_copy(const double* __restrict__ in, const double* __restrict__ in2,
                                                 double* __restrict__ out, const Pat p) {
  extern __shared__ double lds[];
  long long t = xcd_remap(blockIdx.x, p.ntiles);
  long long base = 0;
#pragma unroll
  for (int k = 3; k >= 0; --k)
    if (k < p.nouter) { base += (t % p.oext[k]) * p.ostride[k]; t /= p.oext[k]; }
  const int ru = p.run >> 1;
  const int tot = p.nrows * ru;
  const int B = blockDim.x, tid = threadIdx.x;
  double2 v[EPT], w[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int u = tid + k * B;
    if (u < tot) {
      const int r = u / ru, c = u - r * ru;
      const long long off = base + (long long)(r / p.m1) * p.S0 + (long long)(r % p.m1) * p.S1 + 2 * c;
      v[k] = *reinterpret_cast<const double2*>(in + off);
      if (p.third) w[k] = *reinterpret_cast<const double2*>(in2 + off);
    }
  }
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int u = tid + k * B;
    if (u < tot) {
      if (p.third) { v[k].x += w[k].x; v[k].y += w[k].y; }
      *reinterpret_cast<double2*>(lds + 2 * u) = v[k];
    }
  }
  __syncthreads();
  for (int s = 0; s < p.sleep; ++s) __builtin_amdgcn_s_sleep(127);
  __syncthreads();
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int u = tid + k * B;
    if (u < tot) {
      const int r = u / ru, c = u - r * ru;
      const long long off = base + (long long)(r / p.m1) * p.S0 + (long long)(r % p.m1) * p.S1 + 2 * c;
      *reinterpret_cast<double2*>(out + off) = *reinterpret_cast<const double2*>(lds + 2 * u);
    }
  }
}

__global__ void __launch_bounds__(256) plain_copy(const double2* __restrict__ in, double2* __restrict__ out, long long n2) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long long)gridDim.x * 256) out[i] = in[i];
}

typedef void (*kfn)(const double*, const double*, double*, const Pat);
static kfn pick(int ept) {
  switch (ept) {
    case 4: return tile_copy<4>; case 8: return tile_copy<8>; case 10: return tile_copy<10>;
    case 13: return tile_copy<13>; case 16: return tile_copy<16>; default: return nullptr;
  }
}

static void run(const char* name, Pat p, int block, double* in, double* in2, double* out, double gb) {
  const int tot = p.nrows * p.run / 2;
  int ept = (tot + block - 1) / block;
  const int cand[5] = {4, 8, 10, 13, 16};
  int e = 16;
  for (int c : cand) if (c >= ept) { e = c; break; }
  kfn fn = pick(e);
  const size_t ldsb = (size_t)p.nrows * p.run * 8;
  hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
  int occ = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)fn, block, ldsb);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) fn<<<(unsigned)p.ntiles, block, ldsb>>>(in, in2, out, p);
  hipDeviceSynchronize();
  const int reps = 20;
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) fn<<<(unsigned)p.ntiles, block, ldsb>>>(in, in2, out, p);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  hipError_t err = hipGetLastError();
  printf("%-44s block %4d ept %2d lds %6zu B occ %d sleep %d: %.4f ms  %.2f TB/s%s\n", name, block, e, ldsb, occ, p.sleep, ms,
         gb / ms, err == hipSuccess ? "" : "  (ERROR)");
}

int main(int argc, char** argv) {
  const long long n = 20, N = n * n * n * n * n * n;
  double *in, *in2, *out;
  hipMalloc(&in, N * 8); hipMalloc(&in2, N * 8); hipMalloc(&out, N * 8);
  hipMemset(in, 0, N * 8); hipMemset(in2, 0, N * 8); hipMemset(out, 0, N * 8);
  const long long n2 = n * n, n3 = n2 * n, n4 = n3 * n, n5 = n4 * n;
  const double gb2 = 2.0 * N * 8 / 1e9, gb3 = 3.0 * N * 8 / 1e9;
  {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    plain_copy<<<4096, 256>>>((const double2*)in, (double2*)out, N / 2); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) plain_copy<<<4096, 256>>>((const double2*)in, (double2*)out, N / 2);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("%-44s %.4f ms  %.2f TB/s\n", "plain double2 copy (grid-stride)", ms, gb2 / ms);
  }
  for (int sleep = 0; sleep <= 2; ++sleep) {
    // A2: old P1 -- contiguous (d,e,f) tile
    { Pat p = {400, 20, n2, n, 20, 3, {20, 20, 20, 1}, {n5, n4, n3, 0}, 8000, 0, sleep};
      run("old P1: (d,e,f) contiguous 64 KB", p, 512, in, in2, out, gb2); }
    // A: old P2 -- (b,c,f), 160-B runs
    { Pat p = {400, 20, n4, n3, 20, 3, {20, 20, 20, 1}, {n5, n2, n, 0}, 8000, 0, sleep};
      run("old P2: (b,c,f) 160-B runs", p, 512, in, in2, out, gb2); }
    // A3: old P3 -- (a, e/2, f): 1600-B runs, third stream
    { Pat p = {200, 10, n5, n, 20, 4, {20, 20, 20, 2}, {n4, n3, n2, 10 * n}, 16000, 1, sleep};
      run("old P3: (a,e/2,f) 1600-B runs, 3 streams", p, 256, in, in2, out, gb3); }
    // B: line tiles (c,d | 16-chunk of (e,f))
    { Pat p = {400, 20, n3, n2, 16, 3, {20, 20, 25, 1}, {n5, n4, 16, 0}, 10000, 0, sleep};
      run("LT  P2: (c,d|16 of ef) 128-B lines", p, 320, in, in2, out, gb2);
      run("LT  P2: (c,d|16 of ef) 128-B lines", p, 256, in, in2, out, gb2); }
    // C: line tiles (a,b | 16-chunk of (c,d,e,f)), third stream
    { Pat p = {400, 20, n5, n4, 16, 1, {10000, 1, 1, 1}, {16, 0, 0, 0}, 10000, 1, sleep};
      run("LT  P3: (a,b|16 of cdef) 3 streams", p, 320, in, in2, out, gb3);
      run("LT  P3: (a,b|16 of cdef) 3 streams", p, 256, in, in2, out, gb3);
      p.third = 0;
      run("LT  P3: (a,b|16 of cdef) 2 streams", p, 320, in, in2, out, gb2); }
    // B8: 8-double chunks (64-B rows): half-size line tiles, twice the workgroups per CU (round 3: is a 128-byte line split over
    // two tiles that run side by side still fetched once?)
    { Pat p = {400, 20, n3, n2, 8, 3, {20, 20, 50, 1}, {n5, n4, 8, 0}, 20000, 0, sleep};
      run("LT  P2: (c,d|8 of ef) 64-B rows", p, 256, in, in2, out, gb2);
      run("LT  P2: (c,d|8 of ef) 64-B rows", p, 128, in, in2, out, gb2);
      p.third = 1;
      run("LT  P3: (c,d|8 of ef) 64-B rows, 3 streams", p, 256, in, in2, out, gb3);
      run("LT  P3: (c,d|8 of ef) 64-B rows, 3 streams", p, 128, in, in2, out, gb3); }
    { Pat p = {400, 20, n3, n2, 16, 3, {20, 20, 25, 1}, {n5, n4, 16, 0}, 10000, 1, sleep};
      run("LT  P3: (c,d|16 of ef) 128-B rows, 3 streams", p, 256, in, in2, out, gb3); }
    // B32: 32-double chunks (256-B runs), one block per CU
    { Pat p = {400, 20, n3, n2, 32, 3, {20, 20, 12, 1}, {n5, n4, 32, 0}, 4800, 0, sleep};
      run("LT  P2: (c,d|32 of ef) 256-B runs [partial]", p, 512, in, in2, out, gb2 * 4800 * 12800 / (double)N); }
    // D: wave-private contiguous slices (4 slices of 20x20 per wave)
    { Pat p = {80, 20, n2, n, 20, 1, {40000, 1, 1, 1}, {1600, 0, 0, 0}, 40000, 0, sleep};
      run("WP  P1: 4 (e,f) slices per wave, contiguous", p, 64, in, in2, out, gb2); }
    // D2: 256-thread blocks, 16 slices contiguous (25.6 KB)
    { Pat p = {320, 20, n2, n, 20, 1, {10000, 1, 1, 1}, {6400, 0, 0, 0}, 10000, 0, sleep};
      run("    P1: 16 (e,f) slices per 256-thr block", p, 256, in, in2, out, gb2); }
    // E: wave-private (a | 80-chunk), third stream
    { Pat p = {20, 1, n5, 0, 80, 1, {40000, 1, 1, 1}, {80, 0, 0, 0}, 40000, 1, sleep};
      run("WP  P3: (a|80 of rest) 640-B runs, 3 streams", p, 64, in, in2, out, gb3); }
    // E2: block (a | 400-chunk) 256 threads (like old P3 but line aligned, 3200-B runs)
    { Pat p = {20, 1, n5, 0, 400, 1, {8000, 1, 1, 1}, {400, 0, 0, 0}, 8000, 1, sleep};
      run("    P3: (a|400 of rest) 3200-B runs, 3 str", p, 256, in, in2, out, gb3); }
  }
  return 0;
}
