// Probe: do fp64 VALU FMAs and fp64 MFMAs from different waves of one SIMD overlap on gfx950?
//   hipcc --offload-arch=gfx950 -O2 -o coissue_probe coissue_probe.hip && ./coissue_probe
// One 512-thread block per CU (8 waves, 2 per SIMD).  mode bit0: waves 0-3 (one per SIMD) run an MFMA loop,
// bit1: waves 4-7 (one per SIMD) run a dependent-free FMA loop, mode 4: every wave runs both loops back to back,
// mode 5: every wave interleaves MFMA and FMA instructions in one loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double mfma_loop(int reps, double a, double b) {
  v4d c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
  for (int i = 0; i < reps; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
  }
  return c0[0] + c1[1];
}
__device__ __forceinline__ double fma_loop(int reps, double a, double b) {
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
  for (int i = 0; i < reps; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      c0 = fma(a, b, c0); c1 = fma(a, b, c1); c2 = fma(a, b, c2); c3 = fma(a, b, c3);
      c4 = fma(a, b, c4); c5 = fma(a, b, c5); c6 = fma(a, b, c6); c7 = fma(a, b, c7);
    }
  }
  return c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}
// 32 independent-ish integer VALU instructions per group (v_add_u32 / v_xor_b32 on 8 chains)
__device__ __forceinline__ double int_loop(int reps, int a, int b) {
  int c0 = a, c1 = b, c2 = a + 1, c3 = b + 1, c4 = a + 2, c5 = b + 2, c6 = a + 3, c7 = b + 3;
  for (int i = 0; i < reps; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      asm volatile("v_add_u32 %0, %0, %8\n\tv_xor_b32 %1, %1, %9\n\tv_add_u32 %2, %2, %8\n\tv_xor_b32 %3, %3, %9\n\t"
                   "v_add_u32 %4, %4, %8\n\tv_xor_b32 %5, %5, %9\n\tv_add_u32 %6, %6, %8\n\tv_xor_b32 %7, %7, %9"
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));
    }
  }
  return (double)(c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7);
}
__device__ __forceinline__ double mixed_loop(int reps, double a, double b) {
  v4d m0 = {0, 0, 0, 0}, m1 = {0, 0, 0, 0};
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
  for (int i = 0; i < reps; ++i) {
    m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, m0, 0, 0, 0);
    c0 = fma(a, b, c0); c1 = fma(a, b, c1); c2 = fma(a, b, c2); c3 = fma(a, b, c3);
    c4 = fma(a, b, c4); c5 = fma(a, b, c5); c6 = fma(a, b, c6); c7 = fma(a, b, c7);
    c0 = fma(a, b, c0); c1 = fma(a, b, c1); c2 = fma(a, b, c2); c3 = fma(a, b, c3);
    c4 = fma(a, b, c4); c5 = fma(a, b, c5); c6 = fma(a, b, c6); c7 = fma(a, b, c7);
    m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, m1, 0, 0, 0);
    c0 = fma(a, b, c0); c1 = fma(a, b, c1); c2 = fma(a, b, c2); c3 = fma(a, b, c3);
    c4 = fma(a, b, c4); c5 = fma(a, b, c5); c6 = fma(a, b, c6); c7 = fma(a, b, c7);
    c0 = fma(a, b, c0); c1 = fma(a, b, c1); c2 = fma(a, b, c2); c3 = fma(a, b, c3);
    c4 = fma(a, b, c4); c5 = fma(a, b, c5); c6 = fma(a, b, c6); c7 = fma(a, b, c7);
  }
  return m0[0] + m1[1] + c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}

// reps_m MFMA-pairs (2 x 64 cycles each), reps_f FMA groups (32 x 4 cycles each) -> equal issue time
__global__ void __launch_bounds__(512) probe(int mode, int reps, long long* cyc, double* sink) {
  const int wave = threadIdx.x >> 6;
  double a = threadIdx.x * 0.001, b = 1.0 + threadIdx.x * 0.002, r = 0;
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  if (mode == 4) {
    r = mfma_loop(reps, a, b);
    r += fma_loop(reps, a, b);
  } else if (mode == 5) {
    r = mixed_loop(reps, a, b);
  } else if (mode >= 6) {
    if (((wave >> 2) & 1) == 0) { if (mode == 7) r = mfma_loop(reps, a, b); }
    else r = int_loop(reps, threadIdx.x, threadIdx.x * 3 + 1);
  } else if (((wave >> 2) & 1) == 0) {
    if (mode & 1) r = mfma_loop(reps, a, b);
  } else {
    if (mode & 2) r = fma_loop(reps, a, b);
  }
  __syncthreads();
  long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 512 + threadIdx.x] = r;
}

int main() {
  const int blocks = 256, reps = 20000;
  long long* cyc; double* sink;
  hipMalloc(&cyc, blocks * sizeof(long long)); hipMalloc(&sink, blocks * 512 * sizeof(double));
  const char* names[] = {"", "mfma only (waves 0-3)", "fma only (waves 4-7)", "mfma 0-3 + fma 4-7", "all waves: mfma then fma", "all waves: interleaved", "int VALU only (waves 4-7)", "mfma 0-3 + int VALU 4-7"};
  for (int mode = 1; mode <= 7; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<<<blocks, 512>>>(mode, 100, cyc, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<<<blocks, 512>>>(mode, reps, cyc, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    printf("mode %d %-28s  %8.3f ms   (memtime ticks block0 %lld)\n", mode, names[mode], ms, h[0]);
  }
  printf("expected issue time per wave: mfma loop = reps*2*64 cycles, fma loop = reps*32*4 cycles (equal); int loop = reps*32 v_add/v_xor\n");
  return 0;
}
