// Probe: lane maps and issue cost of the two f64 MFMA shapes on gfx950.
//   hipcc --offload-arch=gfx950 -O2 -o mfma_f64_probe mfma_f64_probe.hip && ./mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void onehot_4x4(int la, int lb, double* out) {
  const int l = threadIdx.x;
  double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
  double c = 0.0;
  c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
  out[l] = c;
}

__global__ void time_16(long long* cyc, double* sink, int reps) {
  double a = threadIdx.x * 0.001, b = 1.0 + threadIdx.x * 0.002;
  v4d c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0}, c3 = {0, 0, 0, 0};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < reps; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  sink[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ void time_16dep(long long* cyc, double* sink, int reps) {
  double a = threadIdx.x * 0.001, b = 1.0 + threadIdx.x * 0.002;
  v4d c0 = {0, 0, 0, 0};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < reps; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  sink[threadIdx.x] = c0[0];
}
__global__ void time_4(long long* cyc, double* sink, int reps) {
  double a = threadIdx.x * 0.001, b = 1.0 + threadIdx.x * 0.002;
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < reps; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  sink[threadIdx.x] = c0 + c1 + c2 + c3;
}
__global__ void time_fma(long long* cyc, double* sink, int reps) {
  double a = threadIdx.x * 0.001, b = 1.0 + threadIdx.x * 0.002;
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < reps; ++i) {
    c0 = fma(a, b, c0); c1 = fma(a, b, c1); c2 = fma(a, b, c2); c3 = fma(a, b, c3);
    c4 = fma(a, b, c4); c5 = fma(a, b, c5); c6 = fma(a, b, c6); c7 = fma(a, b, c7);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  sink[threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}

int main() {
  double* out; hipMalloc(&out, 64 * 8);
  std::vector<double> h(64);
  // map[la][lb] -> list of output lanes
  printf("4x4x4_4b one-hot map: for la (A lane) and lb (B lane), output lane with value 1\n");
  int cnt = 0;
  for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) {
    onehot_4x4<<<1, 64>>>(la, lb, out);
    hipMemcpy(h.data(), out, 64 * 8, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) if (h[l] != 0.0) { if (cnt < 400) printf("A%02d B%02d -> D%02d (%g)\n", la, lb, l, h[l]); ++cnt; }
  }
  printf("nonzero pairs: %d\n", cnt);
  long long* cyc; hipMalloc(&cyc, 8 * 8); double* sink; hipMalloc(&sink, 64 * 8);
  long long hc[8];
  const int reps = 10000;
  time_16<<<1, 64>>>(cyc, sink, reps); hipMemcpy(hc, cyc, 8, hipMemcpyDeviceToHost);
  printf("16x16x4 f64, 4 independent accumulators: %.1f cycles/MFMA (s_memtime ticks)\n", (double)hc[0] / (4.0 * reps));
  time_16dep<<<1, 64>>>(cyc, sink, reps); hipMemcpy(hc, cyc, 8, hipMemcpyDeviceToHost);
  printf("16x16x4 f64, dependent chain: %.1f cycles/MFMA\n", (double)hc[0] / (4.0 * reps));
  time_4<<<1, 64>>>(cyc, sink, reps); hipMemcpy(hc, cyc, 8, hipMemcpyDeviceToHost);
  printf("4x4x4_4b f64, 4 independent accumulators: %.1f cycles/MFMA\n", (double)hc[0] / (4.0 * reps));
  time_fma<<<1, 64>>>(cyc, sink, reps); hipMemcpy(hc, cyc, 8, hipMemcpyDeviceToHost);
  printf("v_fma_f64, 8 independent: %.2f cycles/FMA (one wave)\n", (double)hc[0] / (8.0 * reps));
  time_fma<<<1, 256>>>(cyc, sink, reps); hipMemcpy(hc, cyc, 8, hipMemcpyDeviceToHost);
  printf("v_fma_f64, 8 independent, 4 waves (one per SIMD): %.2f cycles/FMA per wave\n", (double)hc[0] / (8.0 * reps));
  time_fma<<<1, 512>>>(cyc, sink, reps); hipMemcpy(hc, cyc, 8, hipMemcpyDeviceToHost);
  printf("v_fma_f64, 8 independent, 8 waves (two per SIMD): %.2f cycles/FMA per wave\n", (double)hc[0] / (8.0 * reps));
  return 0;
}
