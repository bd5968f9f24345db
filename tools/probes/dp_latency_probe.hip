// Probe: fp64 FMA issue/latency on gfx950: NCH independent dependent-chains per wave, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NCH>
__global__ void __launch_bounds__(1024) k(int reps, long long* cyc, double* sink) {
  double a = 1.0 + threadIdx.x * 1e-9, b = threadIdx.x * 1e-3;
  double c[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) c[i] = i;
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) c[i] = fma(c[i], a, b);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < NCH; ++i) s += c[i];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NCH> void run(int wps, long long* cyc, double* sink) {
  const int reps = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NCH><<<256, 256 * wps>>>(10, cyc, sink); hipDeviceSynchronize();
  hipEventRecord(e0); k<NCH><<<256, 256 * wps>>>(reps, cyc, sink); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long t; hipMemcpy(&t, cyc, 8, hipMemcpyDeviceToHost);
  const double nf = (double)reps * 8 * NCH;            // FMAs per wave
  printf("chains %d waves/SIMD %d: %.3f ms  -> %.2f ns per FMA per wave, %.2f ns per FMA per SIMD (memtime ticks %lld)\n",
         NCH, wps, ms, ms * 1e6 / nf, ms * 1e6 / nf / wps, t);
}
int main() {
  long long* cyc; double* sink; hipMalloc(&cyc, 256 * 8); hipMalloc(&sink, 256 * 1024 * 8);
  for (int w = 1; w <= 4; w *= 2) { run<1>(w, cyc, sink); run<2>(w, cyc, sink); run<4>(w, cyc, sink); run<8>(w, cyc, sink); }
  return 0;
}
