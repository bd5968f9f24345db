// Probe: cost of the pow prologue loop of pass_kernel in isolation (LDS resident, no global traffic),
// with 1 or 2 workgroups of 512 threads per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../sdfs_via_autodiff_amd/csrc/pass_kernel.hpp"
using namespace sdfs;
template <bool HI>
__global__ void __launch_bounds__(512, 4) k(int reps, double y, double* sink) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 8000; i += 512) lds[i] = 400.0 + i * 0.05;
  __syncthreads();
  const PowLane PT = pow_lane_init(lane);
  for (int r = 0; r < reps; ++r) {
#pragma unroll 1
    for (int it = 0; it < 8; ++it) {
      const int u = tid + it * 512;
      const bool valid = u < 4000;
      const int lo = valid ? 2 * u : 0;
      double2 x = *reinterpret_cast<double2*>(lds + lo);
      double xin[2] = {valid ? x.x : 1.0, valid ? x.y : 1.0}, xw[2];
      pow_fast_n<HI, 2>(xin, y, PT, xw);
      if (valid) { x.x = 400.0 + xw[0] * 1e-300; x.y = 401.0 + xw[1] * 1e-300; *reinterpret_cast<double2*>(lds + lo) = x; }
    }
  }
  __syncthreads();
  sink[blockIdx.x * 512 + tid] = lds[tid];
}
template <bool HI> void run(int blocks_per_cu, double y, double* sink) {
  const int reps = 200;
  auto fn = k<HI>;
  hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  fn<<<256 * blocks_per_cu, 512, 64000>>>(2, y, sink); hipDeviceSynchronize();
  hipEventRecord(e0); fn<<<256 * blocks_per_cu, 512, 64000>>>(reps, y, sink); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("HIPREC %d, %d block(s)/CU: %.3f ms -> %.2f us per tile-pow per block, %.2f us per tile per CU\n", (int)HI,
         blocks_per_cu, ms, ms * 1e3 / reps, ms * 1e3 / reps / blocks_per_cu);
}
int main() {
  double* sink; hipMalloc(&sink, 512 * 512 * 8);
  run<true>(1, -36.0, sink); run<true>(2, -36.0, sink);
  run<false>(1, -1.0 / 36, sink); run<false>(2, -1.0 / 36, sink);
  return 0;
}
