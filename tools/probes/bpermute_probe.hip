// Probe: ds_bpermute_b32 throughput per CU on gfx950 (8 / 16 waves per CU), vs ds_read_b32 and v_permlane-free readlane.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(1024) kb(int reps, int* sink) {
  int v = threadIdx.x * 7, idx = (threadIdx.x * 13) & 63;
  int a = 0, b = 1, c = 2, d = 3;
  for (int r = 0; r < reps; ++r) {
    a = __builtin_amdgcn_ds_bpermute((idx + a) << 2, v);
    b = __builtin_amdgcn_ds_bpermute((idx + b) << 2, v);
    c = __builtin_amdgcn_ds_bpermute((idx + c) << 2, v);
    d = __builtin_amdgcn_ds_bpermute((idx + d) << 2, v);
  }
  sink[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
__global__ void __launch_bounds__(1024) kr(int reps, int* sink) {
  __shared__ int tab[256];
  tab[threadIdx.x & 255] = threadIdx.x;
  __syncthreads();
  int idx = (threadIdx.x * 13) & 63;
  int a = 0, b = 1, c = 2, d = 3;
  for (int r = 0; r < reps; ++r) {
    a = tab[(idx + a) & 255]; b = tab[(idx + b) & 255]; c = tab[(idx + c) & 255]; d = tab[(idx + d) & 255];
  }
  sink[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
template <typename K> void run(const char* nm, K k, int threads, int* sink) {
  const int reps = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<<<256, threads>>>(10, sink); hipDeviceSynchronize();
  hipEventRecord(e0); k<<<256, threads>>>(reps, sink); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)reps * 4 * (threads / 64);    // wave-instructions per CU
  printf("%s %4d threads/CU: %.3f ms -> %.2f ns per wave-instruction per CU (%.1f cycles @2.1GHz)\n", nm, threads, ms,
         ms * 1e6 / n, ms * 1e6 / n * 2.1);
}
int main() {
  int* sink; hipMalloc(&sink, 256 * 1024 * 4);
  run("ds_bpermute_b32", kb, 512, sink); run("ds_bpermute_b32", kb, 1024, sink);
  run("ds_read_b32    ", kr, 512, sink); run("ds_read_b32    ", kr, 1024, sink);
  return 0;
}
