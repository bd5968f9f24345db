// Probe: what does one dependent kernel launch cost inside a replayed hipGraph on this runtime?
//   hipcc --offload-arch=gfx950 -O2 -o launch_floor_probe launch_floor_probe.hip && ./launch_floor_probe
// A chain of `len` kernels, each reading what the previous one wrote (50625 doubles = SSY 15^4), captured once and
// replayed; variants: empty body, copy, copy with `work` dependent FMAs per element (a stand-in for the power).
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(256) k_empty(const double* in, double* out, int n, int work) {}
__global__ void __launch_bounds__(256) k_copy(const double* in, double* out, int n, int work) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    double v = in[i];
    for (int k = 0; k < work; ++k) v = fma(v, 0.999999, 1e-9);
    out[i] = v;
  }
}

template <typename K>
static float chain(K kern, int len, int blocks, int n, int work, double* a, double* b, hipStream_t st) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < len; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, n, work);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, st); hipStreamSynchronize(st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, st);
  for (int r = 0; r < 10; ++r) hipGraphLaunch(ge, st);
  hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return ms * 1e3f / (10.f * len);
}

int main() {
  const int n = 50625;
  double *a, *b;
  hipMalloc(&a, n * 8); hipMalloc(&b, n * 8);
  hipMemset(a, 0, n * 8); hipMemset(b, 0, n * 8);
  hipStream_t st; hipStreamCreate(&st);
  for (int len : {64, 512}) {
    printf("chain of %d kernels per graph replay\n", len);
    printf("  empty, 1 block        %6.2f us per kernel\n", chain(k_empty, len, 1, n, 0, a, b, st));
    printf("  empty, 57 blocks      %6.2f us per kernel\n", chain(k_empty, len, 57, n, 0, a, b, st));
    printf("  empty, 225 blocks     %6.2f us per kernel\n", chain(k_empty, len, 225, n, 0, a, b, st));
    printf("  copy, 198 blocks      %6.2f us per kernel\n", chain(k_copy, len, 198, n, 0, a, b, st));
    printf("  copy + 300 fma        %6.2f us per kernel\n", chain(k_copy, len, 198, n, 300, a, b, st));
    printf("  copy + 1000 fma       %6.2f us per kernel\n", chain(k_copy, len, 198, n, 1000, a, b, st));
  }
  // the same chain as plain stream launches
  {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_copy, dim3(198), dim3(256), 0, st, a, b, n, 0);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_copy, dim3(198), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, n, 0);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("plain stream launches, copy: %6.2f us per kernel\n", ms * 1e3f / 2000.f);
  }
  return 0;
}
