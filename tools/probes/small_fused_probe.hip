// Probe: where do the microseconds of one fused successive-approximation kernel of the small-grid plan go?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I sdfs_via_autodiff_amd/csrc -o small_fused_probe tools/probes/small_fused_probe.hip
// SSY 15^4 shape (225 slices of 15 x 15), synthetic stochastic matrices, a chain of fused kernels replayed from a
// hipGraph (resid 1: atomicMax into one word per kernel; 0: no residual; 2: per-workgroup maxima reduced by the next
// kernel).  Workgroup 0, wave 0 records s_memtime (shader clock) at the phase boundaries of the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>
__device__ unsigned long long g_stamps[16];
#define SDFS_SMALL_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#include "fast_kernels.hpp"
using namespace sdfs;

int main() {
  const int n = 15, N = n * n * n * n;
  std::vector<double> qp(256, 0.0), w(N), a3(N, 1.3);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) qp[i * 16 + j] = 1.0 / n;
  for (int i = 0; i < N; ++i) w[i] = 400.0 + (i % 97);
  std::vector<int> oi(n * n), ri(1, 0);
  for (int i = 0; i < n * n; ++i) oi[i] = i * n * n;
  double *dq, *dw0, *dw1, *dtmp, *da3; int *doi, *dri; unsigned long long* dres;
  hipMalloc(&dq, 256 * 8); hipMalloc(&dw0, N * 8); hipMalloc(&dw1, N * 8); hipMalloc(&dtmp, N * 8); hipMalloc(&da3, N * 8);
  hipMalloc(&doi, n * n * 4); hipMalloc(&dri, 4); hipMalloc(&dres, 8 * 1024); double* dring; hipMalloc(&dring, 8 * 2 * SMALL_RING); hipMemset(dring, 0, 8 * 2 * SMALL_RING);
  hipMemcpy(dq, qp.data(), 256 * 8, hipMemcpyHostToDevice); hipMemcpy(dw0, w.data(), N * 8, hipMemcpyHostToDevice);
  hipMemcpy(dw1, w.data(), N * 8, hipMemcpyHostToDevice); hipMemcpy(dtmp, w.data(), N * 8, hipMemcpyHostToDevice);
  hipMemcpy(da3, a3.data(), N * 8, hipMemcpyHostToDevice); hipMemcpy(doi, oi.data(), n * n * 4, hipMemcpyHostToDevice);
  hipMemcpy(dri, ri.data(), 4, hipMemcpyHostToDevice); hipMemset(dres, 0, 8 * 1024);
  SmallDesc d;
  memset(&d, 0, sizeof d);
  d.nx = n; d.ny = n; d.my = (65536 + n - 1) / n; d.sx = n; d.sy = 1; d.ostride = n * n; d.lrest = 1; d.nchunks = 1; d.ntiles = n * n;
  d.Qxp = dq; d.Qyp = dq; d.theta = -30.0; d.inv_theta = 1.0 / -30.0; d.beta = 0.999; d.cbt = pow(0.999, -30.0); d.a3 = da3; d.out_idx = doi; d.rest_idx = dri;
  d.a3x = n; d.a3y = 1;
  hipStream_t st; hipStreamCreate(&st);
  const int len = 256;
  for (int wpt : {1, 4}) for (int resid : {1, 0, 2}) for (int mode : {SM_FUSED_T, SM_LAST_T, SM_FIRST_T, SM_MID}) {
    if (!resid && mode != SM_FUSED_T && mode != SM_LAST_T) continue;
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < len; ++i) {
      SmallIO io;
      memset(&io, 0, sizeof io);
      io.in = dtmp; io.out = (i & 1) ? dw0 : dw1; io.old = (i & 1) ? dw1 : dw0; io.resid = resid == 1 ? dres + i : nullptr; io.aux_out = dtmp;
      if (resid == 2) { io.gate_part = dring + ((i + 1) & 1) * SMALL_RING; io.gate_n = small_grid(n * n, wpt); io.part_out = dring + (i & 1) * SMALL_RING; io.slot_out = dres + i; io.gate_tol = -1.0; }
      if (mode == SM_FIRST_T || mode == SM_MID) { io.in = (i & 1) ? dw1 : dw0; io.out = (i & 1) ? dw0 : dw1; }
      if (resid != 2) { io.gate = dres + 1000; io.gate_tol = -1.0; }
      hipLaunchKernelGGL(small_variant(mode, 1, wpt), dim3(small_grid(n * n, wpt)), dim3(256), 0, st, d, io);
    }
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipMemset(dres + 1000, 0xff, 8);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st);
    for (int r = 0; r < 10; ++r) hipGraphLaunch(ge, st);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long s[16];
    hipMemcpyFromSymbol(s, HIP_SYMBOL(g_stamps), sizeof s);
    printf("waves per tile %d resid %d mode %d: %.2f us per kernel; stamps (cycles since kernel entry):", wpt, resid, mode, ms * 1e3 / (10.0 * len));
    for (int i = 1; i < 10; ++i) printf(" [%d] %lld", i, (long long)(s[i] - s[0]));
    printf("\n");
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  return 0;
}
