// Probe: where do the microseconds of the fused Anderson passes of the small-grid plan go (SM_AND_FIRST / SM_AND_LAST,
// fast_kernels.hpp)?  SSY 15^4 shape, synthetic stochastic matrices; chains of [first pass, last pass] pairs replayed
// from a hipGraph, the first pass with its step kind forced (0: control workgroup only, 1: rejection check in every
// workgroup, 2: full mixing step in every workgroup), against the plain pair of T.  Workgroup 0 / thread 0 stamps
// s_memtime at the kernel's phase boundaries, the control workgroup at those of the step.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I sdfs_via_autodiff_amd/csrc -o anderson_fused_probe tools/probes/anderson_fused_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>
__device__ unsigned long long g_stamps[2][16];
__device__ unsigned long long g_cstamps[4];
#define SDFS_SMALL_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_stamps[(MODE == 9 || MODE == 4) ? 1 : 0][i] = __builtin_amdgcn_s_memtime(); } while (0)
#define SDFS_AND_STAMP(i) do { if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) g_cstamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define SDFS_NO_VARIANT_TABLES
#include "fast_kernels.hpp"
using namespace sdfs;

int main() {
  const int n = 15, N = n * n * n * n, m = 10;
  std::vector<double> qp(256, 0.0), w(N), a3(N, 1.3);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) qp[i * 16 + j] = 1.0 / n;
  for (int i = 0; i < N; ++i) w[i] = 400.0 + (i % 97);
  std::vector<int> oi(n * n), ri(n * n), one(1, 0);
  for (int i = 0; i < n * n; ++i) { oi[i] = i * n * n; ri[i] = i; }
  double *dq, *dx, *dfx, *dtmp, *da3, *dpart, *derr; int *doi, *dri, *done, *dkind; AndState* dS;
  hipMalloc(&dq, 256 * 8); hipMalloc(&dx, N * 8); hipMalloc(&dfx, N * 8); hipMalloc(&dtmp, N * 8); hipMalloc(&da3, N * 8);
  hipMalloc(&doi, n * n * 4); hipMalloc(&dri, n * n * 4); hipMalloc(&done, 4); hipMalloc(&dpart, 8 * 16 * 512); hipMalloc(&derr, 8 * 4096); hipMalloc(&dkind, 4 * 4096);
  hipMalloc(&dS, 2 * sizeof(AndState)); unsigned* dflag; hipMalloc(&dflag, 4 * 4096); hipMemset(dflag, 0, 4 * 4096);
  hipMemcpy(dq, qp.data(), 256 * 8, hipMemcpyHostToDevice); hipMemcpy(dx, w.data(), N * 8, hipMemcpyHostToDevice);
  hipMemcpy(dfx, w.data(), N * 8, hipMemcpyHostToDevice); hipMemcpy(dtmp, w.data(), N * 8, hipMemcpyHostToDevice);
  hipMemcpy(da3, a3.data(), N * 8, hipMemcpyHostToDevice); hipMemcpy(doi, oi.data(), n * n * 4, hipMemcpyHostToDevice);
  hipMemcpy(dri, ri.data(), n * n * 4, hipMemcpyHostToDevice); hipMemcpy(done, one.data(), 4, hipMemcpyHostToDevice);
  AndPtrs hp; memset(&hp, 0, sizeof hp);
  for (int j = 0; j < m; ++j) { hipMalloc(&hp.X[j], N * 8); hipMalloc(&hp.R[j], N * 8); hipMemcpy(hp.X[j], w.data(), N * 8, hipMemcpyHostToDevice); hipMemset(hp.R[j], 0, N * 8); }
  SmallDesc dF, dL;
  memset(&dF, 0, sizeof dF);
  dF.nx = n; dF.ny = n; dF.my = (65536 + n - 1) / n; dF.sx = n; dF.sy = 1; dF.ostride = n * n; dF.lrest = 1; dF.nchunks = 1; dF.ntiles = n * n;
  dF.Qxp = dq; dF.Qyp = dq; dF.theta = -30.0; dF.inv_theta = 1.0 / -30.0; dF.beta = 0.999; dF.cbt = pow(0.999, -30.0); dF.a3 = da3; dF.out_idx = doi; dF.rest_idx = done;
  dF.a3x = n; dF.a3y = 1;
  dL = dF;
  dL.sx = n * n * n; dL.sy = n * n; dL.ostride = N; dL.lrest = n * n; dL.nchunks = n * n; dL.ntiles = n * n; dL.out_idx = done; dL.rest_idx = dri; dL.a3x = n * n * n; dL.a3y = n * n;
  hipStream_t st; hipStreamCreate(&st);
  const int len = 128;
  const int nb = n * n;
  unsigned long long* dgate; hipMalloc(&dgate, 8); hipMemset(dgate, 0xff, 8);
  for (int xcd : {1}) for (int exp : {0}) for (int variant = -1; variant <= 2; ++variant) {
    dL.cpx = xcd ? (nb + 7) / 8 : 0;
    const int gl = xcd ? 8 * ((nb + 7) / 8) : nb;
    AndState I; memset(&I, 0, sizeof I);
    I.err = 1e300; I.prev_pos = -1.0; I.mix_rel = -1; I.gate = ~0ULL;
    hipMemcpy(dS, &I, sizeof I, hipMemcpyHostToDevice); hipMemcpy(dS + 1, &I, sizeof I, hipMemcpyHostToDevice);
    hipMemcpy(dfx, w.data(), N * 8, hipMemcpyHostToDevice);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 1; i <= len; ++i) {
      SmallIO io; AndArgs an;
      memset(&io, 0, sizeof io); memset(&an, 0, sizeof an);
      an.h = hp; an.beta = 1.0; an.m = m; an.nb = nb;
      an.par.tol = -1.0; an.par.max_iter = 1e18; an.par.ridge = 1e-6; an.par.mixing_freq = variant == 2 ? 1 : 1 << 30;
      io.out = dtmp;
      if (variant < 0) {
        io.in = (i & 1) ? dx : dfx;
        hipLaunchKernelGGL((small_tile_kernel<SM_FIRST_T, 1, 4>), dim3(gl), dim3(256), 0, st, dL, io);
        memset(&io, 0, sizeof io);
        io.in = dtmp; io.out = (i & 1) ? dfx : dx;
        hipLaunchKernelGGL((small_tile_kernel<SM_LAST_T, 1, 4>), dim3(nb), dim3(256), 0, st, dF, io);
        continue;
      }
      io.in = dfx;
      an.Sin = dS + ((i - 1) & 1); an.Sout = dS + (i & 1); an.partial = dpart; an.x = dx; an.r_pos = hp.R[(i - 1) % m];
      an.err_slot = derr + i; an.kind_slot = dkind + i; an.flag = dflag + i - 1; an.pos = (i - 1) % m; an.rel = i - 1; an.step_kind = variant;
      hipLaunchKernelGGL((small_and_kernel<SM_AND_FIRST, 1, 4>), dim3(gl + 1), dim3(256), 0, st, dL, io, an);
      memset(&io, 0, sizeof io);
      io.in = dtmp; io.out = dfx; io.old = (exp & 2) ? dtmp : dx; io.gate = (exp & 1) ? dgate : &(dS + (i & 1))->gate;
      an.Sin = nullptr; an.Sout = nullptr; an.partial = nullptr; an.partial_out = dpart;
      an.x_pos = hp.X[i % m]; an.r_pos = hp.R[i % m]; an.pos = i % m; an.rel = i; an.step_kind = 0; an.flag = dflag + i;
      hipLaunchKernelGGL((small_and_kernel<SM_AND_LAST, 1, 4>), dim3(nb), dim3(256), 0, st, dF, io, an);
    }
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st);
    for (int r = 0; r < 10; ++r) hipGraphLaunch(ge, st);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long s[2][16], c[4];
    hipMemcpyFromSymbol(s, HIP_SYMBOL(g_stamps), sizeof s); hipMemcpyFromSymbol(c, HIP_SYMBOL(g_cstamps), sizeof c);
    AndState F; hipMemcpy(&F, dS, sizeof F, hipMemcpyDeviceToHost);
    printf("exp %d (bit 0: constant gate word, bit 1: old = tmp); xcd-aware last pass %d, variant %d (-1 plain T pair; step kind 0/1/2): %.2f us per pass  [it %.0f gate %llx err %.3e mode %d]\n", exp, xcd, variant, ms * 1e3 / (10.0 * len), F.it, F.gate, F.err, F.mix_mode);
    for (int k = 0; k < 2; ++k) { printf("   %s pass stamps (cycles):", k ? "last " : "first"); for (int i = 1; i < 10; ++i) printf(" [%d] %lld", i, (long long)(s[k][i] - s[k][0])); printf("\n"); }
    if (variant >= 0) printf("   control workgroup: sums done +%lld, step done +%lld (since its start)\n", (long long)(c[1] - c[0]), (long long)(c[2] - c[0]));
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  return 0;
}
