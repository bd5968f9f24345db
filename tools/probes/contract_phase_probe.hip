// Probe: cost of ONE contraction step of pass_kernel (contract_step, LDS resident, no global traffic) per tile,
// by tile slot, block size and resident blocks per CU.  Ideal for n = 20: 25 column tiles x 5 k-steps x
// (64 + 20) cycles / 4 SIMDs = 2625 cycles per 20x20x20 tile and CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "../../sdfs_via_autodiff_amd/csrc/pass_kernel.hpp"
using namespace sdfs;

__global__ void __launch_bounds__(512, 4) k(const PassDesc P, int reps, int step, double* sink, unsigned long long* cyc) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int tot = P.L[0] * P.m[0] + 3 * 1024;
  for (int i = tid; i < tot; i += blockDim.x) lds[i] = 1.0 / (1.0 + (i % 97));
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    contract_step(lds, P, step, lane, wave, nwaves);
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  sink[blockIdx.x * blockDim.x + tid] = lds[tid];
}

static void run(const char* name, int m0, int m1, int m2, int pad, int slot, int block, int blocks_per_cu, double* sink,
                unsigned long long* cyc) {
  PassDesc P;
  memset(&P, 0, sizeof P);
  P.m[0] = m0; P.m[1] = m1; P.m[2] = m2;
  int L1 = m2;
  if (pad) while ((L1 & 3) != 2) ++L1;
  P.L[2] = 1; P.L[1] = L1; P.L[0] = L1 * m1;
  P.nsteps = 1; P.sslot[0] = slot; P.sn[0] = P.m[slot];
  int le = P.L[0] * m0; le += le & 1;
  P.qlds[0] = le; P.qlds[1] = le + 1024; P.qlds[2] = le + 2048;
  const size_t ldsb = (size_t)(le + 3 * 1024) * 8;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
  const int reps = 400;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<<<256 * blocks_per_cu, block, ldsb>>>(P, 4, 0, sink, cyc); hipDeviceSynchronize();
  hipEventRecord(e0); k<<<256 * blocks_per_cu, block, ldsb>>>(P, reps, 0, sink, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double cols = (double)m0 * m1 * m2 / P.sn[0];
  const double ideal = (cols / 16.0) * 5 * 84 / 4;     // n = 20
  printf("%-28s tile %2dx%2dx%2d L1 %2d slot %d block %3d x%d/CU: %.3f us/contraction/block  %6.0f memtime-ticks(100MHz)->%.0f cyc@2.1GHz  per CU-tile %.3f us (ideal %.0f cyc)\n",
         name, m0, m1, m2, L1, slot, block, blocks_per_cu, ms * 1e3 / reps, (double)c / reps, (double)c / reps * 21.0,
         ms * 1e3 / reps / blocks_per_cu, ideal);
}

int main() {
  double* sink; hipMalloc(&sink, 1024 * 512 * 8);
  unsigned long long* cyc; hipMalloc(&cyc, 8);
  for (int bpc = 1; bpc <= 2; ++bpc) {
    run("P1 f (slot 2, padded)", 20, 20, 20, 1, 2, 512, bpc, sink, cyc);
    run("P1 e (slot 1, padded)", 20, 20, 20, 1, 1, 512, bpc, sink, cyc);
    run("P1 d (slot 0, padded)", 20, 20, 20, 1, 0, 512, bpc, sink, cyc);
    run("P2 c (slot 1)", 20, 20, 20, 0, 1, 512, bpc, sink, cyc);
    run("P2 b (slot 0)", 20, 20, 20, 0, 0, 512, bpc, sink, cyc);
    run("P2 b (slot 0) 320 thr", 20, 20, 20, 0, 0, 320, bpc, sink, cyc);
    run("P2 b (slot 0) 256 thr", 20, 20, 20, 0, 0, 256, bpc, sink, cyc);
  }
  for (int bpc = 1; bpc <= 4; bpc *= 2) {
    run("P3 a (slot 0) half tile", 20, 10, 20, 0, 0, 256, bpc, sink, cyc);
    run("LT (x,y,16) slot 0", 20, 20, 16, 0, 0, 320, bpc, sink, cyc);
    run("LT (x,y,16) slot 1", 20, 20, 16, 0, 1, 320, bpc, sink, cyc);
    run("LT (x,y,16) slot 0 256thr", 20, 20, 16, 0, 0, 256, bpc, sink, cyc);
  }
  run("LT (x,y,16) slot 0 x3", 20, 20, 16, 0, 0, 320, 3, sink, cyc);
  run("LT (x,y,16) slot 1 x3", 20, 20, 16, 0, 1, 320, 3, sink, cyc);
  return 0;
}
