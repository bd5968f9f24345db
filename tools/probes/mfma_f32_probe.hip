// Probe: lane maps and issue cost of the f32-input MFMA shapes the fp32 J.v kernels use (BASELINE config 5 "on MFMA"):
// v_mfma_f32_16x16x4_f32 and v_mfma_f32_4x4x1_16b_f32 on gfx950.  The maps are checked against the ones the kernels
// assume (csrc/f32_kernels.hpp):
//   16x16x4:    A lane l = A[row l & 15][k l >> 4],  B lane l = B[k l >> 4][col l & 15],  D lane l reg i = D[row 4 (l >> 4) + i][col l & 15]
//   4x4x1_16b:  A lane l = A_b[row l & 3],  B lane l = B_b[col l & 3],  b = l >> 2;  D lane l reg i = D_b[row i][col l & 3]
//   hipcc --offload-arch=gfx950 -O2 -o mfma_f32_probe mfma_f32_probe.hip && ./mfma_f32_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void onehot_16(int la, int lb, float* out) {
  const int l = threadIdx.x;
  const float a = (l == la) ? 1.f : 0.f, b = (l == lb) ? 1.f : 0.f;
  v4f c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) out[l * 4 + i] = c[i];
}
__global__ void onehot_4(int la, int lb, float* out) {
  const int l = threadIdx.x;
  const float a = (l == la) ? 1.f : 0.f, b = (l == lb) ? 1.f : 0.f;
  v4f c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) out[l * 4 + i] = c[i];
}
template <int KIND>
__global__ void time_k(long long* cyc, float* sink, int reps) {
  const float a = threadIdx.x * 0.001f, b = 1.f + threadIdx.x * 0.002f;
  v4f c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < reps; ++i) {
    if (KIND == 0) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
    } else if (KIND == 1) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
    } else if (KIND == 2) {
      c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
    } else {
      c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  sink[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

int main() {
  float* out; hipMalloc(&out, 256 * 4);
  std::vector<float> h(256);
  int bad16 = 0, n16 = 0, bad4 = 0, n4 = 0;
  for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) {
    onehot_16<<<1, 64>>>(la, lb, out);
    hipMemcpy(h.data(), out, 256 * 4, hipMemcpyDeviceToHost);
    // expected: A[row la & 15][k la >> 4] * B[k lb >> 4][col lb & 15] -> D[row][col] iff the k match
    const bool hit = (la >> 4) == (lb >> 4);
    const int row = la & 15, col = lb & 15;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
      const bool want = hit && (l & 15) == col && 4 * (l >> 4) + i == row;
      if ((h[l * 4 + i] != 0.f) != want) { if (bad16 < 10) printf("16x16x4: A%02d B%02d lane %02d reg %d = %g, expected %d\n", la, lb, l, i, h[l * 4 + i], (int)want); ++bad16; }
      n16 += want;
    }
    onehot_4<<<1, 64>>>(la, lb, out);
    hipMemcpy(h.data(), out, 256 * 4, hipMemcpyDeviceToHost);
    const bool hit4 = (la >> 2) == (lb >> 2);         // same block
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
      const bool want = hit4 && (l >> 2) == (la >> 2) && (l & 3) == (lb & 3) && i == (la & 3);
      if ((h[l * 4 + i] != 0.f) != want) { if (bad4 < 10) printf("4x4x1_16b: A%02d B%02d lane %02d reg %d = %g, expected %d\n", la, lb, l, i, h[l * 4 + i], (int)want); ++bad4; }
      n4 += want;
    }
  }
  printf("16x16x4 f32 lane map: %d mismatches (%d expected hits)\n", bad16, n16);
  printf("4x4x1_16b f32 lane map: %d mismatches (%d expected hits)\n", bad4, n4);
  long long* cyc; hipMalloc(&cyc, 8); float* sink; hipMalloc(&sink, 64 * 4);
  long long hc;
  const int reps = 10000;
  time_k<0><<<1, 64>>>(cyc, sink, reps); hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
  printf("16x16x4 f32, 4 independent accumulators: %.1f cycles/MFMA\n", (double)hc / (4.0 * reps));
  time_k<1><<<1, 64>>>(cyc, sink, reps); hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
  printf("16x16x4 f32, dependent chain: %.1f cycles/MFMA\n", (double)hc / (4.0 * reps));
  time_k<2><<<1, 64>>>(cyc, sink, reps); hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
  printf("4x4x1_16b f32, 4 independent accumulators: %.1f cycles/MFMA\n", (double)hc / (4.0 * reps));
  time_k<3><<<1, 64>>>(cyc, sink, reps); hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
  printf("4x4x1_16b f32, dependent chain: %.1f cycles/MFMA\n", (double)hc / (4.0 * reps));
  return 0;
}
