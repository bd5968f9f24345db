// Probe: HW_ID / XCC_ID as seen by 512 co-resident 512-thread workgroups (2 per CU with 64 KB LDS each).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void __launch_bounds__(512) k(unsigned* out) {
  extern __shared__ double lds[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  lds[threadIdx.x] = hw;
  __syncthreads();
  // stay resident long enough that all 512 blocks coexist
  long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < 200000) __builtin_amdgcn_s_sleep(32);
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
int main() {
  const int nb = 512;
  unsigned* d; hipMalloc(&d, nb * 2 * 4);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  k<<<nb, 512, 65000>>>(d);
  std::vector<unsigned> h(nb * 2);
  hipMemcpy(h.data(), d, nb * 2 * 4, hipMemcpyDeviceToHost);
  std::map<unsigned, int> slots, full;
  unsigned orall = 0, orx = 0;
  for (int b = 0; b < nb; ++b) {
    slots[((h[2 * b + 1] & 15u) << 8) | ((h[2 * b] >> 8) & 0xffu)]++;
    orall |= h[2 * b]; orx |= h[2 * b + 1];
  }
  printf("distinct slots %zu of %d blocks; OR(hw_id)=%08x OR(xcc)=%08x\n", slots.size(), nb, orall, orx);
  std::map<int, int> hist;
  for (auto& kv : slots) hist[kv.second]++;
  for (auto& kv : hist) printf("  %d slots hold %d blocks\n", kv.second, kv.first);
  for (int b = 0; b < 24; ++b) printf("  blk %2d hw %08x xcc %08x\n", b, h[2 * b], h[2 * b + 1]);
  return 0;
}
