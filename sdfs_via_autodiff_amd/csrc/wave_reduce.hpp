// wave_reduce.hpp -- wave-wide sums and maxima of doubles on the VALU.
#pragma once
#include <hip/hip_runtime.h>

namespace sdfs {

// Wave reductions on the VALU (DPP moves inside the rows of 16, v_readlane across the four rows): ~30 instructions
// where six __shfl_xor steps of a double are twelve LDS-crossbar round trips (~1000 cycles measured at the end of
// a small-grid kernel, tools/probes/small_fused_probe.hip).  Every lane must be active; the result is wave-uniform.
template <int CTRL> __device__ __forceinline__ double dpp_mov_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror: afterwards every lane holds its row's result
__device__ __forceinline__ double wave_max_f64(double v) {      // NaN-free input (fmax drops NaNs)
  v = fmax(v, dpp_mov_f64<0xB1>(v));
  v = fmax(v, dpp_mov_f64<0x4E>(v));
  v = fmax(v, dpp_mov_f64<0x141>(v));
  v = fmax(v, dpp_mov_f64<0x140>(v));
  return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}
__device__ __forceinline__ double wave_sum_f64(double v) {
  v += dpp_mov_f64<0xB1>(v);
  v += dpp_mov_f64<0x4E>(v);
  v += dpp_mov_f64<0x141>(v);
  v += dpp_mov_f64<0x140>(v);
  return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

}  // namespace sdfs
