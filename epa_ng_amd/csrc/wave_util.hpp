// Wavefront-level helpers shared by the thorough kernels (gfx950, wave64).
#pragma once

#include <hip/hip_runtime.h>

namespace epa_wave {

__device__ __forceinline__ double readlane_d(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// v + (v moved by a DPP pattern); lanes without a source (or outside row_mask) add 0
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
  return v + __hiloint2double(hi, lo);
}

// wave64 sum without LDS: row_shr 1/2/4/8 inside the 16-lane rows, row_bcast 15 / 31 across
// rows, total lands in lane 63 and is broadcast through an SGPR pair (wave-uniform result).
__device__ __forceinline__ double wave_sum(double v) {
  v = dpp_add<0x111, 0xf>(v);  // row_shr:1
  v = dpp_add<0x112, 0xf>(v);  // row_shr:2
  v = dpp_add<0x114, 0xf>(v);  // row_shr:4
  v = dpp_add<0x118, 0xf>(v);  // row_shr:8
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 -> rows 1,3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 -> rows 2,3
  return readlane_d(v, 63);
}

// two wave64 sums for the price of one: v_permlane32_swap puts a's two 32-lane halves side by
// side in lanes 0..31 and b's in lanes 32..63, one add folds them, then row_shr 1/2/4/8 and
// row_bcast 15 finish each 32-lane half; a's total lands in lane 31, b's in lane 63
// (22 VALU instructions instead of 2 x 20).
__device__ __forceinline__ void wave_sum2(double a, double b, double& sa, double& sb) {
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
  double v = __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
  v = dpp_add<0x111, 0xf>(v);  // row_shr:1
  v = dpp_add<0x112, 0xf>(v);  // row_shr:2
  v = dpp_add<0x114, 0xf>(v);  // row_shr:4
  v = dpp_add<0x118, 0xf>(v);  // row_shr:8
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 -> rows 1,3
  sa = readlane_d(v, 31);
  sb = readlane_d(v, 63);
}

// 1/x: v_rcp_f64 + two Newton steps (the quotient feeds Newton's f, f' only)
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  return fma(r, e, r);
}

}  // namespace epa_wave
