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

// wave64 maximum of doubles / minimum of unsigned integers on the same DPP ladder as wave_sum (no LDS
// round trips: __shfl_xor is ds_bpermute); lanes without a source take the identity (-inf / ~0u).
// The result is wave-uniform (read from lane 63 through scalar registers).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_max(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)0xfff00000u, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return fmax(v, __hiloint2double(hi, lo));
}
__device__ __forceinline__ double wave_max_d(double v) {
  v = dpp_max<0x111, 0xf>(v);
  v = dpp_max<0x112, 0xf>(v);
  v = dpp_max<0x114, 0xf>(v);
  v = dpp_max<0x118, 0xf>(v);
  v = dpp_max<0x142, 0xa>(v);
  v = dpp_max<0x143, 0xc>(v);
  return readlane_d(v, 63);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_min_u(unsigned v) {
  const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, CTRL, ROW_MASK, 0xf, false);
  return o < v ? o : v;
}
__device__ __forceinline__ unsigned wave_min_u(unsigned v) {
  v = dpp_min_u<0x111, 0xf>(v);
  v = dpp_min_u<0x112, 0xf>(v);
  v = dpp_min_u<0x114, 0xf>(v);
  v = dpp_min_u<0x118, 0xf>(v);
  v = dpp_min_u<0x142, 0xa>(v);
  v = dpp_min_u<0x143, 0xc>(v);
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// exp(x) for the arguments of the Newton / P-matrix tables (x = lambda r t <= 0, |x| < 1e7):
//   x = n ln2/64 + r,  |r| <= ln2/128;  e^x = 2^(n >> 6) * 2^((n & 63)/64) * e^r
// e2t = the 64-entry table 2^(j/64) (LDS), e^r - 1 by a degree-6 Taylor polynomial (truncation
// 1e-19 relative), result T + T (e^r - 1) with one fma: ~1 ulp, 16 vector instructions where the
// general-purpose exp() (overflow / NaN handling, degree-11 polynomial) takes ~40.
__device__ __forceinline__ double exp_tab(double x, const double* e2t) {
  const double n = __builtin_rint(x * 92.33248261689366);
  double r = fma(n, -0.010830424696249145, x);      // ln2 / 64, high part
  r = fma(n, -3.623510646634843e-19, r);            //            low part
  const int ni = (int)n;
  const double T = e2t[ni & 63];
  double q = fma(r, 0.001388888888888889, 0.008333333333333333);
  q = fma(q, r, 0.041666666666666664);
  q = fma(q, r, 0.16666666666666666);
  q = fma(q, r, 0.5);
  q = fma(q, r, 1.0);
  q *= r;                                            // e^r - 1
  return __builtin_ldexp(fma(T, q, T), ni >> 6);     // underflows to 0 like exp()
}

}  // namespace epa_wave
