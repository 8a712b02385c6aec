// A/B of the 20-state inner contraction of k_thorough_aa on gfx950, VALU vs MFMA, in the BEST case
// for the matrix cores: the operand vectors already sit in the register layout each variant wants
// (no LDS transposes, no cross-lane shuffles, which a real MFMA variant of the kernel would need on
// top: the kernel's lane = site layout is the VALU layout).
//
// Work item (what one wavefront of k_thorough_aa does per rate category and 64-site pass):
//     Y[20][64] = U[20][20] . V[20][64]        (U wave-uniform, V per site), 51 200 flop
//   valu      per lane = site: 400 v_fma_f64, U through the scalar cache (SGPR operands) -- the
//             kernel's own form (thorough_aa.hip matvec20)
//   mfma16    v_mfma_f64_16x16x4_f64: 20 rows padded to 2 x 16, 5 k-steps, 4 blocks of 16 sites
//             = 40 instructions, 81 920 flop issued for 51 200 useful
//   mfma4     v_mfma_f64_4x4x4_4b_f64: 4 x 4 tiles (20 = 5 x 4, no padding): 5 row tiles x 5 k-steps
//             x 16 site tiles / 4 blocks per instruction = 100 instructions, 51 200 flop issued
// Every variant feeds its result back as the next input (a real dependency chain per accumulator,
// independent accumulators in between) and runs 8 waves per SIMD.
// Build / run:  hipcc --offload-arch=gfx950 -O3 profiles/aa_contraction_ab.hip -o /tmp/aa_ab && /tmp/aa_ab
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) double* ConstD;

__global__ void __launch_bounds__(256) k_valu(const double* __restrict__ Ug, double* out, int iters) {
  ConstD U = (ConstD)Ug;
  double v[20], y[20];
  for (int x = 0; x < 20; ++x) v[x] = 1.0 + 1e-3 * (threadIdx.x + x);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 20; ++i) {
      double acc = U[i * 20] * v[0];
#pragma unroll
      for (int x = 1; x < 20; ++x) acc = fma(U[i * 20 + x], v[x], acc);
      y[i] = acc;
    }
#pragma unroll
    for (int x = 0; x < 20; ++x) v[x] = y[x] * 0x1p-4;
  }
  double s = 0;
  for (int x = 0; x < 20; ++x) s += v[x];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// A operand of tile (rt, t): lane l holds U[16 rt + l % 16][4 t + l / 16] (0 for padded rows)
__global__ void __launch_bounds__(256) k_mfma16(const double* __restrict__ Ug, double* out, int iters) {
  const int l = threadIdx.x & 63;
  double A[2][5];
  for (int rt = 0; rt < 2; ++rt)
    for (int t = 0; t < 5; ++t) {
      const int r = 16 * rt + (l & 15);
      A[rt][t] = r < 20 ? Ug[r * 20 + 4 * t + (l >> 4)] : 0.0;
    }
  // B operand of (block j, k-step t): lane l holds V[4 t + l / 16][16 j + l % 16]
  double Bv[4][5];
  for (int j = 0; j < 4; ++j)
    for (int t = 0; t < 5; ++t) Bv[j][t] = 1.0 + 1e-3 * (l + j + t);
  for (int it = 0; it < iters; ++it) {
    d4 acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        d4 a = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 5; ++t) a = __builtin_amdgcn_mfma_f64_16x16x4f64(A[rt][t], Bv[j][t], a, 0, 0, 0);
        acc[j][rt] = a;
      }
    // feed back (layout conversion D -> B would be extra work in a real kernel; here: any dependency)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < 5; ++t) Bv[j][t] = (t < 4 ? acc[j][0][t & 3] : acc[j][1][0]) * 0x1p-4;
  }
  double s = 0;
  for (int j = 0; j < 4; ++j) for (int t = 0; t < 5; ++t) s += Bv[j][t];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// 4x4x4, 4 blocks: lane l: block l / 16, within the block A[i = l % 4][k = (l / 4) % 4], B[k][j = l % 4];
// D: one double per lane.  Instruction (rt, t, g): row tile rt, k-step t, site tiles 4 g .. 4 g + 3.
__global__ void __launch_bounds__(256) k_mfma4(const double* __restrict__ Ug, double* out, int iters) {
  const int l = threadIdx.x & 63;
  double A[5][5];
  for (int rt = 0; rt < 5; ++rt)
    for (int t = 0; t < 5; ++t) A[rt][t] = Ug[(4 * rt + (l & 3)) * 20 + 4 * t + ((l >> 2) & 3)];
  double Bv[4][5];   // [site-tile group g][k-step t]
  for (int g = 0; g < 4; ++g)
    for (int t = 0; t < 5; ++t) Bv[g][t] = 1.0 + 1e-3 * (l + g + t);
  for (int it = 0; it < iters; ++it) {
    double acc[4][5];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int rt = 0; rt < 5; ++rt) {
        double a = 0.0;
#pragma unroll
        for (int t = 0; t < 5; ++t) a = __builtin_amdgcn_mfma_f64_4x4x4f64(A[rt][t], Bv[g][t], a, 0, 0, 0);
        acc[g][rt] = a;
      }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int t = 0; t < 5; ++t) Bv[g][t] = acc[g][t] * 0x1p-4;
  }
  double s = 0;
  for (int g = 0; g < 4; ++g) for (int t = 0; t < 5; ++t) s += Bv[g][t];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  double *d, *U;
  (void)hipMalloc(&d, 8 * 256 * 2048);
  (void)hipMalloc(&U, 8 * 400);
  double hU[400];
  for (int i = 0; i < 400; ++i) hU[i] = (i % 21 == 0 ? 0.9 : 0.005) * ((i & 1) ? 1 : -1);
  (void)hipMemcpy(U, hU, sizeof(hU), hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 4000, grid = 2048;   // 2048 x 4 waves = 8 waves per SIMD on 256 CUs
  auto run = [&](auto kern, const char* name, double issued_flop_per_item) {
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, U, d, 50);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, U, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double items = (double)grid * 4 * iters;   // one 20x20 . 20x64 contraction per wave and iteration
    printf("%-8s %8.3f ms   %7.2f useful TFLOP/s   %7.2f issued TFLOP/s   %6.1f ns per contraction and wave\n", name, ms,
           items * 51200.0 / ms / 1e9, items * issued_flop_per_item / ms / 1e9, ms * 1e6 / iters);
  };
  run(k_valu, "valu", 51200.0);
  run(k_mfma16, "mfma16", 81920.0);
  run(k_mfma4, "mfma4", 51200.0);
  return 0;
}
