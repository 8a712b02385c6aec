// Do fp64 MFMA and fp64 VALU FMA instructions of DIFFERENT waves on one SIMD overlap on gfx950?
// (profiles/mfma_overlap.hip of round 1 mixed both in one instruction stream and under-measured the
// MFMA ceiling.)  256-thread workgroups, 8 waves per SIMD; a wave runs either a pure v_fma_f64 loop
// (32 independent chains per lane) or a pure v_mfma_f64_4x4x4_4b_f64 loop (20 independent
// accumulators).  MODE 0: all waves FMA; 1: all waves MFMA; 2: even waves FMA, odd waves MFMA, each
// doing the SAME per-wave work as in the pure runs.  If the two pipes are independent, mode 2 takes
// about max(t0, t1) / 2... (each kind has half the waves), i.e. clearly less than (t0 + t1) / 2.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void __launch_bounds__(256) k(double* out, int iters) {
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = MODE == 1 || (MODE == 2 && (wave & 1));
  double r = 0.0;
  if (!do_mfma) {
    double a = threadIdx.x * 1e-3 + 1.0, b = 0.999999;
    double c[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) c[i] = 0.01 * i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 32; ++i) c[i] = fma(a, b, c[i]);
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) r += c[i];
  } else {
    double a = threadIdx.x * 1e-3 + 1.0, b = 0.999999;
    double acc[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) acc[i] = 0.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 20; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 20; ++i) r += acc[i];
  }
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

int main() {
  double* d;
  (void)hipMalloc(&d, 8 * 256 * 2048);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 8000, grid = 2048;
  auto run = [&](auto kern, const char* name, double fma_waves, double mfma_waves) {
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, 50);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double fl = fma_waves * iters * 128.0 * 64 * 2 + mfma_waves * iters * 20.0 * 512;
    printf("%-34s %8.3f ms  %7.2f TFLOP/s\n", name, ms, fl / ms / 1e9);
  };
  const double W = (double)grid * 4;
  run(k<0>, "all waves v_fma_f64", W, 0);
  run(k<1>, "all waves v_mfma_f64_4x4x4_4b", 0, W);
  run(k<2>, "even waves FMA, odd waves MFMA", W / 2, W / 2);
  return 0;
}
