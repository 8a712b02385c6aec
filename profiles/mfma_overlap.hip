// Does the fp64 MFMA pipe run concurrently with fp64 VALU FMAs on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(256) k(double* out, int iters) {
  double a = threadIdx.x * 1e-3 + 1.0, b = 1.0000001, c0 = 0.5, c1 = 0.25, c2 = 0.125, c3 = 0.0625;
  double c4 = 0.1, c5 = 0.2, c6 = 0.3, c7 = 0.4;
  d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  for (int i = 0; i < iters; ++i) {
    if (MODE & 1) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        c0 = fma(a, b, c0); c1 = fma(a, b, c1); c2 = fma(a, b, c2); c3 = fma(a, b, c3);
        c4 = fma(a, b, c4); c5 = fma(a, b, c5); c6 = fma(a, b, c6); c7 = fma(a, b, c7);
      }
    }
    if (MODE & 2) {
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc1, 0, 0, 0);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[3];
}
int main() {
  double* d; (void)hipMalloc(&d, 8 * 256 * 2048);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, grid = 2048;  // 8 waves per SIMD... 2048 blocks x 4 waves
  auto run = [&](auto kern, const char* name, double fma_per_iter_thread, double mfma_flops_per_iter_wave) {
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, iters); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)grid * 256 * iters * fma_per_iter_thread * 2 + (double)grid * 4 * iters * mfma_flops_per_iter_wave;
    printf("%-10s %8.3f ms  %7.2f TFLOP/s\n", name, ms, fl / ms / 1e9);
  };
  run(k<1>, "valu", 32, 0);
  run(k<2>, "mfma", 0, 2 * 2.0 * 16 * 16 * 4);
  run(k<3>, "both", 32, 2 * 2.0 * 16 * 16 * 4);
  return 0;
}
