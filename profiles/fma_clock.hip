// What clock does the chip hold under a sustained fp64 FMA stream, and how many shader cycles does one
// wave-wide v_fma_f64 take?  (is the 0.805 between a bare FMA stream and the
// spec-sheet peak a power-capped clock or an issue-rate limit?)
//
// Every wave stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around the loop; the host
// prints, per configuration: wall ms (HIP events), TFLOP/s, the shader clock the waves saw
// (cycles / 10 ns ticks x 100 MHz; median over waves) and shader cycles per wave-level FMA and SIMD.
//   hipcc --offload-arch=gfx950 -O3 profiles/fma_clock.hip -o /tmp/fma_clock && /tmp/fma_clock
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

template <int CHAINS>
__global__ void __launch_bounds__(256) k_fma(double* out, unsigned long long* stamps, int iters) {
  double a = threadIdx.x * 1e-3 + 1.0, b = 1.0000001;
  double c[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) c[i] = 0.125 * i;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 32 / CHAINS; ++u) {
#pragma unroll
      for (int j = 0; j < CHAINS; ++j) c[j] = fma(a, b, c[j]);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) s += c[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    stamps[2 * w] = t1 - t0;
    stamps[2 * w + 1] = r1 - r0;
  }
}

int main() {
  const int max_grid = 256 * 8;
  double* d;
  unsigned long long* st;
  (void)hipMalloc(&d, sizeof(double) * 256 * max_grid);
  (void)hipMalloc(&st, 16 * 4 * max_grid);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  std::vector<unsigned long long> h(2 * 4 * max_grid);
  printf("# waves/SIMD  chains  iters     wall_ms   TFLOP/s  sclk_MHz(p50)  sclk_MHz(min..max)  cycles_per_wave_FMA_per_SIMD\n");
  auto run = [&](auto kern, int chains, int waves_per_simd, int iters) {
    const int grid = 256 * waves_per_simd;   // 256 CUs x 4 SIMDs x waves_per_simd waves = grid x 4 waves
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, st, 64);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, st, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h.data(), st, 16 * 4 * grid, hipMemcpyDeviceToHost);
    std::vector<double> mhz, cyc;
    for (int w = 0; w < 4 * grid; ++w) {
      if (!h[2 * w + 1]) continue;
      mhz.push_back(100.0 * (double)h[2 * w] / (double)h[2 * w + 1]);
      cyc.push_back((double)h[2 * w]);
    }
    std::sort(mhz.begin(), mhz.end());
    std::sort(cyc.begin(), cyc.end());
    const double fl = (double)grid * 256 * iters * 32.0 * 2;
    // a SIMD executes waves_per_simd waves x 32 x iters wave-level FMAs during the launch: wall time x the clock the
    // waves saw (a single wave's own cycle count is shorter than the launch when the waves of a SIMD do not all start together)
    const double cpf = (double)ms * 1e-3 * mhz[mhz.size() / 2] * 1e6 / ((double)waves_per_simd * 32.0 * iters);
    (void)cyc;
    printf("  %9d  %6d  %7d  %9.3f  %8.2f  %13.0f  %8.0f..%-8.0f  %10.3f\n", waves_per_simd, chains, iters, ms, fl / ms / 1e9,
           mhz[mhz.size() / 2], mhz.front(), mhz.back(), cpf);
  };
  for (int iters : {2000, 20000, 200000}) {
    run(k_fma<8>, 8, 1, iters);
    run(k_fma<8>, 8, 2, iters);
    run(k_fma<8>, 8, 4, iters);
    run(k_fma<8>, 8, 8, iters);
  }
  run(k_fma<16>, 16, 2, 20000);
  run(k_fma<4>, 4, 2, 20000);
  run(k_fma<2>, 2, 2, 20000);
  return 0;
}
