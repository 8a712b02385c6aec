// Go / no-go microbenchmark for moving k_thorough_dna's 4 x 4 phase products onto the matrix cores
// (VERDICT round 3, item 1): what does an instruction of another kind cost when it is issued BETWEEN
// fp64 matrix instructions of the same wave, with two waves per SIMD as in the kernel?
//   per loop iteration and wave: 8 independent v_mfma_f64_4x4x4_4b_f64 accumulation chains, and after
//   every MFMA V filler instructions of one kind.  Time per iteration against V tells whether the kind
//   issues in the MFMA's shadow (flat) or behind it (slope = its issue cost).
// build: hipcc --offload-arch=gfx950 -O3 profiles/mfma_fill_microbench.hip -o /tmp/mfma_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

enum Kind { ADD_U32 = 0, SWAP32 = 1, MUL_F64 = 2, MAX_I32 = 3, CNDMASK = 4, FMA_F64 = 5, DS_READ = 6, SWAP16 = 7, DPP_MOV = 8 };

template <int KIND, int V, bool MFMA>
__global__ void __launch_bounds__(512) k(double* out, int iters) {
  __shared__ double lds[1024];
  lds[threadIdx.x] = threadIdx.x;
  lds[threadIdx.x + 512] = 1.0;
  __syncthreads();
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  double c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int u[4] = {(int)threadIdx.x, 1, 2, 3};
  double d[4] = {1.0, 1.0, 1.0, 1.0};
  const int one = 1;
  const int laddr = (threadIdx.x & 63) * 8;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MFMA) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a), "v"(b));
#pragma unroll
      for (int f = 0; f < V; ++f) {
        const int j = (i * V + f) & 3;
        if (KIND == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[j]) : "v"(one));
        if (KIND == SWAP32) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u[j]), "+v"(u[(j + 1) & 3]));
        if (KIND == SWAP16) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(u[j]), "+v"(u[(j + 1) & 3]));
        if (KIND == MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[j]) : "v"(b));
        if (KIND == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[j]) : "v"(b));
        if (KIND == MAX_I32) asm volatile("v_max_i32 %0, %0, %1" : "+v"(u[j]) : "v"(one));
        if (KIND == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[j]) : "v"(one));
        if (KIND == DPP_MOV) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[j]) : "v"(u[(j + 1) & 3]));
        if (KIND == DS_READ) asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(3)" : "=v"(d[j]) : "v"(laddr));
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  double s = 0;
  for (int i = 0; i < 8; ++i) s += c[i];
  for (int i = 0; i < 4; ++i) s += d[i] + u[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int KIND, int V, bool MFMA>
double run(double* d_out, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, V, MFMA>), dim3(256), dim3(512), 0, 0, d_out, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, V, MFMA>), dim3(256), dim3(512), 0, 0, d_out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

template <int KIND>
void sweep(const char* name, double* d_out, int iters, double clk_ghz) {
  const double t0 = run<KIND, 0, true>(d_out, iters);
  const double ms[5] = {t0, run<KIND, 1, true>(d_out, iters), run<KIND, 2, true>(d_out, iters), run<KIND, 3, true>(d_out, iters),
                        run<KIND, 4, true>(d_out, iters)};
  const double alone = run<KIND, 4, false>(d_out, iters);
  // cycles per MFMA slot (two waves per SIMD share the SIMD: per-SIMD cycles per [MFMA + V fillers] pair of waves)
  printf("%-10s", name);
  for (int v = 0; v < 5; ++v) printf("  V=%d %7.3f ms (%5.1f clk/slot)", v, ms[v], ms[v] * 1e-3 * clk_ghz * 1e9 / (iters * 8.0 * 2.0));
  printf("   fillers alone (V=4, no MFMA) %7.3f ms (%5.1f clk per 4)\n", alone, alone * 1e-3 * clk_ghz * 1e9 / (iters * 8.0 * 2.0));
}

int main() {
  double* d_out;
  hipMalloc(&d_out, sizeof(double) * 256 * 512);
  const int iters = 20000;
  const double clk = 2.4;
  printf("# 256 workgroups x 8 waves (2 per SIMD), %d iterations x 8 MFMA slots per wave; clk/slot = SIMD clocks per\n"
         "# (MFMA + V fillers) of ONE wave at an assumed %.1f GHz (two waves share the SIMD: a bare MFMA stream = 16)\n", iters, clk);
  sweep<ADD_U32>("v_add_u32", d_out, iters, clk);
  sweep<MAX_I32>("v_max_i32", d_out, iters, clk);
  sweep<CNDMASK>("v_cndmask", d_out, iters, clk);
  sweep<DPP_MOV>("v_mov_dpp", d_out, iters, clk);
  sweep<SWAP32>("permlane32", d_out, iters, clk);
  sweep<SWAP16>("permlane16", d_out, iters, clk);
  sweep<MUL_F64>("v_mul_f64", d_out, iters, clk);
  sweep<FMA_F64>("v_fma_f64", d_out, iters, clk);
  sweep<DS_READ>("ds_read_b64", d_out, iters, clk);
  return 0;
}
