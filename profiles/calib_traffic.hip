// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths this build
// uses (the microarch guide calibrates only 16-B-per-lane streaming reads: FETCH_SIZE reports half
// of them).  Each kernel moves a KNOWN number of bytes through HBM, coalesced, once:
//   k_read8   global_load_dwordx2  (8 B per lane: the thorough kernels' operand loads)
//   k_read16  global_load_dwordx4  (16 B per lane: the preplacement slice loads)
//   k_write8  global_store_dwordx2 (8 B per lane: results, table bursts)
//   k_write16 global_store_dwordx4
// Run:  hipcc --offload-arch=gfx950 -O3 profiles/calib_traffic.hip -o /tmp/calib && (cd /tmp;
//       rocprofv3 --pmc FETCH_SIZE --output-format csv -d out1 -- /tmp/calib; same with WRITE_SIZE)
// profiles/run_calib.sh does that and prints counter / known-bytes ratios.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_read8(const double* __restrict__ p, size_t n, double* out) {
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 1.2345e300) out[0] = acc;
}
__global__ void k_read16(const double2* __restrict__ p, size_t n, double* out) {
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double2 v = p[i];
    acc += v.x + v.y;
  }
  if (acc == 1.2345e300) out[0] = acc;
}
__global__ void k_write8(double* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (double)i;
}
__global__ void k_write16(double2* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = make_double2((double)i, 1.0);
}

int main() {
  const size_t bytes = (size_t)2 << 30;   // 2 GiB: 8 x the Infinity Cache
  double *a, *out;
  (void)hipMalloc(&a, bytes);
  (void)hipMalloc(&out, 64);
  (void)hipMemset(a, 0, bytes);
  const size_t n8 = bytes / 8, n16 = bytes / 16;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_read8, dim3(4096), dim3(256), 0, 0, a, n8, out);
    hipLaunchKernelGGL(k_read16, dim3(4096), dim3(256), 0, 0, (const double2*)a, n16, out);
    hipLaunchKernelGGL(k_write8, dim3(4096), dim3(256), 0, 0, a, n8);
    hipLaunchKernelGGL(k_write16, dim3(4096), dim3(256), 0, 0, (double2*)a, n16);
  }
  (void)hipDeviceSynchronize();
  printf("bytes per kernel: %zu\n", bytes);
  return 0;
}
