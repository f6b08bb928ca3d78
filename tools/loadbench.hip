// loadbench — what the load phase of the packed QP kernel costs by itself: B problems of (A [m][n], C [p][n]) doubles, one wavefront per PER problems,
// every lane requests its ~NL elements up front, sums them, stores one value. Variants: dynamic LDS bytes (occupancy), 8- vs 16-byte loads.
//   hipcc --offload-arch=gfx950 -O3 loadbench.hip -o build/loadbench && build/loadbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int NL, int W>
__global__ void __launch_bounds__(64) k_load(const double* __restrict__ src, double* __restrict__ out, const size_t per_wave) {
  extern __shared__ double lds[];
  const double* p = src + (size_t)blockIdx.x * per_wave;
  double acc = 0.0;
  if (W == 1) {
    double v[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) v[k] = ((size_t)(k * 64 + threadIdx.x) < per_wave) ? p[k * 64 + threadIdx.x] : 0.0;
#pragma unroll
    for (int k = 0; k < NL; ++k) acc += v[k];
  } else {
    double2 v[NL / 2];
#pragma unroll
    for (int k = 0; k < NL / 2; ++k) v[k] = ((size_t)(k * 128 + 2 * threadIdx.x) < per_wave) ? *reinterpret_cast<const double2*>(p + k * 128 + 2 * threadIdx.x) : double2{0, 0};
#pragma unroll
    for (int k = 0; k < NL / 2; ++k) acc += v[k].x + v[k].y;
  }
  if (threadIdx.x == 0) lds[0] = acc;
  out[(size_t)blockIdx.x * 64 + threadIdx.x] = acc + lds[0];
}
template <int NL, int W>
static void run(const double* src, double* out, int grid, size_t per_wave, size_t ldsb, const char* what) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_load<NL, W>), dim3(grid), dim3(64), ldsb, 0, src, out, per_wave);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k_load<NL, W>), dim3(grid), dim3(64), ldsb, 0, src, out, per_wave);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  printf("%-44s grid %6d  LDS %6zu B  %8.1f us  %7.1f GB/s\n", what, grid, ldsb, ms * 1e3, grid * per_wave * 8 / (ms * 1e-3) / 1e9);
}
int main() {
  const int grid = 16384; const size_t per_wave = 2 * (32 * 26 + 16 * 26);   // two problems' A and C
  double *src, *out; hipMalloc(&src, grid * per_wave * 8); hipMalloc(&out, (size_t)grid * 64 * 8);
  hipMemset(src, 0, grid * per_wave * 8);
  for (size_t l : {(size_t)0, (size_t)20480, (size_t)32512, (size_t)65536}) {
    run<40, 1>(src, out, grid, per_wave, l, "40 x 8-byte loads per lane");
    run<40, 2>(src, out, grid, per_wave, l, "20 x 16-byte loads per lane");
  }
  return 0;
}
