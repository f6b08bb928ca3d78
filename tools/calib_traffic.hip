// Known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 in the tick kernels' OWN access pattern
// (MI355X guide: "Other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
//   calib_stream16   16 B per lane, coalesced copy                      (the guide's reference case: FETCH_SIZE reads 1/2, WRITE_SIZE exact)
//   calib_stream8    8 B per lane, coalesced copy                        (one double per lane)
//   calib_tick_rows  the packed tick kernel's global accesses, address for address: per instance (16 lanes) q [27] as 16 + 11 doubles,
//                    3 doubles out of a 15-double row twice (gripper target, previous), 4 doubles (box centre); stores qdot [26] as 16 + 10
//                    doubles + status + iters. Every 64-byte line of the five arrays is touched, so the bytes that must cross the fabric are
//                    the arrays' sizes: 216 + 120 + 120 + 32 = 488 B read and 208 + 4 + 4 = 216 B written per instance.
// Build:  hipcc --offload-arch=gfx950 -O2 tools/calib_traffic.hip -o tools/build/calib_traffic
// Run under rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE): tools/calib_traffic.sh
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct d2 { double x, y; };
__global__ void calib_stream16(const d2* __restrict__ in, d2* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}
__global__ void calib_stream8(const double* __restrict__ in, double* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}
__global__ void __launch_bounds__(64) calib_tick_rows(const double* __restrict__ q, const double* __restrict__ eet, const double* __restrict__ eep,
                                                      const double* __restrict__ box, double* __restrict__ qdot, int* __restrict__ status,
                                                      int* __restrict__ iters, int B) {
  const int lane = threadIdx.x, r = lane >> 4, s = lane & 15;
  const int b = 4 * blockIdx.x + r;
  if (b >= B) return;
  const double* qg = q + (size_t)b * 27;
  const double q0 = qg[s], q1 = (16 + s < 27) ? qg[16 + s] : 0.0;
  double ex = 0.0;
  if (s < 3) ex = eet[(size_t)b * 15 + 12 + s];
  else if (s < 6) ex = eep[(size_t)b * 15 + 12 + (s - 3)];
  else if (s < 10) ex = box[(size_t)b * 4 + (s - 6)];
  const double v = q0 + q1 + ex;
  double* qo = qdot + (size_t)b * 26;
  qo[s] = v;
  if (16 + s < 26) qo[16 + s] = v + 1.0;
  if (s == 0) { status[b] = 0; iters[b] = 16; }
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 65536, reps = argc > 2 ? atoi(argv[2]) : 5;
  const size_t n8 = (size_t)B * 61;          // doubles: about the tick's 488 B per instance
  double *a, *b_, *q, *eet, *eep, *box, *qdot;
  int *st, *it;
  CHECK(hipMalloc(&a, n8 * 8)); CHECK(hipMalloc(&b_, n8 * 8));
  CHECK(hipMalloc(&q, (size_t)B * 27 * 8)); CHECK(hipMalloc(&eet, (size_t)B * 15 * 8)); CHECK(hipMalloc(&eep, (size_t)B * 15 * 8));
  CHECK(hipMalloc(&box, (size_t)B * 4 * 8)); CHECK(hipMalloc(&qdot, (size_t)B * 26 * 8)); CHECK(hipMalloc(&st, (size_t)B * 4)); CHECK(hipMalloc(&it, (size_t)B * 4));
  CHECK(hipMemset(a, 0, n8 * 8)); CHECK(hipMemset(q, 0, (size_t)B * 27 * 8)); CHECK(hipMemset(eet, 0, (size_t)B * 15 * 8));
  CHECK(hipMemset(eep, 0, (size_t)B * 15 * 8)); CHECK(hipMemset(box, 0, (size_t)B * 4 * 8));
  for (int k = 0; k < reps; ++k) {
    hipLaunchKernelGGL(calib_stream16, dim3((unsigned)((n8 / 2 + 255) / 256)), dim3(256), 0, 0, (const d2*)a, (d2*)b_, n8 / 2);
    hipLaunchKernelGGL(calib_stream8, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, 0, a, b_, n8);
    hipLaunchKernelGGL(calib_tick_rows, dim3((B + 3) / 4), dim3(64), 0, 0, q, eet, eep, box, qdot, st, it, B);
  }
  CHECK(hipDeviceSynchronize());
  printf("{\"B\": %d, \"reps\": %d, \"calib_stream16\": {\"read_bytes\": %zu, \"write_bytes\": %zu}, \"calib_stream8\": {\"read_bytes\": %zu, \"write_bytes\": %zu}, "
         "\"calib_tick_rows\": {\"read_bytes\": %zu, \"write_bytes\": %zu}}\n", B, reps, n8 / 2 * 16, n8 / 2 * 16, n8 * 8, n8 * 8,
         (size_t)B * 488, (size_t)B * 216);
  return 0;
}
