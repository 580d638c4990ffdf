// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of the coefficient kernels
// (MI355X_MICROARCH.md: "other access widths are uncalibrated: calibrate on a known byte count in your own access
// pattern").  Every kernel reads a 1 GiB buffer exactly once (far beyond the 256 MiB Infinity Cache), so the true HBM
// bytes are 1 GiB per launch; run under   rocprofv3 --pmc FETCH_SIZE -- ./fetch_calib   (and WRITE_SIZE).
//   k_lane8      8 B per lane, coalesced                      (output / coefficient rows, radiance kernels)
//   k_lane16     16 B per lane, coalesced                     (the guide's calibrated case: reports 1/2)
//   k_rec80      per-lane 8-byte fields of 80-byte records, eight records per wave instruction (the row walks)
//   k_scalar64   64 B scalar loads (s_load_dwordx16) of wave-uniform addresses (exact-mode kernels)
//   k_write8     8 B per lane stores of 1 GiB
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr size_t kBytes = (size_t)1 << 30;

__global__ void k_lane8(const double *__restrict__ p, double *out, size_t n) {
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
  if (s == 12345.678) out[0] = s;
}
__global__ void k_lane16(const double2 *__restrict__ p, double *out, size_t n) {
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const double2 v = p[i]; s += v.x + v.y; }
  if (s == 12345.678) out[0] = s;
}
struct Rec { double f[10]; };
__global__ void k_rec80(const Rec *__restrict__ p, double *out, size_t n_rec) {
  // a wave takes eight consecutive records at a time: row r = lane / 8 reads record base + r, every lane of the row
  // the same ten fields one after the other (as the zones / wings rows read a FastRec)
  const int lane = threadIdx.x & 63, row = lane >> 3;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
  double s = 0;
  for (size_t b = wave * 8; b < n_rec; b += n_waves * 8) {
    const Rec &r = p[b + row];
#pragma unroll
    for (int q = 0; q < 10; ++q) s += r.f[q];
  }
  if (s == 12345.678) out[0] = s;
}
struct __attribute__((aligned(64))) Rec64 { double f[8]; };
__global__ void k_scalar64(const Rec64 *__restrict__ p, double *out, size_t n_rec) {
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
  double s = 0;
  for (size_t b = __builtin_amdgcn_readfirstlane((int)wave); b < n_rec; b += n_waves) {
    const Rec64 r = p[b]; // wave-uniform address: one s_load_dwordx16
#pragma unroll
    for (int q = 0; q < 8; ++q) s += r.f[q];
  }
  if (s == 12345.678) out[0] = s;
}
__global__ void k_write8(double *__restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (double)i;
}
int main() {
  void *buf; double *out;
  CHK(hipMalloc(&buf, kBytes)); CHK(hipMalloc(&out, 64));
  CHK(hipMemset(buf, 0, kBytes));
  const dim3 g(256 * 16), b(256);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_lane8, g, b, 0, 0, (const double *)buf, out, kBytes / 8);
    hipLaunchKernelGGL(k_lane16, g, b, 0, 0, (const double2 *)buf, out, kBytes / 16);
    hipLaunchKernelGGL(k_rec80, g, b, 0, 0, (const Rec *)buf, out, kBytes / 80 / 8 * 8);
    hipLaunchKernelGGL(k_scalar64, g, b, 0, 0, (const Rec64 *)buf, out, kBytes / 64);
    hipLaunchKernelGGL(k_write8, g, b, 0, 0, (double *)buf, kBytes / 8);
  }
  CHK(hipDeviceSynchronize());
  printf("each kernel touched %.6f GB per launch\n", kBytes / 1e9);
  return 0;
}
