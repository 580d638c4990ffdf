// A pure v_fma_f64 stream for a few seconds (measurement aid, not part of the product): what the chip sustains in dense
// fp64 under its power management, to set the zones kernel's Joules per flop against (tools/fma_energy.py samples the power).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off fma_energy.hip -o fma_energy && ./fma_energy [seconds]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void fma_kernel(double *out, int iters, double seed) {
  double a[8];
  const double t = seed + threadIdx.x * 1e-3;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = t + i;
  const double b = 1.0000001, c = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = fma(a[i], b, c);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main(int argc, char **argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
  const int blocks = 256 * 16, iters = 20000;
  double *out = nullptr;
  if (hipMalloc(&out, sizeof(double) * blocks * 256) != hipSuccess) return 1;
  const double flop_per_launch = 2.0 * 8 * iters * (double)blocks * 256;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(fma_kernel, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0);
  hipDeviceSynchronize();
  const auto t0 = std::chrono::steady_clock::now();
  long n = 0;
  double dt = 0;
  do {
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(fma_kernel, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0);
    hipDeviceSynchronize();
    n += 20;
    dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  } while (dt < seconds);
  printf("fma stream: %.3f s, %ld launches, %.2f TFLOP/s\n", dt, n, flop_per_launch * n / dt / 1e12);
  hipFree(out);
  return 0;
}
