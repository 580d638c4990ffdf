// ds_add_f64 throughput on gfx950 by address pattern (measurement aid for the zones kernel's LDS image; not part of
// the product).  One wave per SIMD (256-thread blocks), every wave 2 return-less adds per iteration into its own
// 512-double image, optionally with V fp64 fma between them (the region-2 body has ~23).
//   hipcc --offload-arch=gfx950 -O3 lds_atomic.hip -o lds_atomic && ./lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// PATTERN 0: 64 consecutive doubles.  1: 8 rows of 8, row r starts at r (neighbouring lines, 1 point apart: up to
// 8 lanes on one address).  2: 8 rows of 8, row r starts at 9 r (disjoint windows).  3: 16 groups of 4, group g at g.
// 7-11: rows on 8-ALIGNED windows (the candidate layout for the zones kernel's region 2), see below.
// 4: 8 rows of 8, row r starts at 3 r.  5: all lanes one address.  6: rows at 8 r + (r & 1) (disjoint, unaligned)
template <int PATTERN, int V>
__global__ __launch_bounds__(256) void k(double *out, int iters) {
  __shared__ double img[4][2][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double *a = img[wave][0], *e = img[wave][1];
  for (int i = lane; i < 512; i += 64) a[i] = e[i] = 0.;
  const int row = lane >> 3, col = lane & 7;
  int idx;
  if (PATTERN == 0) idx = lane;
  else if (PATTERN == 1) idx = row + col;
  else if (PATTERN == 2) idx = 9 * row + col;
  else if (PATTERN == 3) idx = (lane >> 2) + (lane & 3);
  else if (PATTERN == 4) idx = 3 * row + col;
  else if (PATTERN == 5) idx = 0;
  else if (PATTERN == 6) idx = 8 * row + (row & 1) + col;
  else if (PATTERN == 7) idx = 8 * ((row * 3) & 7) + col;                  // aligned windows, a permutation of 64 consecutive
  else if (PATTERN == 8) idx = 8 * (((row * 3) & 7) + (row >> 2)) + col;    // aligned windows, rows 4..7 one window on: two pairs share a window
  else if (PATTERN == 9) idx = 8 * (((row * 3) & 7) + 2 * (row & 1)) + col; // aligned, spread over 10 windows, some shared
  else if (PATTERN == 10) idx = 8 * (row >> 1) + col;                       // aligned, pairs of rows on one window (2 lanes per address)
  else idx = 8 * (row >> 2) + col;                                          // aligned, four rows per window
  double v = 1.0 + lane * 1e-3, w = 0.5;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < V; ++q) v = fma(v, 1.0000001, 1e-9);
    const int at = idx + ((it * 8) & 255);
    atomicAdd(&a[at], v);
    atomicAdd(&e[at], v * w);
  }
  __syncthreads();
  double s = 0;
  for (int i = lane; i < 512; i += 64) s += a[i] + e[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int P, int V>
int run(const char *name, double *d_out) {
  const int blocks = 256 * 4, iters = 20000;
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<P, V>), dim3(blocks), dim3(256), 0, 0, d_out, 100);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<P, V>), dim3(blocks), dim3(256), 0, 0, d_out, iters);
  CHK(hipEventRecord(e1));
  CHK(hipEventSynchronize(e1));
  float ms;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  // 4 blocks per CU in sequence or together (LDS 16 KB each): per CU blocks*iters*4 waves*2 adds
  const double adds_per_cu = (double)blocks / 256 * iters * 4 * 2;
  printf("%-46s V=%2d  %7.3f ms  %6.1f ns per wave-add per CU  (%.1f clk @2.4GHz)\n", name, V, ms, ms * 1e6 / adds_per_cu,
         ms * 1e6 / adds_per_cu * 2.4);
  return 0;
}

int main() {
  double *d_out;
  CHK(hipMalloc(&d_out, sizeof(double) * 256 * 4 * 256));
  run<0, 0>("64 consecutive", d_out);
  run<1, 0>("8 rows of 8, rows 1 apart (same-address x8)", d_out);
  run<4, 0>("8 rows of 8, rows 3 apart", d_out);
  run<2, 0>("8 rows of 8, rows 9 apart (disjoint)", d_out);
  run<6, 0>("8 rows of 8, rows 8 + (r&1) apart", d_out);
  run<3, 0>("16 groups of 4, groups 1 apart", d_out);
  run<5, 0>("all lanes one address", d_out);
  run<7, 0>("aligned windows, permuted consecutive", d_out);
  run<8, 0>("aligned windows, half shifted one window", d_out);
  run<9, 0>("aligned windows, odd rows two windows on", d_out);
  run<10, 0>("aligned windows, two rows per window", d_out);
  run<11, 0>("aligned windows, four rows per window", d_out);
  run<7, 23>("aligned windows, permuted consecutive", d_out);
  run<8, 23>("aligned windows, half shifted one window", d_out);
  run<0, 23>("64 consecutive", d_out);
  run<1, 23>("8 rows of 8, rows 1 apart", d_out);
  run<2, 23>("8 rows of 8, rows 9 apart", d_out);
  run<3, 23>("16 groups of 4, groups 1 apart", d_out);
  run<1, 90>("8 rows of 8, rows 1 apart", d_out);
  run<2, 90>("8 rows of 8, rows 9 apart", d_out);
  return 0;
}
