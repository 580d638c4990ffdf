// fp64 VALU instruction rates and v_rcp_f64 accuracy on gfx950 (measurement aid
// for sr_abscoeff_kernel's design; not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off fp64_rates.hip -o fp64_rates && ./fp64_rates
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(double *out, int iters, double seed) {
  double a[8];
  const double t = seed + threadIdx.x * 1e-3;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = t + i;
  const double b = 1.0000001, c = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) a[i] = fma(a[i], b, c);
      if (MODE == 1) a[i] = a[i] * b;
      if (MODE == 2) a[i] = a[i] + c;
      if (MODE == 3) a[i] = __builtin_amdgcn_rcp(a[i]);
      if (MODE == 4) a[i] = (double)(float)a[i];                    // cvt f64->f32->f64
      if (MODE == 5) a[i] = (double)__builtin_amdgcn_rcpf((float)a[i]); // f32 rcp seed path
      if (MODE == 6) a[i] = __builtin_amdgcn_rsq(a[i]);
      if (MODE == 7) a[i] = c / a[i];                                // IEEE divide
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the region-1 evaluation body, P points per lane, NR newton steps, SHARE points per reciprocal
template <int NR>
__device__ inline double rcpn(double d) {
  double r = __builtin_amdgcn_rcp(d);
#pragma unroll
  for (int i = 0; i < NR; ++i) { double e = fma(-d, r, 1.0); r = fma(r, e, r); }
  return r;
}
template <int P, int NR, int SHARE>
__global__ __launch_bounds__(256) void eval_kernel(double *out, int n_lines, const double *__restrict__ rec) {
  double acc_a[P], acc_e[P], flp[P];
#pragma unroll
  for (int p = 0; p < P; ++p) { acc_a[p] = 0; acc_e[p] = 0; flp[p] = (threadIdx.x & 63) + 64 * p; }
  for (int l = 0; l < n_lines; ++l) {
    const double *r = rec + (l & 255) * 8;
    const double xb = r[0], xs = r[1], a = r[2], b = r[3], cc = r[4], d = r[5], wa = r[6], we = r[7];
    double num[P], den[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const double x = fma(flp[p], xs, xb);
      const double x2 = x * x;
      num[p] = fma(x2, b, a);
      den[p] = fma(x2, fma(x2, 4.0, d), cc);
    }
    if (SHARE == 1) {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const double q = num[p] * rcpn<NR>(den[p]);
        acc_a[p] = fma(wa, q, acc_a[p]);
        acc_e[p] = fma(we, q, acc_e[p]);
      }
    } else if (SHARE == 2) {
#pragma unroll
      for (int p = 0; p < P; p += 2) {
        const double r12 = rcpn<NR>(den[p] * den[p + 1]);
        const double q0 = (num[p] * den[p + 1]) * r12, q1 = (num[p + 1] * den[p]) * r12;
        acc_a[p] = fma(wa, q0, acc_a[p]); acc_e[p] = fma(we, q0, acc_e[p]);
        acc_a[p + 1] = fma(wa, q1, acc_a[p + 1]); acc_e[p + 1] = fma(we, q1, acc_e[p + 1]);
      }
    } else {
#pragma unroll
      for (int p = 0; p < P; p += 4) {
        const double d01 = den[p] * den[p + 1], d23 = den[p + 2] * den[p + 3];
        const double r = rcpn<NR>(d01 * d23);
        const double r01 = r * d23, r23 = r * d01;
        const double q0 = (num[p] * den[p + 1]) * r01, q1 = (num[p + 1] * den[p]) * r01;
        const double q2 = (num[p + 2] * den[p + 3]) * r23, q3 = (num[p + 3] * den[p + 2]) * r23;
        acc_a[p] = fma(wa, q0, acc_a[p]); acc_e[p] = fma(we, q0, acc_e[p]);
        acc_a[p + 1] = fma(wa, q1, acc_a[p + 1]); acc_e[p + 1] = fma(we, q1, acc_e[p + 1]);
        acc_a[p + 2] = fma(wa, q2, acc_a[p + 2]); acc_e[p + 2] = fma(we, q2, acc_e[p + 2]);
        acc_a[p + 3] = fma(wa, q3, acc_a[p + 3]); acc_e[p + 3] = fma(we, q3, acc_e[p + 3]);
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int p = 0; p < P; ++p) s += acc_a[p] + acc_e[p];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void rcp_acc_kernel(const double *x, double *r0, double *r1, double *r2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { r0[i] = rcpn<0>(x[i]); r1[i] = rcpn<1>(x[i]); r2[i] = rcpn<2>(x[i]); }
}

template <class F>
float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

int main() {
  double *out; CHK(hipMalloc(&out, sizeof(double) * 256 * 8192));
  const int blocks = 256 * 8, iters = 4000;  // 8 blocks of 256 per CU = 8 waves/SIMD
  const char *names[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "cvt f64->f32->f64", "cvt+v_rcp_f32+cvt", "v_rsq_f64", "IEEE div f64"};
  float ms[8];
  ms[0] = timeit([&] { hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); });
  ms[1] = timeit([&] { hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); });
  ms[2] = timeit([&] { hipLaunchKernelGGL(rate_kernel<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); });
  ms[3] = timeit([&] { hipLaunchKernelGGL(rate_kernel<3>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); });
  ms[4] = timeit([&] { hipLaunchKernelGGL(rate_kernel<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); });
  ms[5] = timeit([&] { hipLaunchKernelGGL(rate_kernel<5>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); });
  ms[6] = timeit([&] { hipLaunchKernelGGL(rate_kernel<6>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); });
  ms[7] = timeit([&] { hipLaunchKernelGGL(rate_kernel<7>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); });
  const double nops = (double)blocks * 256 * iters * 8;
  for (int i = 0; i < 8; ++i)
    printf("%-22s %8.3f ms  %8.2f Gop/s  => %.2f cycles/wave-instr/SIMD @2.4GHz\n", names[i], ms[i], nops / ms[i] / 1e6,
           (ms[i] * 1e-3 * 2.4e9) / (nops / 64 / 1024));
  // eval bodies
  std::vector<double> hrec(256 * 8);
  for (int l = 0; l < 256; ++l) { double *r = &hrec[l * 8]; r[0] = 20 + l * 0.1; r[1] = 0.125; r[2] = 1e-3; r[3] = 2e-3; r[4] = 1.0; r[5] = -4.0; r[6] = 1e-20; r[7] = 1e-25; }
  double *rec; CHK(hipMalloc(&rec, sizeof(double) * 256 * 8));
  CHK(hipMemcpy(rec, hrec.data(), sizeof(double) * 256 * 8, hipMemcpyHostToDevice));
  const int nl = 4000;
#define EV(P, NR, SH) { float m = timeit([&] { hipLaunchKernelGGL((eval_kernel<P, NR, SH>), dim3(blocks), dim3(256), 0, 0, out, nl, rec); }); \
    double ev = (double)blocks * 256 * nl * P; printf("eval P=%d NR=%d SHARE=%d: %8.3f ms  %8.2f Geval/s  (%.1f cycles/wave-eval/SIMD)\n", P, NR, SH, m, ev / m / 1e6, (m * 1e-3 * 2.4e9) / (ev / 64 / 1024)); }
  EV(4, 2, 1) EV(4, 1, 1) EV(4, 0, 1) EV(4, 2, 2) EV(4, 1, 2) EV(4, 2, 4) EV(4, 1, 4) EV(8, 2, 1) EV(8, 2, 4) EV(8, 1, 4) EV(2, 2, 1) EV(2, 2, 2)
  // rcp accuracy
  const int n = 1 << 20;
  std::vector<double> hx(n), h0(n), h1(n), h2(n);
  unsigned long long s = 88172645463325252ULL;
  for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (s >> 11) * (1.0 / 9007199254740992.0); hx[i] = std::exp(u * 120 - 20) * (1 + u); }
  double *dx, *d0, *d1, *d2; CHK(hipMalloc(&dx, n * 8)); CHK(hipMalloc(&d0, n * 8)); CHK(hipMalloc(&d1, n * 8)); CHK(hipMalloc(&d2, n * 8));
  CHK(hipMemcpy(dx, hx.data(), n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(rcp_acc_kernel, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
  CHK(hipMemcpy(h0.data(), d0, n * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(h1.data(), d1, n * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(h2.data(), d2, n * 8, hipMemcpyDeviceToHost));
  double e0 = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < n; ++i) {
    long double ex = 1.0L / (long double)hx[i];
    e0 = std::fmax(e0, (double)fabsl(((long double)h0[i] - ex) / ex));
    e1 = std::fmax(e1, (double)fabsl(((long double)h1[i] - ex) / ex));
    e2 = std::fmax(e2, (double)fabsl(((long double)h2[i] - ex) / ex));
  }
  printf("v_rcp_f64 max rel err: raw %.3e (2^%.1f), +1 NR %.3e, +2 NR %.3e\n", e0, std::log2(e0), e1, e2);
  return 0;
}
