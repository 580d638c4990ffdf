// sr_kernels.hip -- HIP kernels of the spectral hot path for gfx950 (MI355X).
//
//   sr_prep_kernel      one thread per (line, layer): widths, G coefficients,
//                       level-population weights, Humlicek region boundaries
//                       -> FastRec / ColdRec tables in HBM.
//   sr_abscoeff_kernel  the dominant kernel.  Gather formulation: a workgroup
//                       owns a tile of grid points of one layer, walks the
//                       nu-sorted lines whose 13010-point windows touch the
//                       tile, stages their FastRecs through LDS, and every lane
//                       accumulates abs/emi for its own P points in registers:
//                       no atomics, no [n_lines x 13010] matrix, coalesced
//                       fp64 stores.  fp64 VALU bound (no MFMA: not a
//                       contraction, one divide per evaluation).
//   sr_radiance_kernel  limb recursion per (point, ray).
//   shims               humliv_bb / sum_all_lines / curgod_fort_N call shapes.
#include "sr_device.hpp"
#include "sr_kernels.hpp"

namespace sr {

// ------------------------------------------------------------------------
// prep: spect_classes.py:174-206 (MakeShapeLine), 312-343 (Calc_Gcoeffs),
//       spect_main_module.py:2049-2080 (population weights), lineshape.f:443-490
// ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sr_prep_kernel(LinesDev L, LayersDev A, GridParams gp,
                                                      int line_lo, int n_sub,
                                                      FastRec *__restrict__ fast,
                                                      ColdRec *__restrict__ cold) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;
  if (i >= n_sub) return;
  const int ln = line_lo + i;
  const double T = A.temps[k];
  const double x0 = L.freq[ln];

  // spect_classes.py:1972, 1984, 1997-1999
  const double lw = pow(A.trat[k], L.t_dep[ln]) * (L.air_broad[ln] * A.p_atm[k]);
  const double dw = x0 / kCcgs * A.sqk[k];
  const double dwp = dw / A.sqrt_ln2;
  const double fac = dw * A.sqrt_pi_ln2;

  const int ic = L.ic[ln];
  WinX xf{gp.lin_start, gp.lin_delta, grid_at(gp, ic)};
  const Bounds B = humliv_bounds(xf, kImxsig, x0, lw, dwp);

  // spect_classes.py:326-337, 1806-1853
  double g_sp = 0., g_in = 0., g_ab = 0.;
  const double a_co = L.a_coeff[ln], gu = L.g_up[ln], gl = L.g_lo[ln];
  if (a_co != 0.0 && gl != 0.0 && gu != 0.0) {
    const double four_pi = 4 * kPi;
    const double el = L.e_lower[ln];
    const double rot_up = gu * exp(-kC2 * (el + x0 - L.evib_up[ln]) / T);
    const double rot_lo = gl * exp(-kC2 * (el - L.evib_lo[ln]) / T);
    const double hcf = L.hcf[ln];
    g_sp = hcf * rot_up * a_co / four_pi;
    g_in = hcf * rot_up * L.b21[ln] / four_pi;
    g_ab = hcf * rot_lo * L.b12[ln] / four_pi;
  }
  // spect_main_module.py:2073-2080 folded per line
  const double *pop = A.pop + (size_t)k * A.n_pop;
  const double pu = pop[L.lev_up[ln]], pl = pop[L.lev_lo[ln]];
  const double wabs = pl * g_ab - pu * g_in;
  const double wemi = pu * g_sp;

  FastRec r;
  r.xl = B.xl;
  r.xr = B.xr;
  r.xstep = B.xstep;
  region1_coef(B.ry, r.a, r.b, r.c, r.d);
  r.wabs = wabs / fac; // shape = y/fac, spect_classes.py:2003
  r.wemi = wemi / fac;
  r.j1 = ic - kHalf;
  r.il = (int16_t)B.il;
  r.ir = (int16_t)B.ir;
  ColdRec c;
  c.ry = B.ry;
  c.dwp = dwp;
  c.x0 = x0;
  c.il2 = (int16_t)B.il2;
  c.ir2 = (int16_t)B.ir2;
  c.pad = 0;
  const size_t o = (size_t)k * n_sub + i;
  fast[o] = r;
  cold[o] = c;
}

// ------------------------------------------------------------------------
// main gather kernel
// ------------------------------------------------------------------------
__device__ inline int lower_bound_ic(const int *__restrict__ ic, int n, int v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (ic[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// XCD-aware bijective remap (8 XCDs, blocks dealt round-robin): blocks that
// share an XCD get consecutive work ids, so neighbouring tiles of one layer --
// which read almost the same FastRecs -- hit the same L2.
__device__ inline int xcd_remap(int b, int nb) {
  const int q = nb >> 3, r = nb & 7, x = b & 7, i = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

template <int P, int NR>
__global__ __launch_bounds__(256) void sr_abscoeff_kernel(
    const FastRec *__restrict__ fast, const ColdRec *__restrict__ cold,
    const int *__restrict__ ic_sub, // [n_sub] window centres of the prepped lines, sorted
    int n_sub, int n_tiles, int g_lo, int g_hi, GridParams gp,
    double *__restrict__ abs_out, double *__restrict__ emi_out) {
  constexpr int WP = 64 * P;  // points per wave
  constexpr int TP = 4 * WP;  // points per workgroup tile
  constexpr int CH = 256;     // lines per LDS chunk
  __shared__ FastRec s_rec[CH]; // 20 KiB

  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int layer = wg / n_tiles, tile = wg - layer * n_tiles;
  const int t0 = g_lo + tile * TP;
  const int thi = min(t0 + TP, g_hi) - 1;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wlo = t0 + wave * WP;
  const int whi = min(wlo + WP, g_hi) - 1;
  const bool wave_on = wlo <= whi;

  // lines whose window [ic-6505, ic+6504] meets [t0, thi]
  const int l0 = lower_bound_ic(ic_sub, n_sub, t0 - (kHalf - 1));
  const int l1 = lower_bound_ic(ic_sub, n_sub, thi + kHalf + 1);

  double acc_a[P], acc_e[P], flp[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    acc_a[p] = 0.;
    acc_e[p] = 0.;
    flp[p] = (double)(lane + 64 * p);
  }
  const FastRec *frow = fast + (size_t)layer * n_sub;
  const ColdRec *crow = cold + (size_t)layer * n_sub;

  for (int c = l0; c < l1; c += CH) {
    const int cnt = min(CH, l1 - c);
    __syncthreads();
    {
      const int4 *src = reinterpret_cast<const int4 *>(frow + c);
      int4 *dst = reinterpret_cast<int4 *>(s_rec);
      for (int i = threadIdx.x; i < cnt * 5; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    if (!wave_on) continue;
    for (int l = 0; l < cnt; ++l) {
      const FastRec &r = s_rec[l];
      const int j1 = __builtin_amdgcn_readfirstlane(r.j1);
      const int jN = j1 + (kImxsig - 1);
      if (jN < wlo || j1 > whi) continue;
      const int il = __builtin_amdgcn_readfirstlane((int)r.il);
      const int ir = __builtin_amdgcn_readfirstlane((int)r.ir);
      const int jil = j1 + il - 1, jir = j1 + ir - 1;
      double xb;
      bool fastp = false;
      if (wlo >= j1 && whi < jil) { // all points strictly left of il: region 1
        xb = fma((double)(wlo - j1), r.xstep, -r.xl);
        fastp = true;
      } else if (wlo > jir && whi <= jN) { // strictly right of ir
        xb = fma((double)(wlo - jir), r.xstep, r.xr);
        fastp = true;
      }
      if (fastp) {
        const double xs = r.xstep, a = r.a, b = r.b, cc = r.c, d = r.d;
        const double wa = r.wabs, we = r.wemi;
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const double x = fma(flp[p], xs, xb);
          const double x2 = x * x;
          const double num = fma(x2, b, a);
          const double den = fma(x2, fma(x2, 4.0, d), cc);
          const double q = num * fast_rcp<NR>(den);
          acc_a[p] = fma(wa, q, acc_a[p]);
          acc_e[p] = fma(we, q, acc_e[p]);
        }
      } else {
        const ColdRec cr = crow[c + l];
        WinX xf{gp.lin_start, gp.lin_delta, grid_at(gp, j1 + kHalf)};
        const FastRec rr = r;
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const int k = wlo + lane + 64 * p - j1 + 1; // 1-based window index
          if (k >= 1 && k <= kImxsig) {
            const double y = humliv_point(k, rr, cr, xf);
            acc_a[p] = fma(rr.wabs, y, acc_a[p]);
            acc_e[p] = fma(rr.wemi, y, acc_e[p]);
          }
        }
      }
    }
  }
  if (wave_on) {
    const size_t row = (size_t)layer * (size_t)(g_hi - g_lo);
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int j = wlo + lane + 64 * p;
      if (j <= whi) {
        abs_out[row + (j - g_lo)] = acc_a[p];
        emi_out[row + (j - g_lo)] = acc_e[p];
      }
    }
  }
}

int launch_prep(const LinesDev &L, const LayersDev &A, const GridParams &gp, int line_lo, int n_sub,
                FastRec *fast, ColdRec *cold, hipStream_t st) {
  if (n_sub <= 0 || A.n_layers <= 0) return 0;
  dim3 grid((n_sub + 255) / 256, A.n_layers);
  hipLaunchKernelGGL(sr_prep_kernel, grid, dim3(256), 0, st, L, A, gp, line_lo, n_sub, fast, cold);
  return (int)hipGetLastError();
}

int abscoeff_tile_points(int variant) { return 256 * (variant == 2 ? 2 : (variant == 1 ? 1 : 4)); }

int launch_abscoeff(int variant, const FastRec *fast, const ColdRec *cold, const int *ic_sub, int n_sub,
                    int n_layers, int g_lo, int g_hi, const GridParams &gp, double *abs_out,
                    double *emi_out, hipStream_t st) {
  const int tp = abscoeff_tile_points(variant);
  const int n_tiles = (g_hi - g_lo + tp - 1) / tp;
  if (n_tiles <= 0 || n_layers <= 0) return 0;
  dim3 grid((unsigned)(n_tiles * n_layers));
#define SR_LAUNCH(P, NR)                                                                          \
  hipLaunchKernelGGL((sr_abscoeff_kernel<P, NR>), grid, dim3(256), 0, st, fast, cold, ic_sub,     \
                     n_sub, n_tiles, g_lo, g_hi, gp, abs_out, emi_out)
  if (variant == 1) SR_LAUNCH(1, 2);
  else if (variant == 2) SR_LAUNCH(2, 2);
  else SR_LAUNCH(4, 2);
#undef SR_LAUNCH
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------
// radiance recursion (build's own definition, see include/spectrobot_hip.h)
// ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sr_radiance_kernel(const double *__restrict__ abs_c,
                                                          const double *__restrict__ emi_c,
                                                          int n_pts, int n_rays,
                                                          const int *__restrict__ seg_off,
                                                          const int *__restrict__ seg_layer,
                                                          const double *__restrict__ seg_col,
                                                          int init_from_rad, double *__restrict__ rad) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int ray = blockIdx.y;
  if (j >= n_pts) return;
  double I = init_from_rad ? rad[(size_t)ray * n_pts + j] : 0.0;
  const int s0 = seg_off[ray], s1 = seg_off[ray + 1];
  for (int s = s0; s < s1; ++s) {
    const size_t o = (size_t)seg_layer[s] * n_pts + j;
    const double u = seg_col[s];
    const double tau = abs_c[o] * u;
    const double em1 = -expm1(-tau);
    const double src = fabs(tau) > 1e-12 ? (emi_c[o] * u) * (em1 / tau) : emi_c[o] * u;
    I = I * exp(-tau) + src;
  }
  rad[(size_t)ray * n_pts + j] = I;
}

int launch_radiance(const double *abs_c, const double *emi_c, int n_pts, int n_rays, const int *seg_off,
                    const int *seg_layer, const double *seg_col, int init_from_rad, double *rad,
                    hipStream_t st) {
  if (n_pts <= 0 || n_rays <= 0) return 0;
  dim3 grid((n_pts + 255) / 256, n_rays);
  hipLaunchKernelGGL(sr_radiance_kernel, grid, dim3(256), 0, st, abs_c, emi_c, n_pts, n_rays, seg_off,
                     seg_layer, seg_col, init_from_rad, rad);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------
// shims
// ------------------------------------------------------------------------
// lineshape.f:226-569, middle branch, on a caller-supplied x(i1..i2)
__global__ __launch_bounds__(256) void sr_humliv_kernel(const double *__restrict__ x, int i1, int n,
                                                        double x0, double lw, double dwp,
                                                        double *__restrict__ y) {
  ArrX xf{x + (i1 - 1)};
  const Bounds B = humliv_bounds(xf, n, x0, lw, dwp); // cheap; every thread recomputes
  FastRec r;
  r.xl = B.xl; r.xr = B.xr; r.xstep = B.xstep;
  region1_coef(B.ry, r.a, r.b, r.c, r.d);
  r.wabs = r.wemi = 1.0; r.j1 = 0;
  r.il = (int16_t)B.il; r.ir = (int16_t)B.ir;
  ColdRec c;
  c.ry = B.ry; c.dwp = dwp; c.x0 = x0; c.il2 = (int16_t)B.il2; c.ir2 = (int16_t)B.ir2; c.pad = 0;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x + 1; k <= n; k += gridDim.x * blockDim.x)
    y[i1 - 1 + k - 1] = humliv_point(k, r, c, xf);
}

int launch_humliv(const double *x, int i1, int i2, double x0, double lw, double dwp, double *y,
                  hipStream_t st) {
  const int n = i2 - i1 + 1;
  hipLaunchKernelGGL(sr_humliv_kernel, dim3((n + 255) / 256), dim3(256), 0, st, x, i1, n, x0, lw, dwp, y);
  return (int)hipGetLastError();
}

// lineshape.f:15-23 as a gather: thread j adds the rows in line order, so the
// floating-point sum order is the Fortran's.
__global__ __launch_bounds__(256) void sr_sum_lines_kernel(double *__restrict__ spe, long n_spe,
                                                           const double *__restrict__ rows,
                                                           const int *__restrict__ init,
                                                           const int *__restrict__ fin, int n_lines,
                                                           int row_len) {
  __shared__ int s_init[256], s_fin[256];
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x + 1; // 1-based
  double acc = j <= n_spe ? spe[j - 1] : 0.0;
  for (int c = 0; c < n_lines; c += 256) {
    __syncthreads();
    if (c + (int)threadIdx.x < n_lines) {
      s_init[threadIdx.x] = init[c + threadIdx.x];
      s_fin[threadIdx.x] = fin[c + threadIdx.x];
    }
    __syncthreads();
    const int cnt = min(256, n_lines - c);
    for (int l = 0; l < cnt; ++l) {
      const int a = s_init[l], b = s_fin[l];
      if (j >= a && j <= b) acc = acc + rows[(size_t)(c + l) * row_len + (j - a)];
    }
  }
  if (j <= n_spe) spe[j - 1] = acc;
}

int launch_sum_lines(double *spe, long n_spe, const double *rows, const int *init, const int *fin,
                     int n_lines, int row_len, hipStream_t st) {
  if (n_spe <= 0) return 0;
  hipLaunchKernelGGL(sr_sum_lines_kernel, dim3((unsigned)((n_spe + 255) / 256)), dim3(256), 0, st, spe,
                     n_spe, rows, init, fin, n_lines, row_len);
  return (int)hipGetLastError();
}

// curgods.f:2-98, one thread per LOS segment
__global__ void sr_curgod_kernel(int which, const double *__restrict__ nd, const double *__restrict__ vmr,
                                 const double *__restrict__ f, const double *__restrict__ x,
                                 const int *__restrict__ off, int n_seg, double *__restrict__ res) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_seg) return;
  double acc = 0.0;
  for (int i = off[s]; i < off[s + 1] - 1; ++i) {
    const double dx = x[i + 1] - x[i];
    if (which == 1) {
      const double fu = nd[i + 1] / nd[i];
      const double D = log(fu) / dx;
      acc = acc + (nd[i + 1] - nd[i]) / D;
    } else if (which == 2) {
      const double A = nd[i] * vmr[i];
      const double B = nd[i] * (vmr[i + 1] - vmr[i]) / dx;
      const double fu = nd[i + 1] / nd[i];
      const double D = log(fu) / dx;
      acc = acc + (A * D * (fu - 1.) + B * fu * (D * dx - 1.) + B) / (D * D);
    } else if (which == 3) {
      const double A = nd[i] * vmr[i] * f[i];
      const double cc = (vmr[i + 1] - vmr[i]) / dx;
      const double bb = (f[i + 1] - f[i]) / dx;
      const double B = nd[i] * (vmr[i] * bb + f[i] * cc);
      const double C = nd[i] * bb * cc;
      const double fu = nd[i + 1] / nd[i];
      const double D = log(fu) / dx;
      acc = acc + (fu * (D * (A * D + B * (D * dx - 1.)) + C * (D * dx * (D * dx - 2.) + 2.)) +
                   D * (B - A * D) - 2 * C) / (D * D * D);
    } else {
      const double A = nd[i] * vmr[i] * f[i];
      const double cc = (vmr[i + 1] - vmr[i]) / dx;
      const double B = nd[i] * f[i] * cc;
      const double fu = nd[i + 1] * f[i + 1] / (nd[i] * f[i]);
      const double D = log(fu) / dx;
      acc = acc + (A * D * (fu - 1.) + B * fu * (D * dx - 1.) + B) / (D * D);
    }
  }
  res[s] = acc;
}

int launch_curgod(int which, const double *nd, const double *vmr, const double *f, const double *x,
                  const int *off, int n_seg, double *res, hipStream_t st) {
  if (n_seg <= 0) return 0;
  hipLaunchKernelGGL(sr_curgod_kernel, dim3((n_seg + 63) / 64), dim3(64), 0, st, which, nd, vmr, f, x, off,
                     n_seg, res);
  return (int)hipGetLastError();
}

} // namespace sr
