// sr_device.hpp -- device-side arithmetic of the Voigt / coefficient path (gfx950).
//
// Compiled with -ffp-contract=off: a*b+c written with operators is a separate
// multiply and add (needed where the reference's x values must be reproduced
// bit for bit); fused multiply-adds are written fma() explicitly.
//
// What is reproduced from the reference (file:line into the reference tree):
//   lineshape.f:443-562  humliv_bb middle branch: index-computed region
//                        boundaries, overlapping inclusive loop ends, running x
//                        per region, fp32-rounded cmplx(ry,-rx) and literals.
//   spect_classes.py:1967-2008  widths and MakeShape normalisation.
//   spect_classes.py:1736-1853  Einstein / G coefficients.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sr {

constexpr int kImxsig = 13010;     // parameters.inc:65
constexpr int kHalf = kImxsig / 2; // window point k=1 sits on grid index ic-6505

// spect_classes.py:39-47 (scipy.constants CODATA-2018)
constexpr double kTref = 296.0;
constexpr double kHpaToAtm = 0.00098692326671601;
constexpr double kHcgs = 6.62607015e-34 * 1.e7;
constexpr double kCcgs = 299792458.0 * 1.e2;
constexpr double kKcgs = 1.380649e-23 * 1.e7;
constexpr double kC2 = kHcgs * kCcgs / kKcgs;
constexpr double kAvogadro = 6.02214076e23;
constexpr double kPi = 3.141592653589793;
constexpr double kLn2 = 0.6931471805599453; // math.log(2.0)

// Grid and line-window arithmetic exactly as numpy builds them
// (spect_main_module.py:1267, spect_classes.py:1446,1455).
struct GridParams {
  double w0, gstep;            // grid[j] = w0 + j*gstep
  double lin_start, lin_delta; // lin_grid[m] = lin_start + m*lin_delta
  int n_grid;
};

__host__ __device__ inline double grid_at(const GridParams &g, int j) {
  return g.w0 + (double)j * g.gstep;
}

// x(k), k 1-based, of the 13010-point window centred on grid[ic] = gc.
struct WinX {
  double lin_start, lin_delta, gc;
  __host__ __device__ inline double operator()(int k) const {
    return (lin_start + (double)(k - 1) * lin_delta) + gc;
  }
};
// x(k) read from a caller-supplied array (the f2py shim).
struct ArrX {
  const double *x;
  __host__ __device__ inline double operator()(int k) const { return x[k - 1]; }
};

// Per (layer, line) record read by every workgroup whose tile the line's window
// touches; staged through LDS in the main kernel.  80 bytes = 5 x 16 B.
struct __attribute__((aligned(16))) FastRec {
  double xl;    // (x0 - x(1))/dw          region-1 left running x at k = 1
  double xr;    // (x(ir) - x0)/dw         region-1 right running x at k = ir
  double xstep; // (x(2) - x(1))/dw
  double a, b, c, d; // region-1 rational (lineshape.f:456-459), e = 4
  double wabs, wemi; // level-population weighted G coefficients / fac
  int32_t j1;        // grid index of window point k = 1 (ic - 6505; may be < 0)
  uint32_t ilir;     // region-1 boundaries il | ir << 16 (1-based window indices);
                     // one dword so that the whole record is scalar-loadable
  __host__ __device__ inline int il() const { return (int)(ilir & 0xffffu); }
  __host__ __device__ inline int ir() const { return (int)(ilir >> 16); }
};
static_assert(sizeof(FastRec) == 80, "FastRec must be 80 bytes");

// Read only where a wave's points meet regions 2-4 or a window edge.
struct __attribute__((aligned(16))) ColdRec {
  double ry, dwp, x0;
  int16_t il2, ir2;
  int32_t pad;
};
static_assert(sizeof(ColdRec) == 32, "ColdRec must be 32 bytes");

// max(nint(v), 0) of lineshape.f:448,453,484,489 (nint rounds half away from zero)
__device__ inline int nint_clamp0(double v) { return v <= 0.0 ? 0 : (int)round(v); }

// 1/d without the IEEE corner-case handling (d is a positive, normal polynomial
// value on this path): v_rcp_f64 seed + NR Newton steps.
template <int NR>
__device__ inline double fast_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    double e = fma(-d, r, 1.0);
    r = fma(r, e, r);
  }
  return r;
}

// ---- region 1: lineshape.f:456-478 ----
__device__ inline void region1_coef(double ry, double &a, double &b, double &c, double &d) {
  double ry2 = ry * ry;
  a = ry * (1.1283792 + 2.2567584 * ry2);
  b = 2.2567584 * ry;
  c = (1. + 2. * ry2) * (1. + 2. * ry2);
  d = -4. + 8. * ry2;
}

// ---- region 2: lineshape.f:492-521 ----
__device__ inline double region2_val(double ry, double x) {
  double ry2 = ry * ry;
  double a = ry * (1.0578555 + ry2 * (4.6545642 + ry2 * (3.1030428 + 0.5641896 * ry2)));
  double b = ry * (2.9619954 + ry2 * (0.5641896 + 1.6925688 * ry2));
  double c = ry * (-2.5388532 + ry2 * 1.6925688);
  double d = ry * 0.5641896;
  double e = 0.5625 + ry2 * (4.5 + ry2 * (10.5 + ry2 * (6. + ry2)));
  double f = -4.5 + ry2 * (9. + ry2 * (6. + 4. * ry2));
  double g = 10.5 + ry2 * (-6. + 6. * ry2);
  double h = 4. * ry2 - 6.;
  double x2 = x * x;
  return (a + x2 * (b + x2 * (c + d * x2))) / (e + x2 * (f + x2 * (g + x2 * (h + x2))));
}

// ---- regions 3 / 4: lineshape.f:526-561 ----
struct cplx {
  double re, im;
};
__device__ inline cplx cmul(cplx a, cplx b) {
  return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
__device__ inline cplx radd(double r, cplx a) { return {r + a.re, a.im}; }  // r + a
__device__ inline cplx rsub(double r, cplx a) { return {r - a.re, -a.im}; } // r - a
__device__ inline cplx rmul(cplx a, double r) { return {a.re * r, a.im * r}; }
__device__ inline double cdiv_re(cplx n, cplx d) {
  return (n.re * d.re + n.im * d.im) / (d.re * d.re + d.im * d.im);
}
#define SR_F32(lit) ((double)(lit##f)) // un-suffixed Fortran literal

__device__ inline double core_point(double rx, double ry) {
  double r2 = (0.195 * rx) - 0.176;
  cplx c2 = {(double)(float)ry, (double)(float)(-rx)}; // default-kind cmplx()
  if (ry < r2) { // region 4
    cplx c1 = cmul(c2, c2);
    cplx n = rsub(SR_F32(1.320522), rmul(c1, SR_F32(.56419)));
    n = rsub(SR_F32(35.76683), cmul(c1, n));
    n = rsub(SR_F32(219.0313), cmul(c1, n));
    n = rsub(SR_F32(1540.787), cmul(c1, n));
    n = rsub(SR_F32(3321.9905), cmul(c1, n));
    n = rsub(SR_F32(36183.31), cmul(c1, n));
    n = cmul(c2, n);
    cplx d = rsub(SR_F32(1.841439), c1);
    d = rsub(SR_F32(61.57037), cmul(c1, d));
    d = rsub(SR_F32(364.2191), cmul(c1, d));
    d = rsub(SR_F32(2186.181), cmul(c1, d));
    d = rsub(SR_F32(9022.228), cmul(c1, d));
    d = rsub(SR_F32(24322.84), cmul(c1, d));
    d = rsub(SR_F32(32066.6), cmul(c1, d));
    return exp(c1.re) * cos(c1.im) - cdiv_re(n, d);
  } else { // region 3
    cplx n = radd(SR_F32(3.778987), rmul(c2, SR_F32(.5642236)));
    n = radd(SR_F32(11.96482), cmul(c2, n));
    n = radd(SR_F32(20.20933), cmul(c2, n));
    n = radd(SR_F32(16.4955), cmul(c2, n));
    cplx d = radd(SR_F32(6.699398), c2);
    d = radd(SR_F32(21.69274), cmul(c2, d));
    d = radd(SR_F32(39.27121), cmul(c2, d));
    d = radd(SR_F32(38.82363), cmul(c2, d));
    d = radd(SR_F32(16.4955), cmul(c2, d));
    return cdiv_re(n, d);
  }
}

// Region boundaries and running-x starts of one humliv_bb call on x(1..n)
// (middle branch, lineshape.f:443-490), with xf(k) giving x(k).
struct Bounds {
  double ry, xstep, xl, xr;
  int il, ir, il2, ir2;
};
template <class XF>
__device__ inline Bounds humliv_bounds(const XF &xf, int n, double x0, double lw, double dwp) {
  Bounds B;
  B.ry = lw / dwp;                      // :261
  B.xstep = (xf(2) - xf(1)) / dwp;      // :265-266
  double rx = (x0 - xf(1)) / dwp;       // :444
  B.xl = rx;                            // :462
  B.il = 1;
  if (rx + B.ry >= 15.) B.il = nint_clamp0((rx - B.ry - 15.) / B.xstep) + 1; // :447-449
  rx = (xf(n) - x0) / dwp;
  B.ir = n;
  if (rx + B.ry >= 15.) B.ir = n - nint_clamp0((rx - B.ry - 15.) / B.xstep); // :452-454
  B.xr = (xf(B.ir) - x0) / dwp;         // :471
  rx = (x0 - xf(B.il)) / dwp;           // :480
  B.il2 = B.il;
  if (rx + B.ry >= 5.5) B.il2 = B.il + nint_clamp0((rx - B.ry - 5.5) / B.xstep); // :483-485
  rx = B.xr;                            // :487
  B.ir2 = B.ir;
  if (rx + B.ry >= 5.5) B.ir2 = B.ir - nint_clamp0((rx - B.ry - 5.5) / B.xstep); // :488-490
  return B;
}

// Value of humliv_bb at 1-based index k (1..n) for one (line, layer), any region;
// follows the write order of lineshape.f:455-562 (last writer wins).
template <class XF>
__device__ inline double humliv_point(int k, const FastRec &r, const ColdRec &cr, const XF &xf) {
  const int il = r.il(), ir = r.ir(), il2 = cr.il2, ir2 = cr.ir2;
  const int il2a = (il2 == il) ? il - 1 : il2; // :524-525
  const int ir2a = (ir2 == ir) ? ir + 1 : ir2;
  if (k > il2a && k < ir2a) { // :526-562
    double rx = fabs(xf(k) - cr.x0) / cr.dwp;
    return core_point(rx, cr.ry);
  }
  if (il < il2 && k >= il && k <= il2) { // :503-512
    double xs = (cr.x0 - xf(il)) / cr.dwp;
    return region2_val(cr.ry, fma(-(double)(k - il), r.xstep, xs));
  }
  if (ir2 < ir && k >= ir2 && k <= ir) { // :513-522
    double xs = (xf(ir2) - cr.x0) / cr.dwp;
    return region2_val(cr.ry, fma((double)(k - ir2), r.xstep, xs));
  }
  double x = (k <= il) ? fma(-(double)(k - 1), r.xstep, r.xl)  // :461-468
                       : fma((double)(k - ir), r.xstep, r.xr); // :470-477
  double x2 = x * x;
  return (r.a + x2 * r.b) / (r.c + x2 * (r.d + 4. * x2));
}

} // namespace sr
