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
#include <type_traits>
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
#ifndef SR_FASTREC64
#define SR_FASTREC64 1 // 64-byte records: the four region-1 coefficients are rebuilt from ry at every use (0: stored, 80 bytes)
#endif
struct __attribute__((aligned(16))) FastRec {
  double xl;    // (x0 - x(1))/dw          region-1 left running x at k = 1
  double xr;    // (x(ir) - x0)/dw         region-1 right running x at k = ir
  double xstep; // (x(2) - x(1))/dw
#if SR_FASTREC64
  double ry;         // lw/dw': a, b, c, d are functions of it alone (region1_coef)
#else
  double a, b, c, d; // region-1 rational (lineshape.f:456-459), e = 4
#endif
  double wabs, wemi; // level-population weighted G coefficients / fac
  int32_t j1;        // grid index of window point k = 1 (ic - 6505; may be < 0)
  uint32_t ilir;     // region-1 boundaries il | ir << 16 (1-based window indices);
                     // one dword so that the whole record is scalar-loadable
#if SR_FASTREC64
  double w3;         // third weight of the multi-channel pass (kWeightChannels: the line's G_ind; 0 in every other mode)
#endif
  __host__ __device__ inline int il() const { return (int)(ilir & 0xffffu); }
  __host__ __device__ inline int ir() const { return (int)(ilir >> 16); }
};
static_assert(sizeof(FastRec) == (SR_FASTREC64 ? 64 : 80), "FastRec must be 80 (64) bytes");

// What regions 2-4 of one (line, layer) need beyond the FastRec, as stored: 48 bytes.  The eight region-2
// coefficients, (double)(float)ry and 1/dw' are functions of ry and dw' alone and are rebuilt where a line's
// record is taken into registers (ColdFull, expand_cold: ~35 flops per line) -- stored, they were 80 of the
// record's 128 bytes, 0.64 GB of config 2's tables written and 0.9 GB read per call.
struct __attribute__((aligned(16))) ColdRec {
  double ry;             // lw/dw'
  double dwp;            // dw' = dw/sqrt(ln2)
  double x0;             // line centre
  double xs2l, xs2r;     // region-2 running-x starts (lineshape.f:504, 514)
  uint32_t il2ir2;       // il2 | ir2 << 16
  uint32_t k3;           // region-3 interval of the core, k3lo | k3hi << 16 (k3lo > k3hi: empty)
  __host__ __device__ inline int il2() const { return (int)(il2ir2 & 0xffffu); }
  __host__ __device__ inline int ir2() const { return (int)(il2ir2 >> 16); }
  __host__ __device__ inline int k3lo() const { return (int)(k3 & 0xffffu); }
  __host__ __device__ inline int k3hi() const { return (int)(k3 >> 16); }
};
static_assert(sizeof(ColdRec) == 48, "ColdRec must be 48 bytes");
// The record in registers, with the derived values.
struct ColdFull {
  double ry, ryf;        // lw/dw' and (double)(float)ry: cmplx() is default kind (lineshape.f:529)
  double dwp, inv_dwp;   // dw' and ~1/dw'
  double x0, xs2l, xs2r;
  double q2[8];          // region-2 coefficients a..h (lineshape.f:492-502)
  uint32_t il2ir2, k3;
  __host__ __device__ inline int il2() const { return (int)(il2ir2 & 0xffffu); }
  __host__ __device__ inline int ir2() const { return (int)(il2ir2 >> 16); }
  __host__ __device__ inline int k3lo() const { return (int)(k3 & 0xffffu); }
  __host__ __device__ inline int k3hi() const { return (int)(k3 >> 16); }
};

// One (line, layer) of a line whose centre lies outside its own window: humliv_bb's sequential
// outer branches (lineshape.f:272-357 for x0 <= x(i1), :358-442 for x0 >= x(i2)) reduced to three
// index segments, each with the running value at the first index the Fortran visits there:
// value(k) = x_ref + dir * (k - k_ref) * xstep.  Window points in no segment keep y = 0.
struct OuterRec {
  double ry, ryf, xstep;
  double a, b, c, d;         // region 1 (lineshape.f:335-339)
  double q2[8];              // region 2 (:316-325)
  double wabs, wemi;
  double x_core, x_r2, x_r1; // running rx / x at c_ref, r2_lo, r1_lo
  int dir;                   // +1: x0 <= x(i1);  -1: x0 >= x(i2)
  int j1;                    // grid index of window point k = 1
  int c_ref, c_lo, c_hi;     // core (while loop): k in [c_lo, c_hi], running rx starts at c_ref
  int r2_lo, r2_hi, r1_lo, r1_hi; // 1-based inclusive; lo > hi: empty
};

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

// The region-1 coefficients of a record: its fields, or (SR_FASTREC64) rebuilt from ry.
struct R1Coef { double a, b, c, d; };
__device__ inline R1Coef r1_of(const FastRec &r) {
  R1Coef q;
#if SR_FASTREC64
  region1_coef(r.ry, q.a, q.b, q.c, q.d);
#else
  q.a = r.a; q.b = r.b; q.c = r.c; q.d = r.d;
#endif
  return q;
}
__device__ inline void r1_set(FastRec &r, double ry) {
#if SR_FASTREC64
  r.ry = ry;
  r.w3 = 0.0;
#else
  region1_coef(ry, r.a, r.b, r.c, r.d);
#endif
}

// ---- region 2: lineshape.f:492-521 ----
__device__ inline void region2_coef(double ry, double (&q)[8]) {
  const double ry2 = ry * ry;
  q[0] = ry * (1.0578555 + ry2 * (4.6545642 + ry2 * (3.1030428 + 0.5641896 * ry2)));
  q[1] = ry * (2.9619954 + ry2 * (0.5641896 + 1.6925688 * ry2));
  q[2] = ry * (-2.5388532 + ry2 * 1.6925688);
  q[3] = ry * 0.5641896;
  q[4] = 0.5625 + ry2 * (4.5 + ry2 * (10.5 + ry2 * (6. + ry2)));
  q[5] = -4.5 + ry2 * (9. + ry2 * (6. + 4. * ry2));
  q[6] = 10.5 + ry2 * (-6. + 6. * ry2);
  q[7] = 4. * ry2 - 6.;
}
// The same coefficients with every Horner step as one fma (25 -> 14 instructions; <= 1 ulp from region2_coef: 1e-16
// of a region-2 value).  For the zones kernel's rows, which rebuild the coefficients per round of eight lines; the
// shim and the exact-mode kernels keep the reference's operation order.  The constants go through SGPR pairs
// (fma_vvs: s_mov right where they are used): as plain fma() operands the compiler parked all 27 of them in VGPRs
// for the whole kernel (120 -> 140 VGPRs: three waves per SIMD instead of four).
__device__ inline double fma_vvs(double a, double b, double c_uniform) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c_uniform));
  return r;
}
__device__ inline double fma_svv(double a_uniform, double b, double c) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "s"(a_uniform), "v"(b), "v"(c));
  return r;
}
__device__ inline double mul_vs(double a, double c_uniform) {
  double r;
  asm("v_mul_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(c_uniform));
  return r;
}
__device__ inline double add_vs(double a, double c_uniform) {
  double r;
  asm("v_add_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(c_uniform));
  return r;
}
__device__ inline double fma_v4s(double a, double c_uniform) { // 4 a + c
  double r;
  asm("v_fma_f64 %0, %1, 4.0, %2" : "=v"(r) : "v"(a), "s"(c_uniform));
  return r;
}
__device__ inline void region2_coef_fma(double ry, double (&q)[8]) {
  const double ry2 = ry * ry;
  // a step with two constants is a multiply and an add (one constant-bus operand per VOP3; 4.0 is an inline constant)
  q[0] = ry * fma_vvs(ry2, fma_vvs(ry2, add_vs(mul_vs(ry2, 0.5641896), 3.1030428), 4.6545642), 1.0578555);
  q[1] = ry * fma_vvs(ry2, add_vs(mul_vs(ry2, 1.6925688), 0.5641896), 2.9619954);
  q[2] = ry * add_vs(mul_vs(ry2, 1.6925688), -2.5388532);
  q[3] = mul_vs(ry, 0.5641896);
  q[4] = fma_vvs(ry2, fma_vvs(ry2, fma_vvs(ry2, add_vs(ry2, 6.), 10.5), 4.5), 0.5625);
  q[5] = fma_vvs(ry2, fma_vvs(ry2, fma_v4s(ry2, 6.), 9.), -4.5);
  q[6] = fma_vvs(ry2, add_vs(mul_vs(ry2, 6.), -6.), 10.5);
  q[7] = fma_v4s(ry2, -6.);
}
__device__ inline double region2_val(const double (&q)[8], double x) {
  const double x2 = x * x;
  const double num = fma(x2, fma(x2, fma(q[3], x2, q[2]), q[1]), q[0]);
  const double den = fma(x2, fma(x2, fma(x2, x2 + q[7], q[6]), q[5]), q[4]);
  return num * fast_rcp<1>(den);
}

// ---- regions 3 / 4: lineshape.f:526-561 ----
#define SR_F32(lit) ((double)(lit##f)) // un-suffixed Fortran literal: rounded to single

// Real-coefficient polynomial at the complex point (zr, zi), coefficients c[0..N]
// in ascending powers: the quadratic-factor recurrence (2 fma per coefficient
// instead of a complex multiply-add; agrees with the reference's complex Horner
// to 4e-15 in region 3 and 7e-14 in region 4, checked on the host in fp80).
template <int N>
__device__ inline void real_poly_at(const double (&c)[N + 1], double zr, double zi, double &pr, double &pi) {
  const double r = zr + zr, s = fma(zr, zr, zi * zi);
  double a = c[N], b = c[N - 1];
#pragma unroll
  for (int j = 2; j <= N; ++j) {
    const double t = fma(r, a, b);
    b = fma(-s, a, c[N - j]);
    a = t;
  }
  pr = fma(zr, a, b);
  pi = zi * a;
}

// a*b + c in the three-address form.  For a Horner chain whose coefficients live in
// VGPRs across a loop the compiler emits the two-address v_fmac_f64 and a v_mov_b64 copy of
// the coefficient per step (21 copies in the region-4 body); this keeps one instruction.
__device__ inline double fma3(double a, double b, double c) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// The same with the addend in an SGPR pair (wave-uniform coefficient fetched by a scalar load):
// otherwise two v_mov_b32 + v_fmac_f64 per Horner step.
__device__ inline double fma3s(double a, double b, double c_uniform) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c_uniform));
  return r;
}

// cos(t) for |t| <= ~1e3 (here |Im c1| = 2 ry rx < 10): three-constant Cody-Waite
// reduction by pi/2 and the fdlibm kernel polynomials, ~1 ulp.
__device__ inline double cos_bounded(double t) {
  // |t| < 0.78 on every active lane (2 ry rx of a low-pressure layer): n = 0, r = t and the
  // result is the cosine kernel -- bit for bit what the general path returns, without the
  // reduction, the sine kernel and the quadrant selects.
  const bool small = __all(fabs(t) < 0.78);
  double n = 0., r = t;
  if (!small) {
    n = rint(t * 6.36619772367581382433e-01);            // 2/pi
    r = fma(-n, 1.57079632673412561417e+00, t);           // pi/2 hi (33 bits)
    r = fma(-n, 6.07710050650619224932e-11, r);           // pi/2 mid
    r = fma(-n, 2.02226624879595063154e-21, r);           // pi/2 lo
  }
  const double z = r * r;
  double pc = fma3(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = fma3(z, pc, -2.75573143513906633035e-07);
  pc = fma3(z, pc, 2.48015872894767294178e-05);
  pc = fma3(z, pc, -1.38888888888741095749e-03);
  pc = fma3(z, pc, 4.16666666666666019037e-02);
  const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
  if (small) return cr;
  double ps = fma3(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = fma3(z, ps, 2.75573137070700676789e-06);
  ps = fma3(z, ps, -1.98412698298579493134e-04);
  ps = fma3(z, ps, 8.33333333332248946124e-03);
  ps = fma3(z, ps, -1.66666666666666324348e-01);
  const double sr = fma(r * z, ps, r);
  const int q = (int)n & 3;
  const double v = (q & 1) ? sr : cr;          // cos(r + q pi/2): c, -s, -c, s
  return (q == 1 || q == 2) ? -v : v;
}

// exp(u) for -700 < u < 700 without the special-case handling of the library routine (here
// u = Re c1 in [-30.3, 0.1]): u = n ln2 + r, degree-11 minimax polynomial (the coefficients
// of ROCm device-libs' double-precision exp), 2^n by ldexp; <= 1 ulp (checked on the host
// against exact rational arithmetic, max 1.5e-16).
__device__ inline double exp_bounded(double u) {
  const double n = rint(u * 0x1.71547652b82fep+0);  // log2(e)
  double r = fma(-n, 0x1.62e42fefa39efp-1, u);      // ln2 hi
  r = fma(-n, 0x1.abc9e3b39803fp-56, r);            // ln2 lo
  double p = fma3(r, 0x1.ade156a5dcb37p-26, 0x1.28af3fca7ab0cp-22);
  p = fma3(r, p, 0x1.71dee623fde64p-19);
  p = fma3(r, p, 0x1.a01997c89e6b0p-16);
  p = fma3(r, p, 0x1.a01a014761f6ep-13);
  p = fma3(r, p, 0x1.6c16c1852b7b0p-10);
  p = fma3(r, p, 0x1.1111111122322p-7);
  p = fma3(r, p, 0x1.55555555502a1p-5);
  p = fma3(r, p, 0x1.5555555555511p-3);
  p = fma3(r, p, 0x1.000000000000bp-1);
  p = fma(r, p, 1.0);
  p = fma(r, p, 1.0);
  return ldexp(p, (int)n);
}

// One segment of the radiance recursion: t = exp(-tau), em1 = 1 - exp(-tau) and f = em1 / tau (1 where
// |tau| <= 1e-12) from ONE range reduction and polynomial: -tau = n ln2 + r, e^r - 1 = r (1 + r P(r)) =: pm1 with
// exp_bounded's polynomial, s = 2^n, t = s + s pm1, em1 = (1 - s) - s pm1 (n = 0: -pm1, exact for thin segments; no
// cancellation otherwise: |r| <= 0.35), the quotient by reciprocal + two Newton steps.  ~30 instructions for what
// exp() + expm1() + an IEEE division spent ~95 on: the recursion kernels are VALU-bound (64 rays x 160 segments
// x 1e5 points), sr_limb_kernel 1.5 -> see DESIGN.md.  tau may be negative (stimulated emission).
struct Atten {
  double t, em1, f, rtau; // rtau = 1 / tau (unspecified where |tau| <= 1e-12)
  bool thin;
};
__device__ inline Atten attenuation(double tau) {
  const double x = -tau;
  const double n = rint(x * 0x1.71547652b82fep+0);  // log2(e)
  double r = fma(-n, 0x1.62e42fefa39efp-1, x);      // ln2 hi
  r = fma(-n, 0x1.abc9e3b39803fp-56, r);            // ln2 lo
  double p = fma3(r, 0x1.ade156a5dcb37p-26, 0x1.28af3fca7ab0cp-22);
  p = fma3(r, p, 0x1.71dee623fde64p-19);
  p = fma3(r, p, 0x1.a01997c89e6b0p-16);
  p = fma3(r, p, 0x1.a01a014761f6ep-13);
  p = fma3(r, p, 0x1.6c16c1852b7b0p-10);
  p = fma3(r, p, 0x1.1111111122322p-7);
  p = fma3(r, p, 0x1.55555555502a1p-5);
  p = fma3(r, p, 0x1.5555555555511p-3);
  p = fma3(r, p, 0x1.000000000000bp-1);
  const double pm1 = r * fma(r, p, 1.0);
  const double s = ldexp(1.0, (int)fmin(fmax(n, -1100.0), 1100.0));
  Atten A;
  A.t = fma(s, pm1, s);
  A.em1 = fma(-s, pm1, 1.0 - s);
  A.thin = !(fabs(tau) > 1e-12);
  A.rtau = fast_rcp<2>(tau);
  A.f = A.thin ? 1.0 : A.em1 * A.rtau;
  return A;
}

// Regions 3 and 4 at c2 = (a, b) = ((double)(float)ry, (double)(float)(-rx)): cmplx() is
// default kind, both parts are rounded to single (lineshape.f:529).
__device__ inline double core_region4(double a, double b) { // :530-546
  const double P4[7] = {SR_F32(36183.31), -SR_F32(3321.9905), SR_F32(1540.787), -SR_F32(219.0313),
                        SR_F32(35.76683), -SR_F32(1.320522), SR_F32(.56419)};
  const double Q4[8] = {SR_F32(32066.6), -SR_F32(24322.84), SR_F32(9022.228), -SR_F32(2186.181),
                        SR_F32(364.2191), -SR_F32(61.57037), SR_F32(1.841439), -1.0};
  const double ur = fma(a, a, -(b * b)), ui = (a + a) * b; // c1 = c2*c2
  double pr, pi, qr, qi;
  real_poly_at<6>(P4, ur, ui, pr, pi);
  real_poly_at<7>(Q4, ur, ui, qr, qi);
  const double nr = fma(a, pr, -(b * pi)), ni = fma(a, pi, b * pr); // c2 * P
  const double ratio = fma(nr, qr, ni * qi) * fast_rcp<1>(fma(qr, qr, qi * qi));
  return exp_bounded(ur) * cos_bounded(ui) - ratio;
}
// a*b + c with b forced into a VGPR and c in an SGPR pair: the first Horner step of a polynomial whose two leading
// coefficients are both wave-uniform constants would otherwise need a v_mov_b64 (one constant-bus operand per VOP3).
__device__ inline double fma3vs(double a, double b_const, double c_uniform) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b_const), "s"(c_uniform));
  return r;
}

__device__ inline double fma3vs_neg(double a, double b_const, double c_uniform) { // -a*b + c
  double r;
  asm("v_fma_f64 %0, -%1, %2, %3" : "=v"(r) : "v"(a), "v"(b_const), "s"(c_uniform));
  return r;
}

// cos(t) with the argument range decided by the caller for the whole wave (tier, a scalar):
//   2: |t| < 6.5e-3 on every lane: 1 - z/2 + z^2/24, z = t^2 (next term z^3/720 <= 1e-16)
//   1: |t| < 0.78: the cosine kernel (what cos_bounded returns there, bit for bit)
//   0: anything (|t| <= ~1e3): cos_bounded
// In the zones kernel t = 2 ry rx with rx < ~5.6: tier 2 above ~390 km of the Titan-like profile (51 of 80 layers),
// tier 1 for all but the lowest few.
__device__ inline double cos_tiered(double t, int tier) {
  if (tier == 2) {
    const double z = t * t;
    return fma(z, fma(z, 4.16666666666666019037e-02, -0.5), 1.0);
  }
  if (tier == 1) {
    const double z = t * t;
    double pc = fma3(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma3(z, pc, -2.75573143513906633035e-07);
    pc = fma3(z, pc, 2.48015872894767294178e-05);
    pc = fma3(z, pc, -1.38888888888741095749e-03);
    pc = fma3(z, pc, 4.16666666666666019037e-02);
    return fma(z * z, pc, fma(-0.5, z, 1.0));
  }
  return cos_bounded(t);
}

// Region 4 (:530-546) for the row walks of the zones kernel: a = (double)(float)ry with a2 = a a and two_a = 2 a
// per line (a is a single: a a is exact), m = (double)(float)rx >= 0.  The value is even in the sign of
// Im c2 (P, Q have real coefficients: flipping b flips Im P, Im Q and Im(c2 P) together), so m stands in for
// b = -rx; Re c1 = a^2 - m^2 in one fma (m m is exact too: the same rounding as a a - b b).  cos by tier.
// p6_vgpr: the leading coefficient of P4 (SR_F32(.56419)) in a VGPR, made once per kernel by vgpr_constant().
__device__ inline double vgpr_constant(double c) {
  asm volatile("" : "+v"(c)); // opaque to the optimiser: stays a loop-invariant VGPR pair instead of a v_mov per use
  return c;
}
__device__ inline double core_region4_m(double a, double a2, double two_a, double m, int tier, double p6_vgpr) {
  const double ur = fma(-m, m, a2), ui = two_a * m;
  const double r = ur + ur, s = fma(ur, ur, ui * ui);
  // P4 = sum p_k c1^k, k = 0..6 (quadratic-factor recurrence, see real_poly_at)
  const double p6 = p6_vgpr;
  double pa = fma3vs(r, p6, -SR_F32(1.320522)), pb = fma3vs_neg(s, p6, SR_F32(35.76683));
  {
    const double P4[4] = {SR_F32(36183.31), -SR_F32(3321.9905), SR_F32(1540.787), -SR_F32(219.0313)};
#pragma unroll
    for (int j = 3; j >= 0; --j) {
      const double t = fma(r, pa, pb);
      pb = fma(-s, pa, P4[j]);
      pa = t;
    }
  }
  const double pr = fma(ur, pa, pb), pi = ui * pa;
  // Q4 = sum q_k c1^k, k = 0..7, q_7 = -1
  double qa = SR_F32(1.841439) - r, qb = s - SR_F32(61.57037);
  {
    const double Q4[5] = {SR_F32(32066.6), -SR_F32(24322.84), SR_F32(9022.228), -SR_F32(2186.181), SR_F32(364.2191)};
#pragma unroll
    for (int j = 4; j >= 0; --j) {
      const double t = fma(r, qa, qb);
      qb = fma(-s, qa, Q4[j]);
      qa = t;
    }
  }
  const double qr = fma(ur, qa, qb), qi = ui * qa;
  const double nr = fma(a, pr, -(m * pi)), ni = fma(a, pi, m * pr); // c2 * P
  const double ratio = fma(nr, qr, ni * qi) * fast_rcp<1>(fma(qr, qr, qi * qi));
  return fma(exp_bounded(ur), cos_tiered(ui, tier), -ratio);
}
__device__ inline double core_region3(double a, double b) { // :554-560
  const double N3[5] = {SR_F32(16.4955), SR_F32(20.20933), SR_F32(11.96482), SR_F32(3.778987),
                        SR_F32(.5642236)};
  const double D3[6] = {SR_F32(16.4955), SR_F32(38.82363), SR_F32(39.27121), SR_F32(21.69274),
                        SR_F32(6.699398), 1.0};
  double pr, pi, qr, qi;
  real_poly_at<4>(N3, a, b, pr, pi);
  real_poly_at<5>(D3, a, b, qr, qi);
  return fma(pr, qr, pi * qi) * fast_rcp<1>(fma(qr, qr, qi * qi));
}
// the reference's region test, :528-530
__device__ inline bool core_is_region4(double rx, double ry) { return ry < (0.195 * rx) - 0.176; }
__device__ inline double core_point(double rx, double ry, double ryf) {
  const double b = (double)(float)(-rx);
  return core_is_region4(rx, ry) ? core_region4(ryf, b) : core_region3(ryf, b);
}

// Region boundaries and running-x starts of one humliv_bb call on x(1..n)
// (middle branch, lineshape.f:443-490), with xf(k) giving x(k).
struct Bounds {
  double ry, xstep, xl, xr;
  double inv_dwp; // ~1/dw' (fast_rcp<2>): the divisions by dw' below and in make_cold share it
  int il, ir, il2, ir2;
};
// num / d with a reciprocal made once: q = num r, one residual correction -- the correctly rounded quotient (what the
// IEEE division returns) in all but ~1e-16 of cases, 3 instructions instead of the ~14 slots of v_div_scale / v_rcp /
// fma x 5 / v_div_fmas / v_div_fixup.  sr_prep_kernel had 31 IEEE divisions per record, 40 % of its instructions.
__device__ inline double div_with(double num, double d, double inv_d) {
  const double q = num * inv_d;
  return fma(fma(-d, q, num), inv_d, q);
}
template <class XF>
__device__ inline Bounds humliv_bounds(const XF &xf, int n, double x0, double lw, double dwp) {
  Bounds B;
  B.inv_dwp = fast_rcp<2>(dwp);
  auto by_dw = [&](double num) { return div_with(num, dwp, B.inv_dwp); };
  B.ry = by_dw(lw);                     // :261
  B.xstep = by_dw(xf(2) - xf(1));       // :265-266
  const double inv_xs = fast_rcp<2>(B.xstep);
  auto by_xs = [&](double num) { return div_with(num, B.xstep, inv_xs); };
  double rx = by_dw(x0 - xf(1));        // :444
  B.xl = rx;                            // :462
  B.il = 1;
  if (rx + B.ry >= 15.) B.il = nint_clamp0(by_xs(rx - B.ry - 15.)) + 1; // :447-449
  rx = by_dw(xf(n) - x0);
  B.ir = n;
  if (rx + B.ry >= 15.) B.ir = n - nint_clamp0(by_xs(rx - B.ry - 15.)); // :452-454
  B.xr = by_dw(xf(B.ir) - x0);          // :471
  rx = by_dw(x0 - xf(B.il));            // :480
  B.il2 = B.il;
  if (rx + B.ry >= 5.5) B.il2 = B.il + nint_clamp0(by_xs(rx - B.ry - 5.5)); // :483-485
  rx = B.xr;                            // :487
  B.ir2 = B.ir;
  if (rx + B.ry >= 5.5) B.ir2 = B.ir - nint_clamp0(by_xs(rx - B.ry - 5.5)); // :488-490
  return B;
}

// Fill the ColdRec of one (line, layer) (prep kernel / humliv shim).
template <class XF>
__device__ inline ColdRec make_cold(const Bounds &B, double dwp, double x0, const XF &xf) {
  ColdRec c;
  c.ry = B.ry;
  c.dwp = dwp;
  c.x0 = x0;
  c.xs2l = div_with(x0 - xf(B.il), dwp, B.inv_dwp);  // lineshape.f:504
  c.xs2r = div_with(xf(B.ir2) - x0, dwp, B.inv_dwp); // :514
  c.il2ir2 = (uint32_t)B.il2 | ((uint32_t)B.ir2 << 16);
  // Region-3 interval inside the core (il2a, ir2a): rx = |x(k)-x0|/dw grows away from the
  // centre and the region test is monotone in rx, so region 3 is one interval around the
  // centre; its ends are found with the reference's own per-point test (IEEE division).
  {
    const int n_ = B.ir; // (unused bound, keeps the signature small)
    (void)n_;
    const int clo = ((B.il2 == B.il) ? B.il - 1 : B.il2) + 1, chi = ((B.ir2 == B.ir) ? B.ir + 1 : B.ir2) - 1;
    int k3lo = 1, k3hi = 0; // empty
    if (clo <= chi) {
      auto is3 = [&](int k) { return !core_is_region4(div_with(fabs(xf(k) - x0), dwp, B.inv_dwp), B.ry); };
      if constexpr (std::is_same<XF, WinX>::value) {
        // The window grid is affine in k up to rounding, and region 3 is |x(k) - x0| <= (ry + 0.176)/0.195 dw up
        // to rounding: start each search at the index that formula gives and let the reference's own tests move it
        // (zero or one step).  Same results as the binary searches below -- both rely only on monotonicity -- for
        // 5 instead of ~16 evaluations of the test with its IEEE division (a quarter of sr_prep_kernel's
        // instructions).
        const double inv_delta = 1.0 / xf.lin_delta, base = xf.lin_start + xf.gc; // x(k) ~ base + (k - 1) delta
        auto near_index = [&](double x, int lo_, int hi_) {
          const double kf = fma(x - base, inv_delta, 1.0);
          const int k = (int)fmin(fmax(kf, (double)lo_ - 1.0), (double)hi_ + 1.0);
          return min(max(k, lo_), hi_);
        };
        int kc = near_index(x0, clo, chi); // first k in [clo, chi] with x(k) >= x0 (or chi)
        while (kc > clo && xf(kc - 1) >= x0) --kc;
        while (kc < chi && xf(kc) < x0) ++kc;
        if (kc > clo && fabs(xf(kc - 1) - x0) < fabs(xf(kc) - x0)) --kc;
        if (is3(kc)) {
          const double reach = (B.ry + 0.176) / 0.195 * dwp;
          int lo = near_index(x0 - reach, clo, kc); // smallest k in [clo, kc] with is3
          while (lo > clo && is3(lo - 1)) --lo;
          while (lo < kc && !is3(lo)) ++lo;
          k3lo = lo;
          int hi = near_index(x0 + reach, kc, chi); // largest k in [kc, chi] with is3
          while (hi < chi && is3(hi + 1)) ++hi;
          while (hi > kc && !is3(hi)) --hi;
          k3hi = hi;
        }
      } else {
        // the point closest to the centre: |x(k) - x0| is V-shaped in k
        int a = clo, b = chi;
        while (a < b) {
          const int m = (a + b) >> 1;
          if (xf(m) < x0) a = m + 1; else b = m;
        }
        int kc = a; // first k with x(k) >= x0 (or chi)
        if (kc > clo && fabs(xf(kc - 1) - x0) < fabs(xf(kc) - x0)) --kc;
        if (is3(kc)) {
          int lo = clo, hi = kc; // smallest k in [clo, kc] with is3 (true ... true towards kc)
          while (lo < hi) {
            const int m = (lo + hi) >> 1;
            if (is3(m)) hi = m; else lo = m + 1;
          }
          k3lo = lo;
          lo = kc; hi = chi;     // largest k in [kc, chi] with is3
          while (lo < hi) {
            const int m = (lo + hi + 1) >> 1;
            if (is3(m)) lo = m; else hi = m - 1;
          }
          k3hi = lo;
        }
      }
    }
    c.k3 = (uint32_t)k3lo | ((uint32_t)k3hi << 16);
  }
  return c;
}

// The derived values of a cold record (see ColdRec).
__device__ inline double cold_ryf(double ry) { return (double)(float)ry; }
__device__ inline double cold_inv_dwp(double dwp) { return fast_rcp<2>(dwp); }
__device__ inline ColdFull expand_cold(const ColdRec &c) {
  ColdFull f;
  f.ry = c.ry;
  f.ryf = cold_ryf(c.ry);
  f.dwp = c.dwp;
  f.inv_dwp = cold_inv_dwp(c.dwp);
  f.x0 = c.x0;
  f.xs2l = c.xs2l;
  f.xs2r = c.xs2r;
  region2_coef(c.ry, f.q2);
  f.il2ir2 = c.il2ir2;
  f.k3 = c.k3;
  return f;
}

// Value of humliv_bb at 1-based index k (1..n) for one (line, layer), any region;
// follows the write order of lineshape.f:455-562 (last writer wins).
template <class XF>
__device__ inline double humliv_point(int k, const FastRec &r, const ColdFull &z, const XF &xf) {
  const int il = r.il(), ir = r.ir(), il2 = z.il2(), ir2 = z.ir2();
  const int il2a = (il2 == il) ? il - 1 : il2; // lineshape.f:524-525
  const int ir2a = (ir2 == ir) ? ir + 1 : ir2;
  if (k > il2a && k < ir2a) { // :526-562
    // rx = |x(k)-x0|/dw, correctly rounded: q0 = a*(1/dw), one residual correction
    const double a = fabs(xf(k) - z.x0);
    double rx = a * z.inv_dwp;
    rx = fma(fma(-z.dwp, rx, a), z.inv_dwp, rx);
    // region 3 / 4 by the record's interval, as the zones kernel does: make_cold found it with the reference's own
    // per-point test (:528), so it IS that test -- and with frozen boundaries (sr_lineset_set_bounds_temps) it is the
    // interval of the boundary temperature in every mode (a per-point test here made the exact mode differ from the
    // far-field modes by a region seam, 1e-5..1e-4, at the one or two points where the interval moves)
    const double b = (double)(float)(-rx);
    return (k >= z.k3lo() && k <= z.k3hi()) ? core_region3(z.ryf, b) : core_region4(z.ryf, b);
  }
  if (il < il2 && k >= il && k <= il2) // :503-512
    return region2_val(z.q2, fma(-(double)(k - il), r.xstep, z.xs2l));
  if (ir2 < ir && k >= ir2 && k <= ir) // :513-522
    return region2_val(z.q2, fma((double)(k - ir2), r.xstep, z.xs2r));
  const double x = (k <= il) ? fma(-(double)(k - 1), r.xstep, r.xl)  // :461-468
                             : fma((double)(k - ir), r.xstep, r.xr); // :470-477
  const double x2 = x * x;
  const R1Coef q = r1_of(r);
  return fma(x2, q.b, q.a) * fast_rcp<2>(fma(x2, fma(x2, 4.0, q.d), q.c));
}

} // namespace sr
