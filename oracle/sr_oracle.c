/*
 * sr_oracle.c -- CPU restatement of the SpectRobot spectral hot path.
 * TEST INFRASTRUCTURE ONLY (see sr_oracle.h). Parity: PINNED against the
 * reference Fortran/Python via tests/golden/ (radiance recursion excepted).
 *
 * Written from the reference's behaviour, statement by statement where the
 * floating-point result depends on it; citations are file:line into the
 * reference tree.
 */
#include "sr_oracle.h"

#include <complex.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* constants: spect_classes.py:40-47 with scipy 1.15.3 (CODATA-2018)   */
/* ------------------------------------------------------------------ */
#define SRO_PI 3.141592653589793
static const double T_REF = 296.0;                       /* spcl:39 */
static const double HPA_TO_ATM = 0.00098692326671601;    /* spcl:40 */
static const double AVOGADRO = 6.02214076e23;            /* scipy.constants */

double sro_h_cgs(void) { return 6.62607015e-34 * 1.e7; } /* spcl:44 */
double sro_c_cgs(void) { return 299792458.0 * 1.e2; }    /* spcl:45 */
double sro_k_cgs(void) { return 1.380649e-23 * 1.e7; }   /* spcl:46 */
double sro_c2(void) { return sro_h_cgs() * sro_c_cgs() / sro_k_cgs(); }

/* un-suffixed Fortran literal: rounded to single, then widened */
#define F32(lit) ((double)(lit##f))

static long nintl(double v) { return lround(v); } /* Fortran nint */
static long lmax(long a, long b) { return a > b ? a : b; }
static long lmin(long a, long b) { return a < b ? a : b; }

/* ------------------------------------------------------------------ */
/* Humlicek regions                                                    */
/* ------------------------------------------------------------------ */
/* lineshape.f:529-560 (also 277-311, 363-398): one core point.
 * cmplx(ry,-rx) is default-kind => both parts rounded to single; the
 * polynomial literals carry no D0 => rounded to single. */
static double core_point(double rx, double ry) {
  double r2 = (0.195 * rx) - 0.176;
  double complex c2 = (double)(float)ry + I * (double)(float)(-rx);
  if (ry < r2) { /* region 4 */
    double complex c1 = c2 * c2;
    double complex num =
        c2 *
        (F32(36183.31) -
         c1 * (F32(3321.9905) -
               c1 * (F32(1540.787) -
                     c1 * (F32(219.0313) -
                           c1 * (F32(35.76683) -
                                 c1 * (F32(1.320522) - c1 * F32(.56419)))))));
    double complex den =
        (F32(32066.6) -
         c1 * (F32(24322.84) -
               c1 * (F32(9022.228) -
                     c1 * (F32(2186.181) -
                           c1 * (F32(364.2191) -
                                 c1 * (F32(61.57037) -
                                       c1 * (F32(1.841439) - c1)))))));
    double complex c3 = num / den;
    return exp(creal(c1)) * cos(cimag(c1)) - creal(c3);
  } else { /* region 3 */
    double complex num =
        (F32(16.4955) +
         c2 * (F32(20.20933) +
               c2 * (F32(11.96482) + c2 * (F32(3.778987) + c2 * F32(.5642236)))));
    double complex den =
        (F32(16.4955) +
         c2 * (F32(38.82363) +
               c2 * (F32(39.27121) +
                     c2 * (F32(21.69274) + c2 * (F32(6.699398) + c2)))));
    return creal(num / den);
  }
}

typedef struct {
  double a, b, c, d, e, f, g, h;
} r2coef;

/* lineshape.f:492-502 (also 321-331, 408-418) */
static r2coef region2_coef(double ry, double ry2) {
  r2coef q;
  q.a = ry * (1.0578555 + ry2 * (4.6545642 + ry2 * (3.1030428 + 0.5641896 * ry2)));
  q.b = ry * (2.9619954 + ry2 * (0.5641896 + 1.6925688 * ry2));
  q.c = ry * (-2.5388532 + ry2 * 1.6925688);
  q.d = ry * 0.5641896;
  q.e = 0.5625 + ry2 * (4.5 + ry2 * (10.5 + ry2 * (6. + ry2)));
  q.f = -4.5 + ry2 * (9. + ry2 * (6. + 4. * ry2));
  q.g = 10.5 + ry2 * (-6. + 6. * ry2);
  q.h = 4. * ry2 - 6.;
  return q;
}
static double region2_val(const r2coef *q, double x2) {
  return (q->a + x2 * (q->b + x2 * (q->c + q->d * x2))) /
         (q->e + x2 * (q->f + x2 * (q->g + x2 * (q->h + x2))));
}

typedef struct {
  double a, b, c, d, e;
} r1coef;
/* lineshape.f:456-460 (also 344-348, 429-433) */
static r1coef region1_coef(double ry, double ry2) {
  r1coef q;
  q.a = ry * (1.1283792 + 2.2567584 * ry2);
  q.b = 2.2567584 * ry;
  q.c = (1. + 2. * ry2) * (1. + 2. * ry2);
  q.d = -4. + 8. * ry2;
  q.e = 4.;
  return q;
}
static double region1_val(const r1coef *q, double x2) {
  return (q->a + x2 * q->b) / (q->c + x2 * (q->d + q->e * x2));
}

#define X(k) x[(k)-1]
#define Y(k) y[(k)-1]

int sro_humliv_bb(const double *x, int i1, int i2, double x0, double lw,
                  double dw, double *y) {
  long j, k, l, ir, ir2, il, il2;
  double rx, ry, tst, xrun, xstep, x2, ry2;

  if (i1 > i2) return -1;   /* lineshape.f:253-256 */
  if (!(dw > 0.0)) return -2; /* lineshape.f:260-264 */
  ry = lw / dw;
  xstep = (X(i1 + 1) - X(i1)) / dw; /* :265-266 */
  ry2 = ry * ry;

  if (x0 <= X(i1)) { /* forward loop, lineshape.f:272-357 */
    tst = 5.5;
    j = i1;
    rx = (X(j) - x0) / dw;
    while ((rx + ry < tst) && (j <= i2)) {
      Y(j) = core_point(rx, ry);
      j = j + 1;
      rx = rx + xstep;
    }
    if (j <= i2) {
      tst = 15.0;
      l = lmax(nintl((tst - ry - rx) / xstep), 0) + j;
      l = lmin(l, i2);
      if (l > j) {
        r2coef q = region2_coef(ry, ry2);
        xrun = (X(j) - x0) / dw;
        for (k = j; k <= l; k++) {
          x2 = xrun * xrun;
          Y(k) = region2_val(&q, x2);
          xrun = xrun + xstep;
        }
        l = l + 1;
      }
      if (l < j) l = j;
      if (l < i2) {
        r1coef q = region1_coef(ry, ry2);
        xrun = (X(l) - x0) / dw;
        for (k = l; k <= i2; k++) {
          x2 = xrun * xrun;
          Y(k) = region1_val(&q, x2);
          xrun = xrun + xstep;
        }
      }
    }
  } else if (x0 >= X(i2)) { /* lineshape.f:358-442 */
    tst = 5.5;
    j = i2;
    rx = (x0 - X(j)) / dw;
    while ((rx + ry < tst) && (j >= i1)) {
      Y(j) = core_point(rx, ry);
      j = j - 1;
      rx = rx + xstep;
    }
    if (j >= i1) {
      tst = 15.0;
      l = j - lmax(nintl((tst - ry - rx) / dw / xstep), 0); /* sic, :404 */
      l = lmax(l, i1);
      if (l == i2) l = i2 + 1;
      if (l < j) {
        r2coef q = region2_coef(ry, ry2);
        xrun = (x0 - X(l)) / dw;
        for (k = l; k <= j; k++) {
          x2 = xrun * xrun;
          Y(k) = region2_val(&q, x2);
          xrun = xrun - xstep;
        }
      }
      if (l >= i1) {
        r1coef q = region1_coef(ry, ry2);
        xrun = (x0 - X(i1)) / dw;
        for (k = i1; k <= l - 1; k++) {
          x2 = xrun * xrun;
          Y(k) = region1_val(&q, x2);
          xrun = xrun - xstep;
        }
      }
    }
  } else { /* x(i1) < x0 < x(i2), lineshape.f:443-562 */
    rx = (x0 - X(i1)) / dw;
    tst = 15.;
    il = i1;
    if (rx + ry >= tst) il = lmax(nintl((rx - ry - tst) / xstep), 0) + i1;
    rx = (X(i2) - x0) / dw;
    ir = i2;
    if (rx + ry >= tst) ir = i2 - lmax(nintl((rx - ry - tst) / xstep), 0);
    if (il > i1 || ir < i2) {
      r1coef q = region1_coef(ry, ry2);
      if (il > i1) {
        xrun = (x0 - X(i1)) / dw;
        for (k = i1; k <= il; k++) {
          x2 = xrun * xrun;
          Y(k) = region1_val(&q, x2);
          xrun = xrun - xstep;
        }
      }
      if (ir < i2) {
        xrun = (X(ir) - x0) / dw;
        for (k = ir; k <= i2; k++) {
          x2 = xrun * xrun;
          Y(k) = region1_val(&q, x2);
          xrun = xrun + xstep;
        }
      }
    }
    rx = (x0 - X(il)) / dw;
    tst = 5.5;
    il2 = il;
    if (rx + ry >= tst) il2 = il + lmax(nintl((rx - ry - tst) / xstep), 0);
    ir2 = ir;
    rx = (X(ir) - x0) / dw;
    if (rx + ry >= tst) ir2 = ir - lmax(nintl((rx - ry - tst) / xstep), 0);
    if (il2 > il || ir2 < ir) {
      r2coef q = region2_coef(ry, ry2);
      if (il < il2) {
        xrun = (x0 - X(il)) / dw;
        for (j = il; j <= il2; j++) {
          x2 = xrun * xrun;
          Y(j) = region2_val(&q, x2);
          xrun = xrun - xstep;
        }
      }
      if (ir2 < ir) {
        xrun = (X(ir2) - x0) / dw;
        for (j = ir2; j <= ir; j++) {
          x2 = xrun * xrun;
          Y(j) = region2_val(&q, x2);
          xrun = xrun + xstep;
        }
      }
    }
    if (il2 == il) il2 = il - 1;
    if (ir2 == ir) ir2 = ir + 1;
    for (j = il2 + 1; j <= ir2 - 1; j++) {
      rx = fabs(X(j) - x0) / dw;
      Y(j) = core_point(rx, ry);
    }
  }
  return 0;
}

/* lineshape.f:150-205: all literals D0, c2 still through default cmplx(). */
double sro_humli_bb(double rx, double ry) {
  double r1 = fabs(rx) + ry;
  double r2 = (0.195 * fabs(rx)) - 0.176;
  double complex c1, c3;
  double complex c2 = (double)(float)ry + I * (double)(float)(-rx);
  if (r1 >= 15.0) {
    c3 = c2 * 0.5641896 / (0.5 + (c2 * c2));
    return creal(c3);
  } else if (r1 >= 5.5) {
    c1 = c2 * c2;
    c3 = c2 * (1.410474 + c1 * .5641896) / (.75 + c1 * (3. + c1));
    return creal(c3);
  } else if (ry >= r2) {
    c3 = (16.4955 +
          c2 * (20.20933 + c2 * (11.96482 + c2 * (3.778987 + c2 * .5642236)))) /
         (16.4955 +
          c2 * (38.82363 +
                c2 * (39.27121 + c2 * (21.69274 + c2 * (6.699398 + c2)))));
    return creal(c3);
  } else {
    c1 = c2 * c2;
    c3 = c2 *
         (36183.31 -
          c1 * (3321.9905 -
                c1 * (1540.787 -
                      c1 * (219.0313 -
                            c1 * (35.76683 - c1 * (1.320522 - c1 * .56419)))))) /
         (32066.6 -
          c1 * (24322.84 -
                c1 * (9022.228 -
                      c1 * (2186.181 -
                            c1 * (364.2191 -
                                  c1 * (61.57037 - c1 * (1.841439 - c1)))))));
    return exp(creal(c1)) * cos(cimag(c1)) - creal(c3);
  }
}

/* lineshape.f:15-23 */
void sro_sum_all_lines(double *spe, long n_spe, const double *rows,
                       const int *init, const int *fin, int n_lines,
                       int row_len) {
  (void)n_spe;
  for (int ilin = 0; ilin < n_lines; ilin++) {
    long i = 0;
    for (long j = init[ilin]; j <= fin[ilin]; j++) {
      spe[j - 1] = spe[j - 1] + rows[(long)ilin * row_len + i];
      i = i + 1;
    }
  }
}

/* ------------------------------------------------------------------ */
/* spect_classes.py scalar functions                                   */
/* ------------------------------------------------------------------ */
double sro_convert_to_atm(double pres_hpa) { return pres_hpa * HPA_TO_ATM; }

/* spcl:1972 with Self_broad = Self_pres_atm = 0.0 (defaults, spcl:190) */
double sro_lorenz_width(double temp, double pres_atm, double t_dep_broad,
                        double air_broad) {
  return pow(T_REF / temp, t_dep_broad) *
         (air_broad * (pres_atm - 0.0) + 0.0 * 0.0);
}

/* spcl:1984 */
double sro_doppler_width(double temp, double mm, double wn0) {
  return wn0 / sro_c_cgs() *
         sqrt(2 * AVOGADRO * sro_k_cgs() * temp * log(2.0) / mm);
}

/* spcl:1997-2003, Strength = 1.0 */
int sro_make_shape(const double *xwin, int n, double wn0, double lw, double dw,
                   double *shape) {
  double fac = dw * sqrt(SRO_PI / log(2.0));
  int rc = sro_humliv_bb(xwin, 1, n, wn0, lw, dw / sqrt(log(2.0)), shape);
  if (rc) return rc;
  for (int i = 0; i < n; i++) shape[i] = 1.0 * shape[i] / fac;
  return 0;
}

/* spcl:1941: np.argmin(np.abs(grid-wn0)) -- first minimum */
long sro_closest_grid(const double *grid, long n_grid, double wn0) {
  long best = 0;
  double bv = fabs(grid[0] - wn0);
  for (long i = 1; i < n_grid; i++) {
    double v = fabs(grid[i] - wn0);
    if (v < bv) {
      bv = v;
      best = i;
    }
  }
  return best;
}

double sro_boltz_ratio_nodeg(double wn, double temp) { /* spcl:1877 */
  return exp(-sro_c2() * wn / temp);
}

double sro_calc_bb_single(double nu, double temp) { /* spcl:1901 */
  double c = sro_c_cgs();
  return 2 * sro_h_cgs() * (c * c) * (nu * nu * nu) /
         (exp(sro_c2() * nu / temp) - 1);
}

/* spcl:312-343 with E_vib resolved by the caller (LinkToMolec, spcl:122-150) */
void sro_calc_gcoeffs(double freq, double a_coeff, double e_lower, double g_up,
                      double g_lo, double e_vib_up, double e_vib_lo,
                      double temp, double G[3]) {
  if (!(a_coeff != 0.0 && g_lo != 0.0 && g_up != 0.0)) { /* spcl:326,337 */
    G[0] = G[1] = G[2] = 0.0;
    return;
  }
  double h = sro_h_cgs(), c = sro_c_cgs();
  /* spcl:1743,1750: fact_2 = 2*h_cgs*c_cgs**2*wavenumber**3 */
  double fact_2 = 2 * h * (c * c) * pow(freq, 3.0);
  double B21 = a_coeff / fact_2;
  double four_pi = 4 * SRO_PI;
  /* sp_emission spcl:1850-1851 */
  double rot_up = g_up * sro_boltz_ratio_nodeg(e_lower + freq - e_vib_up, temp);
  G[0] = h * c * freq * rot_up * a_coeff / four_pi;
  /* ind_emission spcl:1837-1840 */
  G[1] = h * c * freq * rot_up * B21 / four_pi;
  /* absorption spcl:1812-1817; B_12 = B_21*g_2/g_1 spcl:1783 */
  double B12 = B21 * g_up / g_lo;
  double rot_lo = g_lo * sro_boltz_ratio_nodeg(e_lower - e_vib_lo, temp);
  G[2] = h * c * freq * rot_lo * B12 / four_pi;
}

double sro_linestrength_hitran(double a_coeff, double wn, double temp,
                               double q_part, double g_upper, double e_lower) {
  /* spcl:1861, iso_ab = 1 */
  double c2 = sro_c2();
  return 1.0 * a_coeff * g_upper * exp(-c2 * e_lower / temp) *
         (1 - exp(-c2 * wn / temp)) /
         (8 * SRO_PI * sro_c_cgs() * (wn * wn) * q_part);
}

/* spcl:1698-1708.  scipy.interpolate.lagrange builds the polynomial in
 * coefficient form with poly1d products and evaluates it by Horner; the same
 * sequence is followed here so that rounding agrees. */
double sro_calc_partition_sum(const double *t_grid, const double *q_grid,
                              int n_tab, double temp) {
  double xs[4], qs[4];
  int m = 0;
  /* T_grid[T_grid <= temp][-2:] */
  int n_le = 0;
  for (int i = 0; i < n_tab; i++)
    if (t_grid[i] <= temp) n_le++; /* table is increasing */
  int lo0 = n_le - 2 < 0 ? 0 : n_le - 2;
  for (int i = lo0; i < n_le; i++) {
    xs[m] = t_grid[i];
    qs[m] = q_grid[i];
    m++;
  }
  /* T_grid[T_grid > temp][:2] */
  for (int i = n_le; i < n_tab && i < n_le + 2; i++) {
    xs[m] = t_grid[i];
    qs[m] = q_grid[i];
    m++;
  }
  /* p = sum_j w_j * prod_{k!=j} poly1d([1,-x_k])/(x_j-x_k); highest power first */
  double p[4] = {0, 0, 0, 0};
  int plen = 1; /* poly1d(0.0) */
  for (int j = 0; j < m; j++) {
    double pt[4];
    int ptlen = 1;
    pt[0] = qs[j];
    for (int k = 0; k < m; k++) {
      if (k == j) continue;
      double fac = xs[j] - xs[k];
      /* pt *= poly1d([1.0, -x_k]) / fac   (division first: poly1d/scalar) */
      double d0 = 1.0 / fac, d1 = -xs[k] / fac;
      double nw[4];
      for (int i = 0; i < ptlen + 1; i++) {
        double s = 0.0;
        /* np.convolve(pt, [d0,d1]) */
        if (i < ptlen) s += pt[i] * d0;
        if (i - 1 >= 0 && i - 1 < ptlen) s += pt[i - 1] * d1;
        nw[i] = s;
      }
      ptlen++;
      memcpy(pt, nw, sizeof(double) * ptlen);
    }
    /* p += pt (poly1d add aligns the low-order ends) */
    if (ptlen > plen) {
      double tmp[4] = {0, 0, 0, 0};
      for (int i = 0; i < plen; i++) tmp[ptlen - plen + i] = p[i];
      memcpy(p, tmp, sizeof(tmp));
      plen = ptlen;
    }
    for (int i = 0; i < ptlen; i++) p[plen - ptlen + i] += pt[i];
  }
  double yv = 0.0; /* np.polyval */
  for (int i = 0; i < plen; i++) yv = yv * temp + p[i];
  return yv;
}

/* ------------------------------------------------------------------ */
/* curgods.f                                                           */
/* ------------------------------------------------------------------ */
double sro_curgod_1(const double *nd, const double *x, int n_p) {
  double res = 0.0;
  for (int i = 0; i < n_p - 1; i++) {
    double dx = x[i + 1] - x[i];
    double fu = nd[i + 1] / nd[i];
    double D = log(fu) / dx;
    res = res + (nd[i + 1] - nd[i]) / D;
  }
  return res;
}

double sro_curgod_2(const double *nd, const double *vmr, const double *x,
                    int n_p) {
  double res = 0.0;
  for (int i = 0; i < n_p - 1; i++) {
    double dx = x[i + 1] - x[i];
    double A = nd[i] * vmr[i];
    double B = nd[i] * (vmr[i + 1] - vmr[i]) / dx;
    double fu = nd[i + 1] / nd[i];
    double D = log(fu) / dx;
    res = res + (A * D * (fu - 1.) + B * fu * (D * dx - 1.) + B) / (D * D);
  }
  return res;
}

double sro_curgod_3(const double *nd, const double *vmr, const double *f,
                    const double *x, int n_p) {
  double res = 0.0;
  for (int i = 0; i < n_p - 1; i++) {
    double dx = x[i + 1] - x[i];
    double A = nd[i] * vmr[i] * f[i];
    double cc = (vmr[i + 1] - vmr[i]) / dx;
    double bb = (f[i + 1] - f[i]) / dx;
    double B = nd[i] * (vmr[i] * bb + f[i] * cc);
    double C = nd[i] * bb * cc;
    double fu = nd[i + 1] / nd[i];
    double D = log(fu) / dx;
    res = res + (fu * (D * (A * D + B * (D * dx - 1.)) +
                       C * (D * dx * (D * dx - 2.) + 2.)) +
                 D * (B - A * D) - 2 * C) /
                    (D * D * D);
  }
  return res;
}

double sro_curgod_4(const double *nd, const double *vmr, const double *f,
                    const double *x, int n_p) {
  double res = 0.0;
  for (int i = 0; i < n_p - 1; i++) {
    double dx = x[i + 1] - x[i];
    double A = nd[i] * vmr[i] * f[i];
    double cc = (vmr[i + 1] - vmr[i]) / dx;
    double B = nd[i] * f[i] * cc;
    double fu = nd[i + 1] * f[i + 1] / (nd[i] * f[i]);
    double D = log(fu) / dx;
    res = res + (A * D * (fu - 1.) + B * fu * (D * dx - 1.) + B) / (D * D);
  }
  return res;
}

/* ------------------------------------------------------------------ */
/* per-layer coefficients                                              */
/* ------------------------------------------------------------------ */
typedef struct {
  const sro_lines *L;
  double mm;
  int n_levels;
  const double *e_lev;
  int n_layers;
  const double *temps, *press, *q_part, *tvib, *grid;
  long n_grid;
  int mode;
  double *abs_out, *emi_out;
  double *g_out;        /* mode 0 only: [n_layers][n_set][3][n_grid] per-level G spectra, or NULL */
  const long *ic;       /* closest grid index per line */
  const double *lin_grid; /* np.arange(-imxsig*s/2, imxsig*s/2, s) */
  int layer_lo, layer_hi;
  int rc;
} job_t;

static void layer_run(job_t *J, int k, double *xwin, double *shape,
                      double *gbuf) {
  const sro_lines *L = J->L;
  const long n = J->n_grid;
  const int nlev = J->n_levels;
  const double T = J->temps[k], P = J->press[k];
  const double P_atm = sro_convert_to_atm(P); /* spcl:186 */
  double *abs_k = J->abs_out + (long)k * n;
  double *emi_k = J->emi_out + (long)k * n;
  memset(abs_k, 0, sizeof(double) * n);
  memset(emi_k, 0, sizeof(double) * n);
  const int nset = nlev > 0 ? nlev : 1;
  if (J->mode == 0) memset(gbuf, 0, sizeof(double) * nset * 3 * n);

  /* level populations, smm:2049-2073 (pop = 1/Q for the 'all' set, smm:2054) */
  double pop[64];
  if (nlev > 0) {
    for (int lv = 0; lv < nlev; lv++) {
      double vibt = J->tvib ? J->tvib[(long)lv * J->n_layers + k] : T;
      pop[lv] = sro_boltz_ratio_nodeg(J->e_lev[lv], vibt) / J->q_part[k];
    }
  } else {
    pop[0] = 1 / J->q_part[k];
  }

  for (long i = 0; i < L->n_lines; i++) {
    int lu = 0, ll = 0;
    double evu = 0.0, evl = 0.0;
    if (nlev > 0) {
      lu = L->lev_up[i];
      ll = L->lev_lo[i];
      /* spcl:1384-1388 keeps a line only if LinkToMolec found BOTH levels; its
       * if/elif (spcl:137-142) never sets the lower level when it is the same
       * level as the upper one, so such lines are dropped too. */
      if (lu < 0 || ll < 0 || lu == ll) continue;
      evu = J->e_lev[lu];
      evl = J->e_lev[ll];
    }
    /* spcl:1454-1457 */
    long ic = J->ic[i];
    double fr_grid_ok = J->grid[ic];
    for (int m = 0; m < SRO_IMXSIG; m++) xwin[m] = J->lin_grid[m] + fr_grid_ok;
    double lw = sro_lorenz_width(T, P_atm, L->t_dep_broad[i], L->air_broad[i]);
    double dw = sro_doppler_width(T, J->mm, L->freq[i]);
    int rc = sro_make_shape(xwin, SRO_IMXSIG, L->freq[i], lw, dw, shape);
    if (rc) {
      J->rc = rc;
      return;
    }
    double G[3];
    sro_calc_gcoeffs(L->freq[i], L->a_coeff[i], L->e_lower[i], L->g_up[i],
                     L->g_lo[i], evu, evl, T, G);
    /* window -> grid overlap: spcl:1113-1120 (window point m sits on grid
     * point ic-6505+m; only the overlapping part is added). */
    long j0 = ic - SRO_IMXSIG / 2;
    long mlo = j0 < 0 ? -j0 : 0;
    long mhi = j0 + SRO_IMXSIG > n ? n - j0 : SRO_IMXSIG;
    if (J->mode == 0) {
      /* BuildCoeff per level and ctype, spcl:1304-1327; Strength*shape spcl:1041 */
      double *gsp = gbuf + ((long)lu * 3 + 0) * n;
      double *gin = gbuf + ((long)lu * 3 + 1) * n;
      double *gab = gbuf + ((long)ll * 3 + 2) * n;
      for (long m = mlo; m < mhi; m++) {
        gsp[j0 + m] += shape[m] * G[0];
        gin[j0 + m] += shape[m] * G[1];
        gab[j0 + m] += shape[m] * G[2];
      }
    } else {
      double wabs = pop[ll] * G[2] - pop[lu] * G[1];
      double wemi = pop[lu] * G[0];
      for (long m = mlo; m < mhi; m++) {
        abs_k[j0 + m] += shape[m] * wabs;
        emi_k[j0 + m] += shape[m] * wemi;
      }
    }
  }
  if (J->mode == 0 && J->g_out)
    memcpy(J->g_out + (long)k * nset * 3 * n, gbuf, sizeof(double) * nset * 3 * n);
  if (J->mode == 0) {
    /* smm:2052-2080 */
    for (int lv = 0; lv < nset; lv++) {
      const double *gsp = gbuf + ((long)lv * 3 + 0) * n;
      const double *gin = gbuf + ((long)lv * 3 + 1) * n;
      const double *gab = gbuf + ((long)lv * 3 + 2) * n;
      for (long j = 0; j < n; j++) {
        abs_k[j] += gab[j] * pop[lv];
        abs_k[j] -= gin[j] * pop[lv];
        emi_k[j] += gsp[j] * pop[lv];
      }
    }
  }
}

static void *worker(void *arg) {
  job_t *J = (job_t *)arg;
  double *xwin = (double *)malloc(sizeof(double) * SRO_IMXSIG);
  double *shape = (double *)malloc(sizeof(double) * SRO_IMXSIG);
  double *gbuf = NULL;
  if (J->mode == 0) {
    int nset = J->n_levels > 0 ? J->n_levels : 1;
    gbuf = (double *)malloc(sizeof(double) * nset * 3 * J->n_grid);
  }
  for (int k = J->layer_lo; k < J->layer_hi && J->rc == 0; k++)
    layer_run(J, k, xwin, shape, gbuf);
  free(xwin);
  free(shape);
  free(gbuf);
  return NULL;
}

static int abscoeff_impl(const sro_lines *L, double mm, int n_levels,
                         const double *e_lev, int n_layers, const double *temps,
                         const double *press, const double *q_part,
                         const double *tvib, const double *grid, long n_grid,
                         int mode, int n_threads, double *abs_out,
                         double *emi_out, double *g_out);

int sro_abscoeff_layers(const sro_lines *L, double mm, int n_levels,
                        const double *e_lev, int n_layers, const double *temps,
                        const double *press, const double *q_part,
                        const double *tvib, const double *grid, long n_grid,
                        int mode, int n_threads, double *abs_out,
                        double *emi_out) {
  return abscoeff_impl(L, mm, n_levels, e_lev, n_layers, temps, press, q_part, tvib, grid, n_grid, mode,
                       n_threads, abs_out, emi_out, NULL);
}

int sro_gcoeff_layers(const sro_lines *L, double mm, int n_levels,
                      const double *e_lev, int n_layers, const double *temps,
                      const double *press, const double *q_part,
                      const double *tvib, const double *grid, long n_grid,
                      int n_threads, double *abs_out, double *emi_out,
                      double *g_out) {
  return abscoeff_impl(L, mm, n_levels, e_lev, n_layers, temps, press, q_part, tvib, grid, n_grid, 0,
                       n_threads, abs_out, emi_out, g_out);
}

static int abscoeff_impl(const sro_lines *L, double mm, int n_levels,
                         const double *e_lev, int n_layers, const double *temps,
                         const double *press, const double *q_part,
                         const double *tvib, const double *grid, long n_grid,
                         int mode, int n_threads, double *abs_out,
                         double *emi_out, double *g_out) {
  if (n_levels > 64 || n_grid < 2) return -3;
  long *ic = (long *)malloc(sizeof(long) * (L->n_lines > 0 ? L->n_lines : 1));
  for (long i = 0; i < L->n_lines; i++) {
    /* spcl:1941; grids are increasing so the arg-min sits next to the
     * insertion point: scan a small neighbourhood, first minimum wins. */
    long lo = 0, hi = n_grid;
    while (lo < hi) {
      long mid = (lo + hi) / 2;
      if (grid[mid] < L->freq[i]) lo = mid + 1; else hi = mid;
    }
    long a = lo - 2 < 0 ? 0 : lo - 2, b = lo + 2 > n_grid ? n_grid : lo + 2;
    long best = a;
    double bv = fabs(grid[a] - L->freq[i]);
    for (long j = a + 1; j < b; j++) {
      double v = fabs(grid[j] - L->freq[i]);
      if (v < bv) { bv = v; best = j; }
    }
    ic[i] = best;
  }
  /* spcl:1445-1446: numpy arange = start + i*delta, delta = (start+step)-start */
  double sp_step = grid[1] - grid[0];
  double start = -SRO_IMXSIG * sp_step / 2;
  double delta = (start + sp_step) - start;
  double *lin_grid = (double *)malloc(sizeof(double) * SRO_IMXSIG);
  for (int m = 0; m < SRO_IMXSIG; m++) lin_grid[m] = start + m * delta;

  if (n_threads < 1) n_threads = 1;
  if (n_threads > n_layers) n_threads = n_layers > 0 ? n_layers : 1;
  job_t *jobs = (job_t *)calloc(n_threads, sizeof(job_t));
  pthread_t *th = (pthread_t *)calloc(n_threads, sizeof(pthread_t));
  /* interleave-free contiguous split; layers cost about the same */
  for (int t = 0; t < n_threads; t++) {
    job_t *J = &jobs[t];
    J->L = L; J->mm = mm; J->n_levels = n_levels; J->e_lev = e_lev;
    J->n_layers = n_layers; J->temps = temps; J->press = press;
    J->q_part = q_part; J->tvib = tvib; J->grid = grid; J->n_grid = n_grid;
    J->mode = mode; J->abs_out = abs_out; J->emi_out = emi_out; J->ic = ic;
    J->g_out = g_out;
    J->lin_grid = lin_grid;
    J->layer_lo = (int)((long)n_layers * t / n_threads);
    J->layer_hi = (int)((long)n_layers * (t + 1) / n_threads);
    J->rc = 0;
  }
  if (n_threads == 1) {
    worker(&jobs[0]);
  } else {
    for (int t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, worker, &jobs[t]);
    for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
  }
  int rc = 0;
  for (int t = 0; t < n_threads; t++) if (jobs[t].rc) rc = jobs[t].rc;
  free(jobs); free(th); free(ic); free(lin_grid);
  return rc;
}

/* ------------------------------------------------------------------ */
/* radiance recursion (build's own definition; parity unpinned)        */
/* ------------------------------------------------------------------ */
void sro_radiance_ray(const double *abs_c, const double *emi_c, long n_grid,
                      int n_seg, const int *seg_layer, const double *col,
                      double *rad) {
  for (int s = 0; s < n_seg; s++) {
    const double *a = abs_c + (long)seg_layer[s] * n_grid;
    const double *e = emi_c + (long)seg_layer[s] * n_grid;
    double u = col[s];
    for (long j = 0; j < n_grid; j++) {
      double tau = a[j] * u;
      double em1 = -expm1(-tau); /* 1 - exp(-tau) */
      double src = fabs(tau) > 1e-12 ? (e[j] * u) * (em1 / tau) : e[j] * u;
      rad[j] = rad[j] * exp(-tau) + src;
    }
  }
}

/* ------------------------------------------------------------------ */
/* N2: SpectralIntensity.hires_to_lowres (spect_classes.py:1180-1191)   */
/* ------------------------------------------------------------------ */
/* cm-1 -> nm conversion of grid and spectrum (spcl:404-407, 779-783), Gaussian
 * ILS of convolve_to_grid_from_irregular (spcl:883-918) with gaussian()
 * (spcl:1926-1934) and conv_single = np.trapz (spcl:1162-1164), then convertto
 * (spcl:1200-1235) from 'ergscm2': out_units 0 = Wm2, 1 = ergscm2, 2 = nWcm2. */
void sro_hires_to_lowres(const double *grid_cm, const double *spec, long n, const double *cen_nm,
                         const double *wid_nm, int nb, double n_sigma, int out_units, double *out) {
  double *xn = (double *)malloc(sizeof(double) * n), *yn = (double *)malloc(sizeof(double) * n);
  for (long i = 0; i < n; i++) {
    long j = n - 1 - i; /* [::-1] */
    xn[i] = 1.e7 / grid_cm[j];
    yn[i] = spec[j] * (grid_cm[j] * grid_cm[j]) * 1.e-7;
  }
  for (int b = 0; b < nb; b++) {
    const double f = cen_nm[b], w = wid_nm[b];
    const double lo = f - n_sigma * w, hi = f + n_sigma * w;
    const double fac = 1 / (w * sqrt(2. * SRO_PI));
    double acc = 0.0, xp = 0.0, yp = 0.0;
    int have = 0;
    for (long i = 0; i < n; i++) {
      if (!(xn[i] >= lo && xn[i] <= hi)) continue;
      const double t = (xn[i] - f) / w;
      const double y = yn[i] * (fac * exp(-0.5 * (t * t)));
      if (have) acc += (xn[i] - xp) * (y + yp) / 2.0;
      xp = xn[i];
      yp = y;
      have = 1;
    }
    double v = acc * 1.e-3; /* ergscm2 -> Wm2 */
    if (out_units == 1) v = v * 1.e3;
    if (out_units == 2) v = v * 1.e5;
    out[b] = v;
  }
  free(xn);
  free(yn);
}
