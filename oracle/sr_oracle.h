/*
 * sr_oracle.h -- CPU restatement of the SpectRobot spectral hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call it, and there only as the checker.  The product (spectrobot_amd/)
 * never links or imports this file.
 *
 * Parity status: PINNED.  Every function below is checked against the
 * reference's own Fortran (compiled from /root/reference by oracle/Makefile
 * into oracle/_ref/) and against the reference's Python (spect_classes.py
 * imported under Python 3) through the fixtures in tests/golden/, see
 * tests/golden/make_golden.py.  The radiance recursion (sro_radiance_*) is the
 * one exception: its reference source (spect_base_module) is not in the
 * reference tree, so it is "parity unpinned" and checked analytically only.
 *
 * Citations are file:line into the reference tree.
 */
#ifndef SR_ORACLE_H
#define SR_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define SRO_IMXSIG 13010 /* parameters.inc:65, spect_classes.py:27 */

/* Physical constants exactly as spect_classes.py:44-47 computes them from
 * scipy.constants (CODATA-2018 in scipy 1.15.3). */
double sro_h_cgs(void);
double sro_c_cgs(void);
double sro_k_cgs(void);
double sro_c2(void);

/* lineshape.f:226-569  humliv_bb(x,i1,i2,x0,lw,dw,y); i1,i2 are 1-based and
 * inclusive; x and y hold at least i2 doubles.  Returns 0, or -1 / -2 where the
 * Fortran would `stop` (lineshape.f:253-256, 260-264). */
int sro_humliv_bb(const double *x, int i1, int i2, double x0, double lw,
                  double dw, double *y);

/* lineshape.f:150-205  scalar humli_bb (cross-check only). */
double sro_humli_bb(double rx, double ry);

/* lineshape.f:2-25  sum_all_lines: spe[j] += rows[ilin][i] for j=init..fin
 * (1-based, inclusive).  rows is row-major [n_lines][row_len]. */
void sro_sum_all_lines(double *spe, long n_spe, const double *rows,
                       const int *init, const int *fin, int n_lines,
                       int row_len);

/* spect_classes.py:1967-1974, 1976-1986, 2029-2036 */
double sro_convert_to_atm(double pres_hpa);
double sro_lorenz_width(double temp, double pres_atm, double t_dep_broad,
                        double air_broad);
double sro_doppler_width(double temp, double mm, double wn0);

/* spect_classes.py:1990-2008 MakeShape on a 13010-point window. */
int sro_make_shape(const double *xwin, int n, double wn0, double lw, double dw,
                   double *shape);

/* spect_classes.py:1937-1943 closest_grid (first arg-min). */
long sro_closest_grid(const double *grid, long n_grid, double wn0);

/* spect_classes.py:1736-1754, 1778-1785, 1806-1853, 312-343:
 * G[0]=sp_emission, G[1]=ind_emission, G[2]=absorption. */
void sro_calc_gcoeffs(double freq, double a_coeff, double e_lower, double g_up,
                      double g_lo, double e_vib_up, double e_vib_lo,
                      double temp, double G[3]);

/* spect_classes.py:1856-1863 (LTE identity known-answer test). */
double sro_linestrength_hitran(double a_coeff, double wn, double temp,
                               double q_part, double g_upper, double e_lower);

/* spect_classes.py:1876-1878, 1895-1903 */
double sro_boltz_ratio_nodeg(double wn, double temp);
double sro_calc_bb_single(double nu, double temp);

/* spect_classes.py:1692-1710 CalcPartitionSum given the TIPS-2003 table
 * (fparts_mod.f:33-295): 4-point scipy.interpolate.lagrange restated. */
double sro_calc_partition_sum(const double *t_grid, const double *q_grid,
                              int n_tab, double temp);

/* curgods.f:2-98 */
double sro_curgod_1(const double *nd, const double *x, int n_p);
double sro_curgod_2(const double *nd, const double *vmr, const double *x,
                    int n_p);
double sro_curgod_3(const double *nd, const double *vmr, const double *f,
                    const double *x, int n_p);
double sro_curgod_4(const double *nd, const double *vmr, const double *f,
                    const double *x, int n_p);

/* Line list, structure of arrays (one iso-molecule). lev_up / lev_lo are
 * indices into the level table, or -1 when the line's level is not in the
 * table (such lines are dropped when n_levels > 0: spect_classes.py:1384-1388,
 * 122-150; so are lines with lev_up == lev_lo, by the if/elif at 137-142). */
typedef struct {
  long n_lines;
  const double *freq, *a_coeff, *e_lower, *g_up, *g_lo, *air_broad,
      *t_dep_broad;
  const int *lev_up, *lev_lo;
} sro_lines;

/* Per-layer absorption / emission coefficients, the `useLUTs=False` path of
 * spect_main_module.py:1880-2131: calc_shapes_lines (spect_classes.py:1378-1462)
 * + LutSet.add_PT (spect_main_module.py:1122-1168) + BuildCoeff
 * (spect_classes.py:1277-1337) + the population-weighted combine
 * (spect_main_module.py:2036-2106).
 *
 * grid = w0 + step*j, j<n_grid (prepare_spe_grid, spect_main_module.py:1262-1272)
 * temps/press[n_layers] (K, hPa); q_part[n_layers] = CalcPartitionSum(T);
 * e_lev[n_levels]; tvib[n_levels*n_layers] (level-major) or NULL for LTE.
 * mode 0: faithful -- materialise per-level, per-ctype G spectra then combine.
 * mode 1: direct   -- fold the level populations into per-line weights and
 *                     accumulate abs/emi straight away (same maths, used for
 *                     the CPU timing leg).
 * abs_out/emi_out: [n_layers][n_grid].  n_threads>1 splits layers over
 * pthreads. Returns 0 or a negative error code. */
int sro_abscoeff_layers(const sro_lines *L, double mm, int n_levels,
                        const double *e_lev, int n_layers, const double *temps,
                        const double *press, const double *q_part,
                        const double *tvib, const double *grid, long n_grid,
                        int mode, int n_threads, double *abs_out,
                        double *emi_out);

/* As sro_abscoeff_layers in mode 0, also returning the per-level, per-ctype G-coefficient spectra it
 * builds on the way -- what LutSet.add_PT / SpectralGcoeff.BuildCoeff produce (spect_main_module.py:1122-1168,
 * spect_classes.py:1277-1337): g_out[n_layers][n_set][3][n_grid], n_set = max(n_levels, 1), ctype 0
 * sp_emission, 1 ind_emission, 2 absorption. */
int sro_gcoeff_layers(const sro_lines *L, double mm, int n_levels,
                      const double *e_lev, int n_layers, const double *temps,
                      const double *press, const double *q_part,
                      const double *tvib, const double *grid, long n_grid,
                      int n_threads, double *abs_out, double *emi_out,
                      double *g_out);

/* Build's own definition of the limb radiance recursion (reference source
 * absent: parity unpinned).  For one ray crossing n_seg segments in photon
 * order; seg_layer[s] indexes the layer whose coefficients apply, col[s] is
 * the absorber column (molecules cm^-2, already times iso abundance):
 *   tau = abs*col;  I <- I*exp(-tau) + (emi/abs)*(1-exp(-tau))
 * with emi/abs -> emi*col when |tau| tiny.  rad[n_grid] in, out. */
void sro_radiance_ray(const double *abs_c, const double *emi_c, long n_grid,
                      int n_seg, const int *seg_layer, const double *col,
                      double *rad);

/* N2, spect_classes.py:1180-1191 hires_to_lowres: cm-1 -> nm, Gaussian ILS (n_sigma
 * sigmas, np.trapz on the irregular nm grid), unit conversion from 'ergscm2'
 * (out_units 0 = Wm2, 1 = ergscm2, 2 = nWcm2).  out[nb]. */
void sro_hires_to_lowres(const double *grid_cm, const double *spec, long n, const double *cen_nm,
                         const double *wid_nm, int nb, double n_sigma, int out_units, double *out);

#ifdef __cplusplus
}
#endif
#endif
