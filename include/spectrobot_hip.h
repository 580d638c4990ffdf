/*
 * spectrobot_hip.h -- C ABI of libspectrobot_hip.so, the MI355X (gfx950) engine
 * for SpectRobot's spectral hot path: per-line Voigt evaluation and per-layer
 * absorption / emission coefficient accumulation, plus the Curtis-Godson column
 * integrals and the limb radiance recursion that consume them.
 *
 * This is the drop-in boundary.  The reference reaches this path through three
 * f2py extension modules (`import lineshape`, `import fparts_mod`, `import
 * curgods`; spect_classes.py:18, 1685 and the absent spect_base_module) and
 * through the Python entry points spect_classes.calc_shapes_lines
 * (spect_classes.py:1378) + LutSet.add_PT (spect_main_module.py:1122) called
 * from make_abscoeff_isomolec(..., useLUTs=False) (spect_main_module.py:1880).
 * Each entry point below names the reference interface it replaces.  Plain
 * pointers and sizes only; no exceptions cross the boundary; every function
 * returns SR_OK (0) or a negative sr_status (the Fortran `stop`s of
 * lineshape.f:253-264 become SR_ERR_ARG).  All buffers are caller-owned.
 *
 * Pointer spaces: functions suffixed `_dev` take DEVICE pointers for the bulk
 * in/out arrays (HBM-resident data path); the others take HOST pointers and
 * stage through the device themselves (compat shims).  `stream` is a
 * hipStream_t passed as void* (NULL = default stream).
 */
#ifndef SPECTROBOT_HIP_H
#define SPECTROBOT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  SR_OK = 0,
  SR_ERR_ARG = -1,      /* bad argument (also: where humliv_bb would `stop`) */
  SR_ERR_LIMIT = -2,    /* a documented size limit exceeded */
  SR_ERR_HIP = -3,      /* HIP runtime error, see sr_last_error() */
  SR_ERR_NODEVICE = -4, /* no gfx950 device visible */
  SR_ERR_UNSUPPORTED = -5,
  SR_ERR_TABLE = -6     /* (mol, iso) not in the TIPS-2003 tables */
} sr_status;

#define SR_IMXSIG 13010 /* parameters.inc:65 -- points per line window */
#define SR_MAX_LEVELS 64

const char *sr_strerror(int status);
const char *sr_last_error(void); /* text of the last HIP failure on this thread */
int sr_abi_version(void);

/* Device selection / query (one process per GPU: call once per rank). */
int sr_set_device(int device);
int sr_device_info(char *name, int name_len, int *cu_count, double *hbm_gib);
/* Hardware queues.  The coefficient op runs on SIX HIP streams (the caller's + five of its own: table preparation, two
 * for the far-field chain, the zones kernel, a copy stream).  The ROCm runtime maps a process's streams onto
 * GPU_MAX_HW_QUEUES hardware queues (environment variable, default 4, read ONCE when HIP initialises, i.e. at the
 * process's first HIP call): with fewer queues than streams two streams share one and kernels no event orders wait for
 * each other (measured: 5.63 instead of 5.47 ms per BASELINE step with 4; 6.01 with 2).  A host application exports
 * GPU_MAX_HW_QUEUES=8 before its first HIP call (the Python package does so at import unless the caller already set it).
 * *recommended = 8; *configured = what this process runs with (the variable's value, 4 when unset).  Returns SR_OK;
 * either pointer may be NULL.  Results do not depend on the setting, only the schedule does. */
int sr_recommended_hw_queues(int *recommended, int *configured);

/* ------------------------------------------------------------------------ *
 * Fine-grained shims: the f2py call shapes, one reference routine each.     *
 * Host pointers.  All evaluated on the GPU.                                  *
 * ------------------------------------------------------------------------ */

/* lineshape.humliv_bb(x, i1, i2, x0, lw, dw) -> y      (lineshape.f:226-569)
 * x[n], y[n]; i1,i2 1-based inclusive.  All three branches: x(i1) < x0 < x(i2)
 * (the only one the reference's Python reaches) in parallel, the two
 * with x0 at or beyond an end as the Fortran's sequential loops. */
int sr_humliv_bb(const double *x, int n, int i1, int i2, double x0, double lw,
                 double dw, double *y);

/* lineshape.sum_all_lines(spe_ini, matrix, init, fin, n_lines, n_spe)
 * (lineshape.f:2-25).  spe[n_spe] is updated in place; rows is row-major
 * [n_lines][row_len] (the Fortran's matrix(ilin, i)); init/fin 1-based. */
int sr_sum_all_lines(double *spe, int64_t n_spe, const double *rows,
                     const int32_t *init, const int32_t *fin, int n_lines,
                     int row_len);

/* fparts_mod.bd_tips_2003(MOL, ISO) -> gi, t_grid[119], QT_grid[119]
 * (fparts_mod.f:33-295). */
int sr_bd_tips_2003(int mol, int iso, double *gi, double *t_grid119,
                    double *qt_grid119);

/* spect_classes.CalcPartitionSum(mol, iso, temp) (spect_classes.py:1692-1710),
 * vectorised over temps[n]. */
int sr_calc_partition_sum(int mol, int iso, const double *temps, int n,
                          double *q_out);

/* curgods.curgod_fort_{1..4}(nd, [vmr, [f,]] x, n_p) -> res (curgods.f:2-98),
 * batched: segment s covers samples off[s] .. off[s+1]-1 of the concatenated
 * arrays; res[n_seg].  vmr / f may be NULL for the variants that do not use
 * them. */
int sr_curgod(int which, const double *nd, const double *vmr, const double *f,
              const double *x, const int32_t *off, int n_seg, double *res);

/* ------------------------------------------------------------------------ *
 * Coarse-grained op: the real hot path.                                      *
 * ------------------------------------------------------------------------ */

/* Line list of one iso-molecule, structure of arrays, host pointers
 * (SpectLine fields, spect_classes.py:50-51). lev_up/lev_lo: index into the
 * level table or -1 (unidentified); ignored when n_levels == 0. */
typedef struct {
  int64_t n_lines;
  const double *freq;        /* Freq        cm^-1 */
  const double *a_coeff;     /* A_coeff     s^-1  */
  const double *e_lower;     /* E_lower     cm^-1 */
  const double *g_up, *g_lo; /* statistical weights */
  const double *air_broad;   /* Air_broad   cm^-1/atm */
  const double *t_dep_broad; /* T_dep_broad */
  const int32_t *lev_up, *lev_lo;
} sr_lines_desc;

/* The iso-molecule the lines belong to (sbm IsoMolec as the path uses it:
 * .mol .iso .MM .levels[].energy). n_levels == 0 is the reference's
 * "unidentified_lines" / 'all' set (spect_main_module.py:1937-1953). */
typedef struct {
  int mol, iso;
  double mm;
  int n_levels;
  const double *level_energy; /* [n_levels] cm^-1 */
} sr_isomolec_desc;

/* Spectral grid as prepare_spe_grid builds it (spect_main_module.py:1262-1272):
 * grid[j] = w0 + j*step (numpy arange), j < n_grid <= 2e6 (imxsig_long). */
typedef struct {
  double w0, step;
  int64_t n_grid;
} sr_grid_desc;

typedef struct sr_lineset sr_lineset; /* opaque, device resident */

/* Upload a line list.  Replaces the per-call Python list of SpectLine objects
 * handed to calc_shapes_lines (spect_classes.py:1378).  Applies the reference's
 * line filter (spect_classes.py:1384-1388 with LinkToMolec 122-150), finds each
 * line's window centre (closest_grid, spect_classes.py:1937-1943) and sorts by
 * centre.  n_kept (may be NULL) receives the number of lines retained.
 * Lines farther than half a window (~3.25 cm^-1 at the default step) from the grid are kept too:
 * their window sits on the first / last grid point, humliv_bb takes one of its outer branches
 * (lineshape.f:272-442) and their far wing inside the grid is added, as in the reference. */
int sr_lineset_create(const sr_lines_desc *lines, const sr_isomolec_desc *iso,
                      const sr_grid_desc *grid, sr_lineset **out,
                      int64_t *n_kept);
int sr_lineset_destroy(sr_lineset *ls);

/* Finite differences in temperature (no reference counterpart: the reference has no temperature Jacobian,
 * spect_main_module.py:300-306 is commented out).  The Humlicek region boundaries of a (line, layer) are index
 * computed (nint, lineshape.f:443-490) and the regions disagree by 1e-5..1e-4 at their seams, so c(T + dT) - c(T)
 * jumps wherever a boundary moves by a point: spikes of (1e-5 y) / dT in a difference quotient.  After this call
 * the coefficient calls on `ls` place il, ir, il2, ir2 and the region-3 interval of every (line, layer) as at the
 * temperatures temps_bounds[n_layers] (HOST, copied) while every width, running x and weight follows the call's
 * own temperatures: c(T + dT) and c(T) then share their region boundaries and the quotient is smooth, also for
 * a one-sided difference with a small dT.  temps_bounds == NULL or n_layers == 0 restores the default. */
int sr_lineset_set_bounds_temps(sr_lineset *ls, const double *temps_bounds, int n_layers);

/* With frozen boundaries (sr_lineset_set_bounds_temps(T_b)): on != 0 makes the next coefficient ops take every line
 * weight (G coefficients / normalisation; NOT the level populations, which the caller passes) at T_b and continue it to
 * the call's temperature by its first-order Taylor term, w(T_b) (1 + (T - T_b) d ln w / d T), while widths, running x and
 * shapes follow the call's T as always.  For temperature derivatives by difference (build's own: the reference has
 * none): (c_lin(T_b + dT) - c(T_b)) / dT is then free of the Boltzmann factors' curvature, the step can grow from
 * 0.002 to 0.05 K and the reference's single-precision staircase (1e-7 |c| / dT) shrinks with it.  0: exact weights. */
int sr_lineset_set_linear_weights(sr_lineset *ls, int on);

/* Layer stack (the Temps / Press lists of make_abscoeff_isomolec,
 * spect_main_module.py:1880, plus level.local_vibtemp, :2065). Host pointers.
 * tvib: [n_levels][n_layers] or NULL for LTE (:2062-2063).  q_part: [n_layers]
 * or NULL to have the library evaluate CalcPartitionSum(mol, iso, T). */
typedef struct {
  int n_layers;
  const double *temps; /* K   */
  const double *press; /* hPa */
  const double *tvib;
  const double *q_part;
} sr_layers_desc;

/* Stream contract of the coarse-grained calls: work is enqueued on `stream` and returns without
 * synchronising.  A lineset owns scratch that consecutive calls share; every call therefore first
 * orders `stream` after the end of the previous sr_abscoeff_layers* / sr_gcoeff_* call on the SAME
 * lineset (an event wait, free when both use one stream), so calls on unrelated streams are safe.
 * Calls on one lineset must still be ISSUED from one host thread at a time.  The sr_set_* mode
 * switches are process-wide atomics; a call reads them once at entry. */

/* abs/emi coefficient spectra for every layer over the grid shard
 * [g_lo, g_hi): what make_abscoeff_isomolec(..., useLUTs=False) returns as
 * AbsSetLOS lists (spect_main_module.py:2128-2131), i.e. calc_shapes_lines +
 * add_PT + the population-weighted combine (:1974-1990, :2036-2080).
 * abs_out / emi_out: DEVICE pointers, [n_layers][g_hi-g_lo] doubles. */
int sr_abscoeff_layers_dev(sr_lineset *ls, const sr_layers_desc *atm,
                           int64_t g_lo, int64_t g_hi, double *abs_out,
                           double *emi_out, void *stream);
/* Same with HOST output buffers (copies back; PCIe-inclusive). */
int sr_abscoeff_layers(sr_lineset *ls, const sr_layers_desc *atm, int64_t g_lo,
                       int64_t g_hi, double *abs_out, double *emi_out);

/* Per-level, per-ctype G-coefficient spectra: what LutSet.add_PT -> SpectralGcoeff.BuildCoeff(lines,
 * Temp, Pres, preCalc_shapes=True) produce for ONE level at every (P, T) of the layer stack
 * (spect_main_module.py:1122-1168, spect_classes.py:1277-1337; called per level from
 * LookUpTable.make, spect_main_module.py:770-774, and make_abscoeff_isomolec, :1983-1990):
 *   sp_emission, ind_emission: sum over the lines whose UPPER level is `level` of G_ctype * shape,
 *   absorption:                sum over the lines whose LOWER level is `level` (spect_classes.py:1304-1313);
 * for an iso-molecule without levels (the 'all' set, spect_classes.py:1316-1319) level must be 0 and
 * every line counts; level = -1 on an iso-molecule WITH levels sums every linked line with the G
 * coefficients of its own levels (the 'all' LutSet of an LTE table, spect_main_module.py:742-748, 774).  No population enters (that is the caller's combine, spect_main_module.py:2073-2080).
 * g_out: DEVICE [3][n_layers][g_hi-g_lo], ctype 0 'sp_emission', 1 'ind_emission', 2 'absorption'.
 * atm->tvib / q_part are not used.  Lines of the level are cut into a sub-lineset on first use. */
int sr_gcoeff_layers_dev(sr_lineset *ls, const sr_layers_desc *atm, int level, int64_t g_lo, int64_t g_hi,
                         double *g_out, void *stream);

/* One level's share of the abs / emi coefficients, the reference's track_levels output
 * (spect_main_module.py:2083-2087): abs = pop_L (Gabs_L - Gind_L), emi = pop_L Gsp_L with
 * pop_L = exp(-c2 E_L / Tvib_L) / Q(T).  Summed over the levels this is sr_abscoeff_layers_dev.
 * abs_out / emi_out: DEVICE [n_layers][g_hi-g_lo]. */
int sr_abscoeff_level_dev(sr_lineset *ls, const sr_layers_desc *atm, int level, int64_t g_lo, int64_t g_hi,
                          double *abs_out, double *emi_out, void *stream);

/* Level-factored route (round 4).  The reference never re-evaluates a line shape when only the vibrational
 * temperatures change: the G spectra of a level depend on (P, T) alone (LutSet.add_PT -> BuildCoeff,
 * spect_main_module.py:1122-1168, spect_classes.py:1277-1337) and every LOS step is the population-weighted sum
 * abs += pop_L (Gabs_L - Gind_L), emi += pop_L Gsp_L over the levels (make_abscoeff_isomolec :2036-2106,
 * make_abscoeff_LUTS_fast :2200-2276).  Only the two spectra pop_L multiplies enter that sum, so the tables hold the
 * PAIR  A_L = Gabs_L - Gind_L,  E_L = Gsp_L  per level and (P, T) row.
 * Round 6: the MULTI-CHANNEL pass -- the near-field kernels walk the full line list ONCE (sr_zones_mc_kernel,
 * sr_wings_mc_kernel: every line's three weighted contributions go to the LDS plane of the level they belong to), the
 * far field stays one far-only pass per level sub-lineset (it is linear per output spectrum).  Before, every line was
 * evaluated twice (in its upper level's pass and in its lower level's) and a sparse level's pass cost twice its lines'
 * share; sr_set_level_route(0) keeps that route (one coefficient op per level), which the exact mode and counting
 * passes always take.  The two routes differ by summation order only.  The three ctypes apart: sr_gcoeff_levels_dev.
 * An iso-molecule without levels has the one pair of its 'all' set (n_levels counts as 1).
 * out: DEVICE [max(n_levels, 1)][2][n_layers][g_hi-g_lo] (level, A | E, row, point).  atm->tvib / q_part are not used.
 * Frozen region boundaries (sr_lineset_set_bounds_temps) apply as in every coefficient op. */
int sr_glevel_pairs_dev(sr_lineset *ls, const sr_layers_desc *atm, int64_t g_lo, int64_t g_hi, double *out, void *stream);
/* LookUpTable.make / LutSet.add_PT for ALL levels at once (spect_main_module.py:718-788, 1122-1168): the G spectra of
 * every level and ctype on the (P, T) rows of atm, by the multi-channel pass (or, where it does not apply, by
 * sr_gcoeff_layers_dev level by level).  g_out: DEVICE [max(n_levels, 1)][3][n_layers][g_hi-g_lo] =
 * level, (sp_emission | ind_emission | absorption), row, point -- per level the layout of sr_gcoeff_layers_dev. */
int sr_gcoeff_levels_dev(sr_lineset *ls, const sr_layers_desc *atm, int64_t g_lo, int64_t g_hi, double *g_out, void *stream);
/* 1 (default): level tables by the multi-channel pass; 0: one coefficient op per level (the A/B partner and fallback). */
int sr_set_level_route(int multi_channel);
/* HIP-event times [ms] of the last multi-channel table build on the handle (its last row batch; sr_set_timing(1)), on the
 * caller's stream: ms4[0] the full list's tables (sr_prep_kernel), [1] sr_zones_mc_kernel, [2] what was left of the far-only
 * passes when it ended, [3] sr_wings_mc_kernel.  With sr_set_overlap(0) the far passes run on the caller's stream too,
 * behind the zones kernel: [1] and [3] are then the two kernels' stand-alone durations.  SR_ERR_ARG if no timed build. */
int sr_last_level_tables_ms(sr_lineset *ls, float *ms4);

/* The combine loop of the level-factored route for n_steps LOS steps at once, each step on one (P, T) row of the
 * pair tables `tab` (as sr_glevel_pairs_dev writes them, n_rows rows):
 *   abs[s] = sum_L pop[s][L] A_L[row[s]],   emi[s] = sum_L pop[s][L] E_L[row[s]]          (smm:2073-2080)
 * in ONE pass over the tables: the steps of a row are taken together, its 2 n_levels spectra are read once.
 * Temperature derivative (optional; the reference has none): with tab_dT = the pair tables at T + dT
 * (region boundaries frozen at T), inv_dT = 1 / dT and dpop[s][L] = d pop_L / dT of the step,
 *   dabs[s] = sum_L dpop[s][L] A_L + pop[s][L] (A'_L - A_L) inv_dT,   demi[s] likewise with E
 * -- the population part analytic, only d G / d T by difference.  tab_dT == NULL: dabs_out / demi_out are not written.
 * step_row [n_steps], pop / dpop [n_steps][n_levels]: HOST.  abs_out / emi_out / dabs_out / demi_out: DEVICE
 * [n_steps][n_pts], rows in the caller's step order (a coefficient row per LOS step for the recursion kernels). */
int sr_glevel_combine_dev(const double *tab, const double *tab_dT, int n_levels, int n_rows, int64_t n_pts, int n_steps,
                          const int32_t *step_row, const double *pop, const double *dpop, double inv_dT,
                          double *abs_out, double *emi_out, double *dabs_out, double *demi_out, void *stream);

/* Look-up-table route: LutSet.calculate (spect_main_module.py:997-1066) for one level and
 * n_steps LOS steps on a table of G spectra resident in HBM, g_tab: DEVICE [3][n_pt][n_pts] (ctype-major,
 * one row per tabulated (P, T) couple, as sr_gcoeff_layers_dev writes them).  Per step, idx4[s] = table rows
 * (P1,T1), (P1,T2), (P2,T1), (P2,T2) and wgt4[s] = (wP1, wP2, wT1, wT2): first linear in P at both
 * temperatures, then linear in T (SpectralGcoeff.interpolate, spect_classes.py:1349-1375); idx4[s][2] < 0:
 * T only between rows idx4[s][0], idx4[s][1] with (wT1, wT2) (below the lowest tabulated pressure, :1007-1025).
 * combine == 0: out_a: DEVICE [3][n_steps][n_pts] receives the interpolated set (out_e unused);
 * combine != 0: the population-weighted combine of make_abscoeff_isomolec (:2073-2080) is applied on the fly,
 * out_a[s] += pop[s] G_abs; out_a[s] -= pop[s] G_ind; out_e[s] += pop[s] G_sp (DEVICE [n_steps][n_pts],
 * zeroed by the caller before the first level).  idx4 / wgt4 / pop: HOST. */
int sr_lut_interp_dev(const double *g_tab, int n_pt, int64_t n_pts, int n_steps, const int32_t *idx4,
                      const double *wgt4, const double *pop, int combine, double *out_a, double *out_e,
                      void *stream);

/* Limb radiance recursion for a batch of rays over the shard (the build's own
 * definition standing in for the absent sbm LineOfSight.radtran_fast, call
 * sites spect_main_module.py:2838, 3214; parity unpinned):
 * ray r crosses segments seg_off[r] .. seg_off[r+1]-1 in photon order; segment
 * s applies layer seg_layer[s] with absorber column seg_col[s] (cm^-2):
 *   tau = abs*col;  I <- I*exp(-tau) + emi*col*(1-exp(-tau))/tau.
 * abs_c/emi_c: DEVICE [n_layers][n_pts]; rad: DEVICE [n_rays][n_pts], read as
 * the initial intensity when init_from_rad != 0, else started from 0.
 * seg_* are HOST arrays. */
int sr_radiance_rays_dev(const double *abs_c, const double *emi_c, int n_layers,
                         int64_t n_pts, int n_rays, const int32_t *seg_off,
                         const int32_t *seg_layer, const double *seg_col,
                         int init_from_rad, double *rad, void *stream);

/* ------------------------------------------------------------------------ *
 * Device LOS pipeline: Curtis-Godson columns per segment on the device, then the      *
 * recursion, for a batch of rays through n_gas gases.  Stands in for the absent sbm                    *
 * LineOfSight.calc_radtran_steps + radtran_fast (call sites spect_main_module.py:2746-2767, 2834-2845, *
 * radtran_3D_ch4.py:297-315); the build's own definition, parity unpinned (columns: curgod_fort_2,     *
 * pinned).                                                                                             *
 * ------------------------------------------------------------------------ */
typedef struct {
  int n_rays, n_gas;        /* n_gas <= 4 */
  const int32_t *seg_off;   /* HOST [n_rays+1]: ray r crosses segments seg_off[r] .. seg_off[r+1]-1 */
  const int32_t *seg_layer; /* HOST [n_seg]: row of abs_c / emi_c (the LOS step's (P, T)) a segment applies */
  const int32_t *pt_off;    /* HOST [n_seg+1]: LOS sample points of segment s: pt_off[s] .. pt_off[s+1]-1 (>= 2) */
  const double *x;          /* HOST [n_pt] path coordinate of the sample points, cm, increasing inside a segment */
  const double *nd;         /* HOST [n_pt] number density, cm^-3 (exponential between sample points, curgods.f) */
  const double *vmr;        /* HOST [n_gas][n_pt] volume mixing ratio (linear between sample points) */
  const double *col_scale;  /* HOST [n_gas] factor on each gas's columns (isotopic abundance), NULL = 1 */
  int los_order;            /* 0 'photon': segments listed along the photon path (spect_main_module.py:2748);
                               1: listed from the observer outwards (walked backwards by the recursion) */
  int solo_absorption;      /* != 0: no emission term, I <- I exp(-tau) (radtran_3D_ch4.py:312) */
  int init_mode;            /* 0: I = 0; 1: rad holds the initial_intensity; 2: Planck spectrum at t_init on the
                               grid w0 + (g_lo + j) step (Calc_BB, spect_classes.py:1881-1892) */
  double t_init, w0, step;
  int64_t g_lo;             /* first grid index of the shard (Planck only) */
} sr_los_desc;

/* Curtis-Godson columns only: col_out HOST [n_gas][n_seg] = col_scale[g] * curgod_fort_2(nd, vmr_g, x) per segment
 * (curgods.f:24-45), evaluated on the device.  Synchronises. */
int sr_los_columns(const sr_los_desc *los, double *col_out);

/* Radiances of the ray batch.  abs_c / emi_c: DEVICE [n_gas][n_layers][n_pts]; rad: DEVICE [n_rays][n_pts].
 * Per-level partial radiances (single_rad[(gas, iso, lev)], spect_main_module.py:2883-2887): pass the level's
 * emission share (sr_abscoeff_level_dev) as emi_c with the total abs_c. */
int sr_limb_rays_dev(const double *abs_c, const double *emi_c, int n_layers, int64_t n_pts, const sr_los_desc *los,
                     double *rad, void *stream);

/* A LOS batch resident on the device: the description staged, the Curtis-Godson columns of its segments integrated
 * (curgod_fort_2) and, where the rays share their shells, the folded sweep's records packed -- ONCE, as the reference
 * computes a line of sight's steps once (los.calc_radtran_steps, spect_main_module.py:2746-2767, 3147) and runs
 * radtran / radtran_fast on them for every call of the forward model (:2838, 3214).  The description's arrays are
 * copied; init_mode / t_init / w0 / step / los_order / solo_absorption are taken as given (g_lo per call).  n_layers:
 * rows of the coefficient tables the handle will be used with (seg_layer is checked against it).  Synchronises. */
typedef struct sr_los sr_los;
int sr_los_create(const sr_los_desc *los, int n_layers, sr_los **out);
int sr_los_destroy(sr_los *h);
/* The same with n_par column (VMR-profile) parameters staged alongside (par_gas, par_w as for sr_limb_rays_jac_dev): the
 * batch of a retrieval, whose paths, densities and parameter masks stay while the VMRs change every iteration.
 * sr_los_set_vmr: new VMRs [n_gas][n_pt] (HOST) at the batch's sample points -- one copy and the column kernel on
 * `stream` (photon-order batches only).  sr_limb_rays_jac_los_dev: radiances and d rad / d x_p through the resident
 * batch, launches only (the folded kernel for <= 8 parameters on shared shells, else the forward sensitivities). */
int sr_los_create_par(const sr_los_desc *los, int n_layers, int n_par, const int32_t *par_gas, const double *par_w,
                      sr_los **out);
int sr_los_set_vmr(sr_los *h, const double *vmr, void *stream);
/* The columns of a resident batch integrated again from its staged sample points (two launches, no copy): for callers
 * whose step includes the column integration by definition. */
int sr_los_refresh_columns(sr_los *h, void *stream);
int sr_limb_rays_jac_los_dev(const double *abs_c, const double *emi_c, int n_layers, int64_t n_pts, sr_los *h, int64_t g_lo,
                             double *rad, double *jac, void *stream);
/* The forward model of ONE retrieval iteration on such a batch in one call -- what spect_main_module.py:2736-2940 does
 * between `add_clim` of the new profiles and `chicalc`: x [n_par] (HOST, the parameter vector in the batch's parameter
 * order) -> VMRs of the retrieved gases at the sample points (sum_p x_p w_p: a profile that is sum_p mask_p x_p on the
 * levels, spect_main_module.py:256-283, interpolated like its masks) -> columns -> radiances and parameter Jacobians ->
 * instrument bands (sr_hires_to_lowres_shard_dev's arguments) -> the closed-form field-of-view integral of every pixel
 * (FOV_integr_1D, :3342-3374; three rays per pixel in the batch's order).
 * fov: [n_rays / 3][7] = delta, delta^3, 2 dmax^2, edge, m2, esse, has_edge per pixel (the geometry factors of the
 * rotated square pixel), or NULL: the rays themselves.  buf: DEVICE scratch [n_rays (1 + n_par)][n_pts].
 * out (HOST): [n_rays / 3 or n_rays][1 + n_par][n_bands], row 0 the radiance, row 1 + p the derivative to x_p.  A
 * spectral shard (g_lo, n_pts) returns its partial band integrals (the integral is linear: all-reduce `out`).
 * Synchronises `stream`. */
int sr_retrieval_forward_dev(const double *abs_c, const double *emi_c, int n_layers, int64_t n_pts, sr_los *h, int64_t g_lo,
                             const double *x, double w0, double step, const double *centers_nm, const double *widths_nm,
                             int n_bands, double n_sigma, int out_units, const double *fov, double *buf, double *out,
                             void *stream);
/* ... and the rest of the iteration in the same call: chi square of the simulated pixels against the observations and the
 * Levenberg-Marquardt step of the optimal-estimation algebra (inversion_algebra, spect_main_module.py:3433-3469) on the
 * band spectra and Jacobians the call has just brought to the host -- K^T S_y^-1 K, S^-1 = G + S_a^-1, S_x, the averaging
 * kernel, dx = (S^-1 + lambda diag S^-1)^-1 (K^T S_y^-1 (y - F) + S_a^-1 (x_a - x)); n_par x n_par LU with partial
 * pivoting, host fp64 (the system is n_par <= 64 wide: nothing for a GPU).  What the caller keeps: the positivity rule
 * of the update (:633-641), the stopping rule (:2963-2973), its bookkeeping.  Single process only (a spectral shard's
 * band integrals are partial: all-reduce first and use the Python algebra).
 *   oe->obs / noise [n_pix n_bands] pixel-major, mask [n_pix n_bands] (1 = used) or NULL, sa_inv [n_par][n_par],
 *   x_apriori [n_par]; pixels = rays / 3 with fov, or the centre ray of every three without (fov == NULL).
 *   out as sr_retrieval_forward_dev's with fov ([n_pix][1 + n_par][n_bands]; without fov the centre rays' rows);
 *   chi_sum = sum over the used elements of ((obs - sim) / noise)^2, n_used their number; dx [n_par], s_x / avk
 *   [n_par][n_par].  SR_ERR_TABLE when the system is singular. */
typedef struct {
  int32_t n_obs;
  const double *obs, *noise;
  const uint8_t *mask;
  const double *sa_inv, *x_apriori;
  double lambda_lm;
} sr_oe_desc;
int sr_retrieval_step_dev(const double *abs_c, const double *emi_c, int n_layers, int64_t n_pts, sr_los *h, int64_t g_lo,
                          const double *x, double w0, double step, const double *centers_nm, const double *widths_nm,
                          int n_bands, double n_sigma, int out_units, const double *fov, double *buf, double *out,
                          const sr_oe_desc *oe, double *chi_sum, int32_t *n_used, double *dx, double *s_x, double *avk,
                          void *stream);
/* ... and the loop around it, for a retrieval whose coefficient spectra stay as they are (only VMRs are retrieved: the
 * loop of inversion_fast_limb, spect_main_module.py:2725-2987): per iteration sr_retrieval_step_dev at the current
 * parameter vector, chi = chi_sum / (n_used - n_dof_par) (:2949), the stopping rule (:2960-2973: relative change below
 * chi_threshold -> *stop = 1 "converged", chi increased -> 2 "raised", else 0 after max_it), then x += dx with the
 * positivity rule (:616-624: a step that would leave a constrained parameter <= 0 is halved until it does not).
 *   chi_hist [max_it], *n_it of them filled; x_hist [max_it + 1][n_par]: the vector before every iteration and after
 *   the last update (*n_it updates when *stop == 0, *n_it - 1 otherwise); out: the LAST iteration's band spectra and
 *   Jacobians; s_x / avk: of the last iteration whose update was applied (as the reference stores them after
 *   inversion_algebra; untouched when there was none).  What the caller keeps is object bookkeeping. */
typedef struct {
  int32_t max_it;
  double chi_threshold;
  const uint8_t *positive; /* [n_par]: 1 = constrain_positive */
  int32_t n_dof_par;       /* parameters in use */
} sr_loop_desc;
int sr_retrieval_loop_dev(const double *abs_c, const double *emi_c, int n_layers, int64_t n_pts, sr_los *h, int64_t g_lo,
                          const double *x0, double w0, double step, const double *centers_nm, const double *widths_nm,
                          int n_bands, double n_sigma, int out_units, const double *fov, double *buf, double *out,
                          const sr_oe_desc *oe, const sr_loop_desc *lp, double *chi_hist, double *x_hist, int32_t *n_it,
                          int32_t *stop, double *s_x, double *avk, void *stream);
/* sr_limb_rays_dev on a resident LOS: kernel launches only (no staging copy, no column kernel, no host plan).
 * g_lo: grid index of abs_c's first point (Planck initial intensity, init_mode 2). */
int sr_limb_rays_los_dev(const double *abs_c, const double *emi_c, int n_layers, int64_t n_pts, sr_los *h, int64_t g_lo,
                         double *rad, void *stream);
/* One forward-model step of one gas in ONE call: sr_abscoeff_layers_dev into abs_out / emi_out, then the recursion of
 * the resident LOS (n_gas = 1) through them into rad [n_rays][g_hi - g_lo] -- what a caller of make_abscoeff_isomolec
 * + radtran_fast does per spectrum (spect_main_module.py:1979-1990, 2838).  A 1/8 spectral shard of BASELINE
 * configs[1] is 0.8 ms of device time: the host's enqueue time per step is what eight ranks must stay under. */
int sr_limb_step_dev(sr_lineset *ls, const sr_layers_desc *atm, int64_t g_lo, int64_t g_hi, double *abs_out,
                     double *emi_out, sr_los *h, double *rad, void *stream);

/* + Jacobian w.r.t. n_par VMR-profile parameters: the VMR of gas par_gas[p] at LOS sample point i is
 * sum_p par_w[p][i] x_p (mask values of RetParam / LinearProfile at the point, spect_main_module.py:319-375), so
 * d col_g[s] / d x_p = col_scale[g] curgod_fort_2(nd, par_w[p], x).  par_gas: HOST [n_par]; par_w: HOST
 * [n_par][n_pt]; jac: DEVICE [n_rays][n_par][n_pts].  Checked against finite differences (unpinned). */
int sr_limb_rays_jac_dev(const double *abs_c, const double *emi_c, int n_layers, int64_t n_pts,
                         const sr_los_desc *los, int n_par, const int32_t *par_gas, const double *par_w, double *rad,
                         double *jac, void *stream);

/* + Jacobian w.r.t. one scalar per layer acting through the coefficients (temperature; BASELINE configs[3]):
 * dabs / demi: DEVICE [n_gas][n_layers][n_pts]; jac: DEVICE [n_rays][n_layers][n_pts].  init_mode 0 or 2. */
int sr_limb_rays_jac_layer_dev(const double *abs_c, const double *emi_c, const double *dabs, const double *demi,
                               int n_layers, int64_t n_pts, const sr_los_desc *los, double *jac, void *stream);

/* Radiances and both kinds of Jacobian in ONE pass over each ray (round 3; BASELINE configs[3]: "Jacobians w.r.t. T
 * and VMR per layer").  Any of the three outputs may be left out: rad NULL; jac_layer NULL (then dabs = demi = NULL);
 * n_par = 0 (then par_gas = par_w = jac_par = NULL) -- but at least one Jacobian.  Each segment's own sensitivity
 * times the transmission of everything behind it is added to the rows it acts on; the host plans per ray which
 * access stores, which adds and which contributions are carried in registers across consecutive segments, so a
 * Jacobian row is written about once per crossing instead of memset + read-add-store per segment.  Falls back to the
 * forward-sensitivity kernels (sr_limb_rays_jac_dev / _jac_layer_dev under sr_set_jac_layer_mode(1)) when a segment
 * touches more than four parameters; same definitions, same layouts.
 * seg_jac_row (HOST [n_seg], or NULL: = seg_layer, n_jac_rows ignored): the row of jac_layer [n_rays][n_jac_rows][n_pts]
 * a segment's per-layer sensitivity is added to.  A 3-D path (spect_main_module.py:2746-2767 with use_tangent_sza =
 * False: vibrational temperatures follow the local SZA along the LOS) gives every LOS step its own coefficient row
 * (seg_layer) while the retrieved scalar still belongs to the step's altitude layer (seg_jac_row). */
int sr_limb_rays_jacobians_dev(const double *abs_c, const double *emi_c, const double *dabs, const double *demi,
                               int n_layers, int64_t n_pts, const sr_los_desc *los, const int32_t *seg_jac_row,
                               int n_jac_rows, int n_par, const int32_t *par_gas, const double *par_w, double *rad,
                               double *jac_layer, double *jac_par, void *stream);

/* Radiances and their Jacobian with respect to n_par retrieval parameters on which the absorber
 * columns depend linearly, col_s = sum_p dcol_dpar[s][p] * x_p (VMR profile parameters of the
 * reference's RetParam / LinearProfile classes, spect_main_module.py:319-375; the reference's own
 * derivative code lives in the absent spect_base_module, call site spect_main_module.py:2874, so this
 * is the build's definition, parity unpinned, checked against finite differences).  Same recursion
 * and layouts as sr_radiance_rays_dev; dcol_dpar: HOST [n_seg][n_par]; rad: DEVICE [n_rays][n_pts];
 * jac: DEVICE [n_rays][n_par][n_pts] = d rad / d x_p. */
int sr_radiance_jac_dev(const double *abs_c, const double *emi_c, int n_layers, int64_t n_pts, int n_rays,
                        const int32_t *seg_off, const int32_t *seg_layer, const double *seg_col,
                        const double *dcol_dpar, int n_par, double *rad, double *jac, void *stream);

/* Radiance Jacobian with respect to one scalar per layer that acts through the layer's own
 * coefficients -- its temperature (BASELINE configs[3]: "Jacobians w.r.t. T ... per layer"; the
 * reference has no temperature Jacobian, spect_main_module.py:300-306 is commented out: build's
 * definition, parity unpinned, checked against finite differences of the whole chain).
 * dabs / demi: DEVICE [n_layers][n_pts] = d(abs, emi of layer k)/d(parameter of layer k), e.g. central
 * differences of two sr_abscoeff_layers_dev calls at T +- dT; jac: DEVICE [n_rays][n_layers][n_pts]. */
int sr_radiance_jac_layer_dev(const double *abs_c, const double *emi_c, const double *dabs, const double *demi,
                              int n_layers, int64_t n_pts, int n_rays, const int32_t *seg_off,
                              const int32_t *seg_layer, const double *seg_col, double *jac, void *stream);

/* Instrument step that follows the path: what
 * SpectralIntensity.hires_to_lowres(lowres_obs, spectral_widths) does
 * (spect_classes.py:1180-1191) for a hi-res spectrum on the cm^-1 grid
 * w0 + j*step in 'ergscm2': conversion of grid and spectrum to nm
 * (spect_classes.py:404-407, 779-783), Gaussian ILS over +-n_sigma sigma by the
 * trapezoid rule on the irregular nm grid (spect_classes.py:883-918, 1926-1934,
 * 1162-1164), conversion to the observation's units (out_units 0 'Wm2',
 * 1 'ergscm2', 2 'nWcm2'; spect_classes.py:1200-1235).
 * rad: DEVICE [n_rays][n_pts] (the whole spectrum); centers_nm / widths_nm:
 * HOST [n_bands] (band centres and Gaussian sigmas, nm); out_host: HOST
 * [n_rays][n_bands].  Synchronises the stream. */
int sr_hires_to_lowres_dev(const double *rad, int n_rays, int64_t n_pts, double w0, double step,
                           const double *centers_nm, const double *widths_nm, int n_bands, double n_sigma,
                           int out_units, double *out_host, void *stream);
/* The same for a spectral shard: rad DEVICE [n_rays][n_pts] holds grid points g_lo .. g_lo + n_pts - 1 of the grid
 * w0 + j step; out_host receives the PARTIAL band integrals over the trapezoids between those points (bands whose
 * window misses the shard: 0).  A multi-GPU retrieval gives every rank its shard plus the next rank's first point
 * and sums the partial integrals over the ranks (one all-reduce of n_rays x n_bands doubles):
 * spect_main_module.py:2814-2818 splits the forward model spectrally in the same way. */
int sr_hires_to_lowres_shard_dev(const double *rad, int n_rays, int64_t n_pts, int64_t g_lo, double w0, double step,
                                 const double *centers_nm, const double *widths_nm, int n_bands, double n_sigma,
                                 int out_units, double *out_host, void *stream);

/* Evaluation mode of the coefficient op.  Far region-1 wings by local Taylor expansions per box of grid
 * points (truncation <= sr_far_field_truncation_bound() of a line's own contribution: 1.6e-11 as built by default, degree
 * 19; 2.6e-13 with -DSR_KFD=22), near field exact, with the expansions built
 * 2: from box pairs -- multipole moments of the lines of a source box (sr_s2m_kernel, sr_m2m_kernel)
 *    translated to every well-separated target box of the level (sr_m2l_kernel), per-line expansions only for
 *    the (line, box) pairs no box pair covers;
 * 1: per line and box at every level (sr_farfield_kernel);
 * 3 (default): 2 -- except for line sets with fewer than 0.35 lines per grid point (the per-level sub-linesets of the
 *    pair tables and look-up tables), which take 1: the box pairs cost S2M / M2M / M2L over every box whatever it
 *    holds;
 * 0: every (line, point) evaluated exactly (sr_abscoeff_wings_kernel + sr_abscoeff_cores_kernel). */
int sr_set_far_field(int on);
/* Far-field mode only: how the kernels of a call share the chip.  1 (default): the decoupled, phased pipeline -- the
 * table preparation, the far-field chain (level-0 pass | S2M -> M2M -> M2L) and the zones kernel run on internal
 * streams on scratch of the call's parity, each as soon as what it reads is ready (the preparation of call c + 1
 * while call c computes), the zones kernel gated behind the level-0 pass and S2M of its own call; only the wings
 * kernel, which writes abs_out / emi_out, is on the caller's stream.  0: the kernels one after the other on the
 * caller's stream (per-kernel times for sr_last_kernel_ms; what the exact mode and counting passes always run).  The
 * caller's stream sees the op complete in order in both modes. */
int sr_set_overlap(int on);
/* Measurement hook of the serial schedule (sr_set_overlap(0)) only: kernel `kernel` of every coefficient op is launched n
 * times back to back -- 0 sr_prep_kernel, 1 level-0 far-field pass, 2 S2M + M2M, 3 M2L, 4 zones, 5 wings; -1: off -- so
 * that ONE kernel can run for seconds beside a power sampler (tools/energy_by_kernel.py).  The call's results are not
 * meaningful while a repeat is set (a repeated M2L adds into its coefficients, the repeated zones kernel stores over the
 * wings kernel's sums). */
int sr_set_kernel_repeat(int kernel, int n);
/* Memory knob: the per-(line, layer) record tables (128 B each, plus the far-field scratch of a layer) of one launch are kept
 * under this many bytes (default 48 GiB of the 288 GB); a longer layer stack (the reference
 * allows imxstp = 8000 LOS steps) is processed in batches of layers. */
int sr_set_table_budget(int64_t bytes);
/* Radiance Jacobians of the device LOS pipeline (sr_limb_rays_jac_dev with more than 8 parameters,
 * sr_limb_rays_jac_layer_dev with more than 8 layers, sr_limb_rays_jacobians_dev).  0 (default): one pass over each
 * ray, every segment's sensitivity times the transmission behind it added to the rows it acts on
 * (sr_limb_adjoint_kernel); where the rays share their coefficient rows (no seg_jac_row) and walk them monotonically
 * inwards and outwards again (limb, slant and nadir paths of a 1-D atmosphere), the FOLDED kernel: the shells are
 * walked once per sweep, a ray's far-side and near-side segment of a shell together, two rays per thread, every
 * Jacobian value stored once (sr_limb_adjoint_fold_kernel; what enters a near-side segment is taken as the observed
 * radiance minus what the segments in front of it emit: values agree with mode 2 to ~1e-15 of the radiance x d tau).
 * 1: the forward-sensitivity kernels always (they carry 16 derivatives through the recursion and repeat it per block
 * of 16) -- kept as the check of the others.  2: one pass per ray in path order, one ray per thread, always.
 * 3: path order, two rays per thread sharing a shell's coefficient loads (sr_limb_adjoint_sync_kernel; the bits of 2). */
int sr_set_jac_layer_mode(int forward);
/* The far field's truncation bound, relative to a line's own contribution at the point: 18 theta^-(degree + 1) of the
 * library as built (theta = 4, degree 19: 1.6e-11; -DSR_KFD=22: 2.6e-13).  The exact mode (sr_set_far_field(0)) has none.
 * What the far-field mode may differ by from the exact mode and from the CPU oracle beyond rounding. */
double sr_far_field_truncation_bound(void);
/* sr_retrieval_forward_dev / _step_dev / _loop_dev with up to 8 parameters on a folded batch: 1 (default) the recursion
 * kernel integrates the instrument bands in its epilogue (partial sums per 64 points; the 1 + n_par spectra per ray
 * are never written: `buf` stays untouched); 0: spectra into `buf`, then sr_hires_to_lowres_shard_dev's kernels -- the
 * same numbers up to the summation order (<= 1e-13 relative), kept as the check and the A/B partner. */
int sr_set_band_fusion(int on);
/* Which recursion kernel the most recent sr_limb_rays_dev call on this thread launched: 1 the path-order kernels
 * (sr_limb_kernel / sr_limb_split_kernel), 2 the folded sweep (sr_limb_fold_fwd_kernel: ray batches that share their
 * shells), 0 none yet.  Diagnostic (tests pin the choice: a 3-D batch with a coefficient row per LOS step must not fold). */
int sr_last_limb_route(void);
/* Tuning knob of the exact wings kernel: grid points per lane (4 or 8; default 8). */
int sr_set_points_per_lane(int p);

/* Executed-work accounting for bench.py's roofline (far-field mode only).  sr_set_counting(1): the
 * following sr_abscoeff_layers* calls run the counting instantiations of the three coefficient kernels
 * (same results; one atomic per wave and counter) -- not for timed runs.  sr_last_eval_counts: the
 * counters of the most recent such call on this lineset, counts10[0] (line, box) far-field expansions,
 * [1] region-1 evaluations done point by point, [2] window-end expansions, [3] (point, level)
 * far-field polynomial evaluations, [4] region-2, [5] region-3, [6] region-4 evaluations, [7] (line, side)
 * multipole expansions and [8] (source box, target box, layer) translations of the box-pair far field
 * (sr_set_far_field(2)), [9] 0.  Synchronises. */
int sr_set_counting(int on);
int sr_last_eval_counts(sr_lineset *ls, uint64_t *counts10);

/* Timing hook for bench.py: HIP-event times (ms) of the kernels of the most
 * recent sr_abscoeff_layers* call on this lineset, measured on the stream they
 * were launched on.  ms5[0] sr_prep_kernel.  Far-field mode with overlap (the
 * default): ms5[1] = the whole coefficient op (far field + near wings + near
 * zones, which run partly side by side), rest 0.  Far-field mode without overlap:
 * ms5[1] sr_farfield_kernel, ms5[2] sr_abscoeff_near_wings_kernel, ms5[3]
 * sr_abscoeff_near_zones_kernel.  Exact mode: ms5[1] sr_abscoeff_wings_kernel,
 * ms5[2] sr_abscoeff_cores_kernel.  Unused entries 0.  Synchronises. */
int sr_last_kernel_ms(sr_lineset *ls, float *ms5);
/* 0: the coefficient op records no timing events (seven hipEventRecord fewer per call: a 1/8 spectral shard's step is
 * bound by the host's enqueue time before anything else); sr_last_kernel_ms then returns SR_ERR_ARG.  1 (default). */
int sr_set_timing(int on);
/* sr_set_timing(2): additionally two events around the recursion of sr_retrieval_forward_dev / _step_dev / _loop_dev
 * (record packing + the one-sweep kernel with the bands in its epilogue); sr_los_last_kernel_ms: their HIP-event time for
 * the most recent such call on the handle (SR_ERR_ARG when it was not timed).  Synchronises.  bench.py --config 4. */
int sr_los_last_kernel_ms(sr_los *h, float *ms);

#ifdef __cplusplus
}
#endif
#endif
