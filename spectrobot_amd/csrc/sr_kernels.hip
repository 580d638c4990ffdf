// sr_kernels.hip -- HIP kernels of the spectral hot path for gfx950 (MI355X).
//
//   sr_prep_kernel      one thread per (line, layer): widths, G coefficients,
//                       level-population weights, Humlicek region boundaries
//                       -> FastRec / ColdRec tables in HBM.
//   coefficient spectra, gather formulation (no global atomics, no [n_lines x 13010]
//   matrix, coalesced fp64 stores; fp64 VALU bound):
//     default  far region-1 wings as per-box Taylor sums (local expansions), built from box pairs:
//                sr_s2m_kernel / sr_m2m_kernel   multipole moments of the lines of each source box, all levels
//                sr_m2l_kernel                   moments -> local coefficients (fp64 MFMA: the one matrix product here)
//                sr_farfield_kernel<., true>     per-line expansions of the (line, box) pairs no box pair covers
//              (sr_set_far_field(1): sr_farfield_kernel<., false>, one expansion per (line, box) at every level)
//              sr_abscoeff_near_wings_kernel  exact region 1 of the near lines + one polynomial per level and point
//              sr_abscoeff_near_zones_kernel  regions 2/3/4 through an LDS image of the group
//     exact    sr_abscoeff_wings_kernel / sr_abscoeff_cores_kernel: every evaluation
//   sr_los_columns_kernel / sr_limb_kernel / sr_limb_jac*_kernel   Curtis-Godson columns, limb recursion, Jacobians
//   sr_radiance_kernel / sr_radiance_jac_kernel   the same recursion on host-built columns
//   sr_lowres_kernel    Gaussian ILS onto low-resolution bands (hires_to_lowres)
//   shims               humliv_bb / sum_all_lines / curgod_fort_N call shapes.
#include <climits>
#include <cstdlib>

#include "sr_device.hpp"
#include "sr_kernels.hpp"

namespace sr {

// ------------------------------------------------------------------------
// prep: spect_classes.py:174-206 (MakeShapeLine), 312-343 (Calc_Gcoeffs),
//       spect_main_module.py:2049-2080 (population weights), lineshape.f:443-490
// ------------------------------------------------------------------------
// Widths, normalisation and the two output weights of line ln in layer k.
//   lw, dw' (= dw / sqrt(ln2)), fac        spect_classes.py:1972, 1984, 1997-1999
//   G coefficients                         spect_classes.py:326-337, 1806-1853
//   weights by W.mode (sr_kernels.hpp: WeightMode), already divided by fac (shape = y / fac, :2003)
struct LinePhys {
  double lw, dwp, wabs, wemi;
  double w3; // kWeightChannels only: the third weight (0 otherwise)
};
__device__ inline LinePhys line_physics(const LinesDev &L, const LayersDev &A, const WeightMode W, int ln, int k) {
  LinePhys P;
  const double T = A.temps[k];
  const double x0 = L.freq[ln];
  // exp_bounded: the library routine without its special cases (|argument| clamped to 700: e^-700 = 1e-304 is as good
  // as the 0 the reference's exp underflows to much later); three of them were 12 % of this kernel's instructions
  auto ex = [](double u) { return exp_bounded(fmin(fmax(u, -700.0), 700.0)); };
  P.lw = ex(L.t_dep[ln] * A.ltrat[k]) * (L.air_broad[ln] * A.p_atm[k]); // (296/T)^n gamma P  (spcl:1972)
  const double dw = x0 / kCcgs * A.sqk[k];
  P.dwp = dw / A.sqrt_ln2;
  // Linearised weights (sr_lineset_set_linear_weights, with frozen boundaries): the G coefficients and the
  // normalisation are taken at the BOUNDARY temperature Tb and continued to the call's T by their first-order Taylor
  // term, w(Tb) (1 + (T - Tb) d ln w / d T): a difference quotient (c_lin(Tb + dT) - c(Tb)) / dT then carries no
  // curvature of the Boltzmann factors (c2 eps / T^2 ~ 0.1 / K: the truncation that held its step to 0.002 K, where the
  // reference's single-precision staircase is 1e-7 |c| / dT) and the step can be 25 times larger.  Widths and
  // running x above follow the call's own T as always.
  const bool lin = A.frozen && A.linear_w;
  const double Tw = lin ? A.temps_b[k] : T;
  const double fac = (lin ? x0 / kCcgs * A.sqk_b[k] : dw) * A.sqrt_pi_ln2;
  const double dTl = T - Tw;
  double g_sp = 0., g_in = 0., g_ab = 0.;
  const double a_co = L.a_coeff[ln], gu = L.g_up[ln], gl = L.g_lo[ln];
  if (a_co != 0.0 && gl != 0.0 && gu != 0.0) {
    const double four_pi = 4 * kPi;
    const double el = L.e_lower[ln];
    const double eps_up = el + x0 - L.evib_up[ln], eps_lo = el - L.evib_lo[ln];
    double rot_up = gu * ex(-kC2 * eps_up / Tw);
    double rot_lo = gl * ex(-kC2 * eps_lo / Tw);
    if (lin) { // d ln(exp(-c2 eps / T) / sqrt(T)) / d T = c2 eps / T^2 - 1 / (2 T)   (fac ~ dw ~ sqrt(T))
      rot_up *= fma(dTl, fma(kC2 * eps_up, 1.0 / (Tw * Tw), -0.5 / Tw), 1.0);
      rot_lo *= fma(dTl, fma(kC2 * eps_lo, 1.0 / (Tw * Tw), -0.5 / Tw), 1.0);
    }
    const double hcf = L.hcf[ln];
    g_sp = hcf * rot_up * a_co / four_pi;
    g_in = hcf * rot_up * L.b21[ln] / four_pi;
    g_ab = hcf * rot_lo * L.b12[ln] / four_pi;
  }
  const int lu = L.lev_up[ln], ll = L.lev_lo[ln];
  const double *pop = A.pop + (size_t)k * A.n_pop;
  double wabs, wemi;
  P.w3 = 0.0;
  if (W.mode == kWeightFolded) { // spect_main_module.py:2073-2080 folded per line
    const double pu = pop[lu], pl = pop[ll];
    wabs = pl * g_ab - pu * g_in;
    wemi = pu * g_sp;
  } else if (W.mode == kWeightGabsGsp) { // BuildCoeff of level W.level: absorption | sp_emission (spcl:1304-1313)
    wabs = (W.level < 0 || ll == W.level) ? g_ab : 0.0; // level < 0: every line (the 'all' LutSet of an LTE table)
    wemi = (W.level < 0 || lu == W.level) ? g_sp : 0.0;
  } else if (W.mode == kWeightGind) {    // ... ind_emission | nothing
    wabs = (W.level < 0 || lu == W.level) ? g_in : 0.0;
    wemi = 0.0;
  } else if (W.mode == kWeightChannels) { // the multi-channel pass: the line's three G coefficients, each to its own channel
    wabs = g_ab;                           // -> the absorption spectrum of the line's LOWER level
    wemi = g_sp;                           // -> the sp_emission spectrum of its UPPER level
    P.w3 = (W.level == 0 ? -g_in : g_in) / fac; // -> ind_emission of its UPPER level (level 0: subtracted from that level's absorption spectrum, smm:2078)
  } else if (W.mode == kWeightLevelPair) { // what pop_level multiplies in the combine loop (smm:2078-2080)
    const bool all = W.level < 0;          // no level table: the 'all' set (smm:2052-2057)
    wabs = ((all || ll == W.level) ? g_ab : 0.0) - ((all || lu == W.level) ? g_in : 0.0);
    wemi = (all || lu == W.level) ? g_sp : 0.0;
  } else {                               // kWeightTracked: one level's share of abs / emi (smm:2083-2087)
    const double pu = pop[lu], pl = pop[ll];
    wabs = (ll == W.level ? pl * g_ab : 0.0) - (lu == W.level ? pu * g_in : 0.0);
    wemi = lu == W.level ? pu * g_sp : 0.0;
  }
  P.wabs = wabs / fac;
  P.wemi = wemi / fac;
  return P;
}

constexpr int kPrepBlock = 64; // one wave per block: a four-wave block with 20 KB of LDS never found room beside the zones
                               // kernel (16 one-wave blocks of 10 KB fill a CU's LDS; a retiring wave frees ONE slot)
                               // or the wings kernel, and the next call's preparation ran after them instead of beside
// (the body apart from its kernel: sr_prep_batch_kernel runs it for every sparse far-only pass of a multi-channel
// table build in one launch; bx, k: the block's line chunk and layer)
__device__ __forceinline__ void prep_body(const LinesDev &L, const LayersDev &A, const GridParams &gp, const WeightMode W,
                                          int line_lo, int n_sub, int cold_lo, int cold_hi, FastRec *__restrict__ fast,
                                          ColdRec *__restrict__ cold, const int bx, const int k) {
  // records leave through LDS: stored from the registers, a lane's 80 / 128 B record goes out in 16-byte pieces at
  // a 80 / 128 B stride across the lanes; the wave's 64 records are contiguous in the table, so they are transposed
  // and stored 1 KB of consecutive bytes per instruction instead (the kernel writes its 1.63 GB at 5.2 TB/s).
  __shared__ uint4 s_rec[kPrepBlock / 64][64 * sizeof(FastRec) / 16]; // per wave: 64 fast records, then its 64 cold records
  static_assert(sizeof(ColdRec) <= sizeof(FastRec), "the cold records share the fast records' staging buffer");
  const int i0 = bx * blockDim.x + threadIdx.x;
  // (k = blockIdx.y; layers on the fast grid axis instead -- the line arrays then stay in L2 -- measured slower: 0.32 vs 0.29 ms, the record rows of consecutive blocks lie 8 MB apart)
  const int wave_first = i0 - (threadIdx.x & 63);
  if (wave_first >= n_sub) return; // whole wave out of range
  const bool valid = i0 < n_sub;
  const int i = valid ? i0 : n_sub - 1;
  const int ln = line_lo + i;
  const double x0 = L.freq[ln];
  const LinePhys ph = line_physics(L, A, W, ln, k);
  const double lw = ph.lw, dwp = ph.dwp;

  const int ic = L.ic[ln];
  WinX xf{gp.lin_start, gp.lin_delta, grid_at(gp, ic)};
  Bounds B = humliv_bounds(xf, kImxsig, x0, lw, dwp), B0 = B;
  double dwp0 = dwp;
  if (A.frozen) {
    // sr_lineset_set_bounds_temps: the region boundaries (indices) as at the layer's boundary temperature, every
    // width and running x at the call's own -- c(T + dT) and c(T) then share their seams
    const double lw_b = exp_bounded(fmin(fmax(L.t_dep[ln] * A.ltrat_b[k], -700.0), 700.0)) * (L.air_broad[ln] * A.p_atm[k]);
    dwp0 = x0 / kCcgs * A.sqk_b[k] / A.sqrt_ln2;
    B0 = humliv_bounds(xf, kImxsig, x0, lw_b, dwp0);
    B.il = B0.il; B.ir = B0.ir; B.il2 = B0.il2; B.ir2 = B0.ir2;
    B.xr = div_with(xf(B.ir) - x0, dwp, B.inv_dwp); // lineshape.f:471 at the frozen ir
  }

  FastRec r;
  r.xl = B.xl;
  r.xr = B.xr;
  r.xstep = B.xstep;
  r1_set(r, B.ry);
#if SR_FASTREC64
  r.w3 = ph.w3;
#endif
  r.wabs = ph.wabs;
  r.wemi = ph.wemi;
  r.j1 = ic - kHalf;
  r.ilir = (uint32_t)B.il | ((uint32_t)B.ir << 16);
  {
    const int lane = threadIdx.x & 63, n_valid = min(64, n_sub - wave_first);
    uint4 *buf = s_rec[threadIdx.x >> 6];
    const size_t o = (size_t)k * n_sub + wave_first;
    constexpr int NF = sizeof(FastRec) / 16, NC = sizeof(ColdRec) / 16;
    const uint4 *rp = reinterpret_cast<const uint4 *>(&r);
#pragma unroll
    for (int q = 0; q < NF; ++q) buf[lane * NF + q] = rp[q];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint4 *gf = reinterpret_cast<uint4 *>(fast + o);
#pragma unroll
    for (int q = 0; q < NF; ++q)
      if (q * 64 + lane < n_valid * NF) gf[q * 64 + lane] = buf[q * 64 + lane];
    __builtin_amdgcn_wave_barrier();
    // The cold record (regions 2-4) is read only for lines whose zone meets [cold_lo, cold_hi]: in
    // far-field mode that is the shard, so the halo lines of a multi-GPU shard (half of its lines
    // at 8 GPUs) skip its computation and its 48 B; exact mode reads it for window ends too and
    // passes the whole index range.
    const int zl = r.j1 + B.il - 1, zh = r.j1 + B.ir - 1;
    if (__any(valid && zl <= cold_hi && zh >= cold_lo)) {
      ColdRec c = make_cold(B0, dwp0, x0, xf); // il2 | ir2 and the region-3 interval (at the boundary temperature)
      if (A.frozen) {
        c.ry = B.ry;
        c.dwp = dwp;
        c.xs2l = div_with(x0 - xf(B.il), dwp, B.inv_dwp);  // lineshape.f:504
        c.xs2r = div_with(xf(B.ir2) - x0, dwp, B.inv_dwp); // :514
      }
      const uint4 *cp = reinterpret_cast<const uint4 *>(&c);
      uint4 *gc = reinterpret_cast<uint4 *>(cold + o);
      // staged like the fast records: 64 x 48 bytes, consecutive in the table, 1 KB per store instruction
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < NC; ++q) buf[lane * NC + q] = cp[q];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < NC; ++q)
        if (q * 64 + lane < n_valid * NC) gc[q * 64 + lane] = buf[q * 64 + lane];
    }
  }
  // (The widest zone of the layer, which the other kernels use to bound their candidate ranges, comes from the host
  // as a bound -- (lw_max + 15 dw'_max) / step: the wave maxima + one atomicMax per wave on the layer's single counter
  // serialised in L2 and WERE this kernel's 0.8 ms: it took as long with its stores or its arithmetic compiled out.)
}
__global__ __launch_bounds__(kPrepBlock) void sr_prep_kernel(LinesDev L, LayersDev A, GridParams gp, WeightMode W,
                                                      int line_lo, int n_sub, int cold_lo, int cold_hi,
                                                      FastRec *__restrict__ fast,
                                                      ColdRec *__restrict__ cold) {
  prep_body(L, A, gp, W, line_lo, n_sub, cold_lo, cold_hi, fast, cold, (int)blockIdx.x, (int)blockIdx.y);
}
// The tables of ALL sparse far-only passes of a table build (FarBatchItem) in one launch: grid (line chunks of the
// largest item, layers, items); fast records only (an empty cold range).
__global__ __launch_bounds__(kPrepBlock) void sr_prep_batch_kernel(const FarBatchItem *__restrict__ items, LayersDev A, GridParams gp) {
  const FarBatchItem it = items[blockIdx.z];
  prep_body(it.L, A, gp, it.W, it.line_lo, it.n_sub, INT_MAX / 2, INT_MIN / 2, it.fast, nullptr, (int)blockIdx.x, (int)blockIdx.y);
}

// ------------------------------------------------------------------------
// Lines whose centre lies outside their own 13010-point window (farther than ~3.25 cm-1 from the
// grid): closest_grid puts their window on the first / last grid point (spect_classes.py:1941) and
// humliv_bb takes one of its two sequential outer branches; the part of the window inside the grid
// is added to the spectrum like any other line (spect_classes.py:1113-1120).  They are few: one
// thread per (line, layer) walks the Fortran's control flow for the segment bounds, then one thread
// per (grid point, layer) adds the lines in list order (gather: deterministic, no atomics).
// ------------------------------------------------------------------------
__global__ __launch_bounds__(64) void sr_outer_prep_kernel(LinesDev L, int n_out, LayersDev A, GridParams gp,
                                                           WeightMode W, OuterRec *__restrict__ recs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
  if (i >= n_out) return;
  const LinePhys ph = line_physics(L, A, W, i, k);
  const double x0 = L.freq[i], dw = ph.dwp;
  const int ic = L.ic[i], n = kImxsig;
  const WinX xf{gp.lin_start, gp.lin_delta, grid_at(gp, ic)};
  OuterRec r;
  r.ry = ph.lw / dw;                        // lineshape.f:261
  r.ryf = (double)(float)r.ry;
  r.xstep = (xf(2) - xf(1)) / dw;           // :265-266
  region1_coef(r.ry, r.a, r.b, r.c, r.d);
  region2_coef(r.ry, r.q2);
  r.wabs = ph.wabs;
  r.wemi = ph.wemi;
  r.j1 = ic - kHalf;
  r.c_lo = r.r2_lo = r.r1_lo = 1;
  r.c_hi = r.r2_hi = r.r1_hi = 0;
  r.c_ref = 1;
  r.x_core = r.x_r2 = r.x_r1 = 0.0;
  const double ry = r.ry, xstep = r.xstep;
  if (x0 <= xf(1)) {                        // :272-357
    r.dir = 1;
    int j = 1;
    double rx = (xf(j) - x0) / dw;
    r.x_core = rx;
    while ((rx + ry < 5.5) && (j <= n)) { j = j + 1; rx = rx + xstep; }
    r.c_lo = 1; r.c_hi = j - 1; r.c_ref = 1;
    if (j <= n) {
      int l = max((int)round((15.0 - ry - rx) / xstep), 0) + j;
      l = min(l, n);
      if (l > j) {
        r.r2_lo = j; r.r2_hi = l;
        r.x_r2 = (xf(j) - x0) / dw;
        l = l + 1;
      }
      if (l < j) l = j;
      if (l < n) {
        r.r1_lo = l; r.r1_hi = n;
        r.x_r1 = (xf(l) - x0) / dw;
      }
    }
  } else {                                  // x0 >= x(i2), :358-442
    r.dir = -1;
    int j = n;
    double rx = (x0 - xf(j)) / dw;
    r.x_core = rx;
    while ((rx + ry < 5.5) && (j >= 1)) { j = j - 1; rx = rx + xstep; }
    r.c_lo = j + 1; r.c_hi = n; r.c_ref = n;
    if (j >= 1) {
      int l = j - max((int)round((15.0 - ry - rx) / dw / xstep), 0); // sic, :404
      l = max(l, 1);
      if (l == n) l = n + 1;
      if (l < j) {
        r.r2_lo = l; r.r2_hi = j;
        r.x_r2 = (x0 - xf(l)) / dw;
      }
      if (l >= 1) {
        r.r1_lo = 1; r.r1_hi = l - 1;
        r.x_r1 = (x0 - xf(1)) / dw;
      }
    }
  }
  recs[(size_t)k * n_out + i] = r;
}

__global__ __launch_bounds__(256) void sr_outer_add_kernel(const OuterRec *__restrict__ recs, int n_out, int p_lo,
                                                           int p_hi, int g_lo, int g_hi,
                                                           double *__restrict__ abs_out,
                                                           double *__restrict__ emi_out) {
  const int j = p_lo + blockIdx.x * blockDim.x + threadIdx.x, layer = blockIdx.y;
  if (j >= p_hi) return;
  const OuterRec *row = recs + (size_t)layer * n_out;
  double acc_a = 0., acc_e = 0.;
  for (int i = 0; i < n_out; ++i) {
    const OuterRec &r = row[i]; // wave-uniform address
    const int k = j - r.j1 + 1;
    if (k < 1 || k > kImxsig) continue;
    const double s = (double)r.dir;
    double y;
    // the Fortran writes core, region 2, region 1 in this order over disjoint index ranges
    if (k >= r.c_lo && k <= r.c_hi) {
      y = core_point(fma(s * (double)(k - r.c_ref), r.xstep, r.x_core), r.ry, r.ryf);
    } else if (k >= r.r2_lo && k <= r.r2_hi) {
      y = region2_val(r.q2, fma(s * (double)(k - r.r2_lo), r.xstep, r.x_r2));
    } else if (k >= r.r1_lo && k <= r.r1_hi) {
      const double x = fma(s * (double)(k - r.r1_lo), r.xstep, r.x_r1);
      const double x2 = x * x;
      y = fma(x2, r.b, r.a) * fast_rcp<2>(fma(x2, fma(x2, 4.0, r.d), r.c));
    } else {
      continue;
    }
    acc_a = fma(r.wabs, y, acc_a);
    acc_e = fma(r.wemi, y, acc_e);
  }
  const size_t o = (size_t)layer * (size_t)(g_hi - g_lo) + (size_t)(j - g_lo);
  abs_out[o] += acc_a;
  emi_out[o] += acc_e;
}

int launch_outer(const LinesDev &Lo, int n_out, const LayersDev &A, const GridParams &gp, const WeightMode &W,
                 OuterRec *recs, int g_lo, int g_hi, double *abs_out, double *emi_out, hipStream_t st) {
  if (n_out <= 0 || A.n_layers <= 0 || g_hi <= g_lo) return 0;
  // their windows sit on the first / last grid point: only points within half a window of a grid end
  const int ranges[2][2] = {{max(g_lo, 0), min(g_hi, kHalf)}, {max(g_lo, max(gp.n_grid - 1 - kHalf, kHalf)), g_hi}};
  if (ranges[0][0] >= ranges[0][1] && ranges[1][0] >= ranges[1][1]) return 0;
  hipLaunchKernelGGL(sr_outer_prep_kernel, dim3((n_out + 63) / 64, A.n_layers), dim3(64), 0, st, Lo, n_out, A, gp, W,
                     recs);
  for (int q = 0; q < 2; ++q) {
    const int lo = ranges[q][0], hi = ranges[q][1];
    if (lo >= hi) continue;
    hipLaunchKernelGGL(sr_outer_add_kernel, dim3((hi - lo + 255) / 256, A.n_layers), dim3(256), 0, st, recs, n_out,
                       lo, hi, g_lo, g_hi, abs_out, emi_out);
  }
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------
// coefficient kernels (gather formulation)
//
// One wavefront owns WP = 64*P consecutive grid points of one layer ("wave
// tile") and walks the nu-sorted lines whose 13010-point windows touch it.
// Everything per line is wave-uniform and lives in SGPRs (records are read with
// scalar loads); every lane accumulates abs/emi for its own P points (stride
// 64, so each slot p is 64 consecutive points and stores are coalesced).
//
//   sr_abscoeff_wings_kernel  (line, group of 256 points) pairs for which the
//       WHOLE group lies in one region-1 wing and inside the window (98 % of all
//       evaluations): x = x_b + lane_p*xstep, (a + x^2 b)/(c + x^2 (d + 4 x^2)),
//       one reciprocal shared by the group's four slots, no branches in the body.
//   sr_abscoeff_cores_kernel  the remaining (line, group) pairs -- group meets
//       the line's region-2/3/4 zone or a window end -- evaluated slot by slot
//       with the general, region-by-index code; adds into the wings' output.
// Both apply the same classify() to the same groups, so every (line, point) is
// counted exactly once.
// ------------------------------------------------------------------------
// first line (index into the shard's record table) whose window centre is >= v
__device__ inline int lower_bound_ic(const IcIndex &ix, int v) {
  const int g = ix.first[min(max(v - ix.x0, 0), ix.n_tab - 1)];
  return min(max(g - ix.line_lo, 0), ix.n_sub);
}

// XCD-aware bijective remap (8 XCDs, blocks dealt round-robin): blocks that
// share an XCD get consecutive work ids, so neighbouring tiles of one layer --
// which read almost the same records -- hit the same L2.
__device__ inline int xcd_remap(int b, int nb) {
  const int q = nb >> 3, r = nb & 7, x = b & 7, i = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// ... for kernels whose work ids differ in COST along the id axis (the per-line far-field kernels: widest level first,
// a top-level box holds ten times the candidates of a level-0 box): xcd_remap hands every XCD one contiguous eighth of
// the ids -- the first XCD all the wide boxes -- and the others wait for it (CUs idle 29 % of the sparse passes' kernel:
// SQ_BUSY_CU_CYCLES).  Here the ids are dealt in groups of G consecutive ones, round-robin over the XCDs: neighbours
// still share an L2, every XCD gets every level.  Bijective on [0, nb).
__device__ inline int xcd_remap_groups(int b, int nb, int G) {
  const int full = nb / (8 * G) * (8 * G);
  if (b >= full) return b;
  const int x = b & 7, i = b >> 3;
  return ((i / G) * 8 + x) * G + i % G;
}

// Executed-work counters (counting instantiations only): sum a per-lane count over the wave, one
// atomic per wave and counter.
__device__ inline void count_add(unsigned long long *cnt, int which, unsigned per_lane, int lane) {
  unsigned v = per_lane;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  if (lane == 0 && v) atomicAdd(&cnt[which], (unsigned long long)v);
}

// 1: every point of [lo, hi] is strictly left of il (region 1, running x from
// k = 1) and inside the window; 2: strictly right of ir; 0: anything else.
__device__ inline int classify(int j1, int il, int ir, int lo, int hi) {
  if (lo >= j1 && hi < j1 + il - 1) return 1;
  if (lo > j1 + ir - 1 && hi <= j1 + (kImxsig - 1)) return 2;
  return 0;
}
// -x (left) or x (right) at grid index `at`: only x^2 is used.
__device__ inline double wing_x_at(const FastRec &r, int cls, int j1, int at) {
  return cls == 1 ? fma((double)(at - j1), r.xstep, -r.xl)
                  : fma((double)(at - (j1 + r.ir() - 1)), r.xstep, r.xr);
}

// Region 1 at four points per lane (x = xb + fl[i]*xs) with ONE reciprocal:
// 1/(d0 d1 d2 d3) by v_rcp_f64 (2^-24) + one Newton step (2e-15), unfolded by
// multiplications.  den >= 4*15^4 and <= ~1e19 here, so the product of four
// cannot leave the fp64 range.
__device__ inline void wing_eval4(const double xb, const double xs, const double a, const double b,
                                  const double c, const double d, const double wa, const double we,
                                  const double *fl, double *acc_a, double *acc_e) {
  double num[4], den[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double x = fma(fl[i], xs, xb);
    const double x2 = x * x;
    num[i] = fma(x2, b, a);
    den[i] = fma(x2, fma(x2, 4.0, d), c);
  }
  const double d01 = den[0] * den[1], d23 = den[2] * den[3];
  const double r = fast_rcp<1>(d01 * d23);
  const double r01 = r * d23, r23 = r * d01;
  const double q0 = (num[0] * den[1]) * r01, q1 = (num[1] * den[0]) * r01;
  const double q2 = (num[2] * den[3]) * r23, q3 = (num[3] * den[2]) * r23;
  acc_a[0] = fma(wa, q0, acc_a[0]); acc_e[0] = fma(we, q0, acc_e[0]);
  acc_a[1] = fma(wa, q1, acc_a[1]); acc_e[1] = fma(we, q1, acc_e[1]);
  acc_a[2] = fma(wa, q2, acc_a[2]); acc_e[2] = fma(we, q2, acc_e[2]);
  acc_a[3] = fma(wa, q3, acc_a[3]); acc_e[3] = fma(we, q3, acc_e[3]);
}

constexpr int kGroup = 256; // points per ownership group: 4 slots of 64 (one shared reciprocal)

template <int P>
__global__ __launch_bounds__(64) void sr_abscoeff_wings_kernel(
    const FastRec *__restrict__ fast,
    IcIndex ix, // index into the sorted window centres of the prepped lines
    int n_sub, int n_tiles, int g_lo, int g_hi, double *__restrict__ abs_out,
    double *__restrict__ emi_out) {
  static_assert(P % 4 == 0, "P must be a multiple of 4");
  constexpr int WP = 64 * P, NG = P / 4;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int layer = wg / n_tiles, tile = wg - layer * n_tiles;
  const int wlo = g_lo + tile * WP;
  const int whi = min(wlo + WP, g_hi) - 1;
  const int lane = threadIdx.x;

  // lines whose window [ic-6505, ic+6504] meets [wlo, whi]
  const int l0 = lower_bound_ic(ix, wlo - (kHalf - 1));
  const int l1 = lower_bound_ic(ix, whi + kHalf + 1);

  double acc_a[P], acc_e[P], fl[4];
#pragma unroll
  for (int p = 0; p < P; ++p) acc_a[p] = acc_e[p] = 0.;
#pragma unroll
  for (int i = 0; i < 4; ++i) fl[i] = (double)(lane + 64 * i);
  const FastRec *frow = fast + (size_t)layer * n_sub;
  FastRec nxt = frow[l0 < l1 ? l0 : 0];
  for (int l = l0; l < l1; ++l) {
    // wave-uniform address: two scalar loads; the next record is fetched while this
    // one is evaluated (the table has one record of slack behind its end)
    const FastRec r = nxt;
    nxt = frow[l + 1];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int glo = wlo + kGroup * g, ghi = min(glo + kGroup - 1, whi);
      const int cls = glo <= whi ? classify(r.j1, r.il(), r.ir(), glo, ghi) : 0;
      if (cls == 0) continue; // left to sr_abscoeff_cores_kernel (or outside the window)
      const double xb = wing_x_at(r, cls, r.j1, glo);
      const R1Coef rq = r1_of(r);
      wing_eval4(xb, r.xstep, rq.a, rq.b, rq.c, rq.d, r.wabs, r.wemi, fl, acc_a + 4 * g, acc_e + 4 * g);
    }
  }
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

// One wave per group of 256 points: the (line, group) pairs the wings kernel skips.
__global__ __launch_bounds__(64) void sr_abscoeff_cores_kernel(
    const FastRec *__restrict__ fast, const ColdRec *__restrict__ cold,
    IcIndex ix, const int *__restrict__ zmax, // [n_layers] max zone half-width
    int n_sub, int n_groups, int g_lo, int g_hi, GridParams gp, double *__restrict__ abs_out,
    double *__restrict__ emi_out) {
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int layer = wg / n_groups, grp = wg - layer * n_groups;
  const int wlo = g_lo + grp * kGroup;
  const int whi = min(wlo + kGroup, g_hi) - 1;
  const int lane = threadIdx.x;
  const int zm = min(zmax[layer], kHalf - 1);

  // candidates (as ranges of the sorted centre list), C <= A <= B by start:
  //  C: window END inside the group      ic+6504 in [wlo, whi)
  //  A: region-2/3/4 zone may meet it    ic in [wlo-zm, whi+zm]
  //  B: window START inside the group    ic-6505 in (wlo, whi]
  const int c0 = lower_bound_ic(ix, wlo - (kHalf - 1));
  const int c1 = lower_bound_ic(ix, whi - (kHalf - 1));
  const int a0 = lower_bound_ic(ix, wlo - zm);
  const int a1 = lower_bound_ic(ix, whi + zm + 1);
  const int b0 = lower_bound_ic(ix, wlo + kHalf + 1);
  const int b1 = lower_bound_ic(ix, whi + kHalf + 1);
  int rs[3], re[3];
  rs[0] = c0; re[0] = max(c1, c0);
  rs[1] = max(a0, re[0]); re[1] = max(a1, rs[1]);
  rs[2] = max(b0, re[1]); re[2] = max(b1, rs[2]);

  double acc_a[4], acc_e[4], fl[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    acc_a[p] = 0.;
    acc_e[p] = 0.;
    fl[p] = (double)(lane + 64 * p);
  }
  const FastRec *frow = fast + (size_t)layer * n_sub;
  const ColdRec *crow = cold + (size_t)layer * n_sub;
  for (int rg = 0; rg < 3; ++rg) {
    if (rs[rg] >= re[rg]) continue;
    FastRec nxt = frow[rs[rg]];
    ColdRec cnxt = crow[rs[rg]];
    for (int l = rs[rg]; l < re[rg]; ++l) {
      const FastRec r = nxt;
      const ColdFull cr = expand_cold(cnxt);
      nxt = frow[l + 1]; // one record of slack behind both tables
      cnxt = crow[l + 1];
      const int j1 = r.j1, jN = j1 + (kImxsig - 1);
      if (jN < wlo || j1 > whi) continue;
      if (classify(j1, r.il(), r.ir(), wlo, whi) != 0) continue; // done by the wings kernel
      const WinX xf{gp.lin_start, gp.lin_delta, grid_at(gp, j1 + kHalf)};
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int slo = wlo + 64 * p, shi = min(slo + 63, whi);
        if (slo > whi || jN < slo || j1 > shi) continue;
        const int scls = classify(j1, r.il(), r.ir(), slo, shi);
        double y;
        if (scls != 0) { // the whole slot in one wing
          const double x = fma(fl[p], r.xstep, wing_x_at(r, scls, j1, wlo));
          const double x2 = x * x;
          const R1Coef rq = r1_of(r);
          y = fma(x2, rq.b, rq.a) * fast_rcp<1>(fma(x2, fma(x2, 4.0, rq.d), rq.c));
        } else {
          const int k = slo + lane - j1 + 1; // 1-based window index
          y = (k >= 1 && k <= kImxsig && slo + lane <= shi) ? humliv_point(k, r, cr, xf) : 0.0;
        }
        acc_a[p] = fma(r.wabs, y, acc_a[p]);
        acc_e[p] = fma(r.wemi, y, acc_e[p]);
      }
    }
  }
  const size_t row = (size_t)layer * (size_t)(g_hi - g_lo);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int j = wlo + lane + 64 * p;
    if (j <= whi) {
      abs_out[row + (j - g_lo)] += acc_a[p];
      emi_out[row + (j - g_lo)] += acc_e[p];
    }
  }
}

// ------------------------------------------------------------------------
// far field by local (Taylor) expansions
//
// Where a whole box of target points lies in one region-1 wing of a line, inside
// its window and at least kTheta box half-widths from its centre, the line's
// contribution w*(a + x^2 b)/(c + x^2(d + 4x^2)), x affine in the grid index, is
// analytic over the box with convergence ratio <= 1/kTheta: its Taylor
// coefficients in t = (j - centre)/half_width come from a 4-term linear
// recurrence (power-series division of a quadratic by a quartic), the
// coefficients of all such lines are summed per box (lanes = lines), and each
// point evaluates ONE polynomial per level instead of one rational per line.
// Truncation at degree kFD = 22 (kTheta = 4) is <= 2.6e-13 of the line's own contribution
// (worst case, nearest admissible line, box edge).  Boxes are nested (width
// 64 << level); a (line, slot) pair is owned by the highest admissible level
// (admissibility is monotone down the hierarchy) or, if none, by the exact
// near-field kernel.  All ownership tests are integer and shared by the kernels.
// ------------------------------------------------------------------------
__device__ inline bool ff_admissible(int j1, int il, int ir, int blo, int bhi, int thr2) {
  if (classify(j1, il, ir, blo, bhi) == 0) return false;
  const int d2 = blo + bhi - 2 * (j1 + kHalf); // twice (box centre - line centre)
  return (d2 < 0 ? -d2 : d2) >= thr2;
}
__device__ inline int ff_thr2(int level, int pm) { return kTheta * (64 << level) + 2 * pm; }

// Box-pair mode (FarParams::m2l).  Source box s and target box t of level l, both counted in boxes of 64 << l
// points from g_lo, o = t - s: the pair is VALID when every line centred in s sees all of t in one region-1 wing
// inside its window, far enough for the multipole / local pair of expansions: gap between the boxes >= the widest
// zone of the layer (zreq) and >= 2 box widths (truncation ratio 1/5: (1/5)^(kFD+1) ~ 1e-16), more where the poles
// of the layer's lines (pms grid points from their centres) are not small against the box (m2l_separated).
// Validity is monotone down the hierarchy (children of a valid pair are valid), so a pair is TRANSLATED at level
// l iff it is valid there and its parent pair is not, and a level-0 pair is covered by some level iff it is valid
// at level 0.  Valid pairs are admissible for every line of s (ff_admissible at level 0), so the near kernels
// need not know about box pairs at all.
__device__ inline bool m2l_separated(int level, int a, int zreq, int pms) { // a = |o|
  const int W = 64 << level;
  // gap >= zone and >= two boxes; (h + pole) <= 0.27 (distance of the centres - h), h = W / 2: 0.27^(kFD+1) = 8e-14
  return a >= 3 && (a - 1) * W >= zreq && 100 * (W + 2 * pms) <= 27 * (2 * a - 1) * W;
}
__device__ inline bool m2l_valid(int level, int o, int zreq, int pms) {
  const int W = 64 << level, a = o < 0 ? -o : o;
  if (!m2l_separated(level, a, zreq, pms)) return false;
  // window [ic - kHalf, ic + kHalf - 1] of every centre ic of s holds all of t
  return (a + 1) * W <= (o > 0 ? kHalf : kHalf + 1);
}
// smallest separated |o| of the level (capped: beyond the window nothing is valid anyway)
__device__ inline int m2l_first_offset(int level, int zreq, int pms) {
  int a = 3;
  while (a < 128 && !m2l_separated(level, a, zreq, pms)) ++a;
  return a;
}

// Sum N per-lane values over the 64 lanes: at step M the pairs (i, i + N/2) are split
// between the lanes with bit M clear / set, an unpaired last value takes a plain butterfly.
// Afterwards v[0] of each lane is the total of value lane_reduce_index(lane) (several lanes
// can hold the same one: `primary` picks one of them).
template <int N, int M>
__device__ inline void lane_reduce(double *v, int lane) {
  if constexpr (M >= 1) {
    constexpr int half = N / 2;
    const bool up = (lane & M) != 0;
#pragma unroll
    for (int i = 0; i < half; ++i) {
      const double keep = up ? v[i + half] : v[i];
      const double send = up ? v[i] : v[i + half];
      v[i] = keep + __shfl_xor(send, M);
    }
    if constexpr (N & 1) v[half] = v[2 * half] + __shfl_xor(v[2 * half], M);
    lane_reduce<half + (N & 1), M / 2>(v, lane);
  }
}
template <int N, int M>
__device__ inline int lane_reduce_index(int lane, bool &primary) {
  if constexpr (M == 0) {
    return 0;
  } else {
    constexpr int half = N / 2;
    const int s = lane_reduce_index<half + (N & 1), M / 2>(lane, primary); // slot after this step
    const bool up = (lane & M) != 0;
    if (s < half) return s + (up ? half : 0);
    if (up) primary = false;
    return 2 * half;
  }
}

// COUNT: the instantiation sr_set_counting(1) selects; it adds the number of (line, box) expansions
// this launch performs to cnt[kCntExpansions] (bench.py's executed-work accounting).  The timed
// instantiation carries no counting code.
// Tuning knob: waves per SIMD to compile for (4: <= 128 VGPRs; 140 unconstrained, 48 B of scratch at 128).  0 (default):
// unconstrained.  Round 4 built this kernel and S2M to 128 VGPRs and padded the zones kernel to 128 so that their waves
// fit the slot a retiring zones wave frees: the pass still ran 4.8 ms beside the zones kernel (the dispatcher serves
// the OLDER dispatch first whenever its next workgroup fits, whatever the younger one needs) and the spills cost
// 0.6 ms of the far-field group alone (1.07 -> 1.66 ms): step 6.07 vs 5.61 ms.
#ifndef SR_FAR_WAVES_PER_EU
#define SR_FAR_WAVES_PER_EU 0
#endif
#if SR_FAR_WAVES_PER_EU > 0
#define SR_FAR_ATTR __attribute__((amdgpu_waves_per_eu(SR_FAR_WAVES_PER_EU)))
#else
#define SR_FAR_ATTR
#endif
template <bool COUNT, bool M2L>
__global__ __launch_bounds__(64) SR_FAR_ATTR void sr_farfield_kernel(const FastRec *__restrict__ fast,
                                                         IcIndex ix,
                                                         const int *__restrict__ zmax, int n_sub, int g_lo,
                                                         int /*g_hi*/, FarParams fp,
                                                         unsigned long long *__restrict__ cnt) {
  // block -> (layer, level, box).  Layers are taken in groups of ceil(n_layers / 8); xcd_remap
  // gives each XCD a contiguous run of work ids, i.e. (about) one group.  Inside a group: first
  // the two widest levels of ALL its layers (a 1024-point box runs ~100 us; left to the end of
  // the last layer, as in a plain layer-major order, those were the tail of the kernel: 0.36 vs
  // 0.27 ms expected on a 1/8 shard), then layer by layer the narrower levels, widest first, so
  // that the blocks resident on an XCD share one stretch of one layer's record row, which its
  // L2 holds.  On the full grid that locality is worth more than the tail (2.14 ms against 2.66 with
  // the wide levels first, 2.76 fully level-major), so the host asks for it on small shards only.
  const int wid = xcd_remap(blockIdx.x, gridDim.x);
  const int lg = (fp.n_layers + 7) / 8;                       // layers per group
  const int grp = wid / (lg * fp.n_boxes_total);
  const int l_first = grp * lg, l_cnt = min(lg, fp.n_layers - l_first);
  int idx = wid - grp * lg * fp.n_boxes_total;                // within the group
  const int n_top = fp.top_first ? min(2, fp.n_levels - 1) : 0; // levels taken first (small shards only)
  int level = fp.n_levels - 1, layer = -1;
  if (M2L) { // box-pair mode: level 0 only (grid = box_count[0] * n_layers), layer-major
    layer = wid / fp.box_count[0];
    idx = wid - layer * fp.box_count[0];
    level = 0;
  }
  for (int t = 0; t < n_top && layer < 0; ++t, --level) {
    const int cnt = l_cnt * fp.box_count[level];
    if (idx < cnt) {
      layer = l_first + idx / fp.box_count[level];
      idx -= (layer - l_first) * fp.box_count[level];
      ++level; // undo the loop's decrement
    } else {
      idx -= cnt;
    }
  }
  if (layer < 0) { // the narrower levels, layer-major
    int per_layer = 0;
    for (int lv = 0; lv <= level; ++lv) per_layer += fp.box_count[lv];
    layer = l_first + idx / per_layer;
    idx -= (layer - l_first) * per_layer;
    if (fp.top_first) {
      while (level > 0 && idx >= fp.box_count[level]) {
        idx -= fp.box_count[level];
        --level;
      }
    } else {
      // Full grids: inside a layer the boxes of ALL levels are taken super-tile by super-tile (kFarSuper
      // points = 16 top-level boxes).  A level-major sweep over the whole grid touches the layer's whole
      // record row (8 MB for 1e5 lines) once per level -- more than an XCD's 4 MB L2 holds, so every level
      // re-read it from HBM (4.7 GB per step); the records of one super-tile and its window halo
      // ((16384 + 13010) lines x 80 B = 2.3 MB) stay in L2 across the five levels.
      const int top_lv = fp.n_levels - 1;
      const int per_super = (kFarSuper >> 6) * 2 - (kFarSuper >> (6 + top_lv)); // sum over levels of kFarSuper / W
      const int n_full = fp.box_count[top_lv] > 0 ? (fp.box_count[0] * 64) / kFarSuper : 0; // complete super-tiles
      int st = idx / per_super;
      if (st >= n_full) st = n_full; // the last, partial super-tile holds the rest
      idx -= st * per_super;
      level = top_lv;
      for (;;) {
        const int per = kFarSuper >> (6 + level);                  // boxes of this level per super-tile
        const int first = st * per;
        const int cnt = min(per, max(fp.box_count[level] - first, 0));
        if (idx < cnt || level == 0) {
          idx += first;
          break;
        }
        idx -= cnt;
        --level;
      }
    }
  }
  const int b = idx;
  const int lane = threadIdx.x;
  const int W = 64 << level, h = W >> 1;
  const int blo = g_lo + b * W, bhi = blo + W - 1;
  const int pm = fp.pm[layer];
  const int zm = min(zmax[layer], kHalf - 1);
  const int thr2 = ff_thr2(level, pm);
  const bool top = level == fp.n_levels - 1 && !M2L;
  const int W2 = 2 * W, plo = g_lo + (b >> 1) * W2, phi = plo + W2 - 1, thr2p = ff_thr2(level + 1, pm);
  const int zreq = zm + 2, pms = M2L ? fp.pm_src[layer] : 0; // box-pair mode: see m2l_valid

  // candidate centre intervals [lo, hi] (inclusive), see DESIGN.md
  int clo[4], chi[4], nr, n_extra = 0;
  const int mid = blo + h, near_in = kTheta * h + pm;
  if (M2L) {
    // the (line, box) pairs no box pair covers: source boxes closer than the first valid offset, and the two
    // source boxes on either side whose lines' windows end inside or just beyond this box (|o| = 101, 102)
    const int a_min = m2l_first_offset(0, zreq, pms);
    // (of those only the lines whose window holds the whole box: centre in [bhi - kHalf + 1, blo + kHalf])
    clo[0] = bhi - kHalf; chi[0] = blo - 100 * 64;
    clo[1] = blo - (a_min - 1) * 64 - 1; chi[1] = mid - near_in + 1;
    clo[2] = mid + near_in - 2; chi[2] = bhi + (a_min - 1) * 64 + 1;
    clo[3] = blo + 101 * 64 - 1; chi[3] = blo + kHalf + 1;
    nr = 4;
    n_extra = 2; // + the lines beyond the grid ends (FarParams::disp_lo_end)
  } else if (top) {
    clo[0] = blo - kHalf - 1; chi[0] = mid - near_in + 1;
    clo[1] = mid + near_in - 2; chi[1] = bhi + kHalf + 1;
    nr = 2;
  } else {
    const int bn = max(2 * kTheta * h + h + pm, zm + 3 * h) + 2;
    clo[0] = plo - kHalf; chi[0] = phi - (kHalf - 1) + 1;       // window end inside the parent
    clo[1] = mid - bn - 1; chi[1] = mid - near_in + 1;           // left near band
    clo[2] = mid + near_in - 2; chi[2] = mid + bn + 1;           // right near band
    clo[3] = plo + kHalf - 1; chi[3] = phi + kHalf + 1;          // window start inside the parent
    nr = 4;
  }
  int rs[6], re[6];
  for (int i = 0; i < nr; ++i) {
    rs[i] = lower_bound_ic(ix, clo[i]);
    re[i] = lower_bound_ic(ix, chi[i] + 1);
  }
  if (n_extra) {
    rs[nr] = 0; re[nr] = fp.disp_lo_end;
    rs[nr + 1] = fp.disp_hi_begin; re[nr + 1] = n_sub;
    nr += 2;
  }
  for (int i = 1; i < nr; ++i) // sort by start
    for (int k = i; k > 0 && rs[k] < rs[k - 1]; --k) {
      int t0 = rs[k]; rs[k] = rs[k - 1]; rs[k - 1] = t0;
      t0 = re[k]; re[k] = re[k - 1]; re[k - 1] = t0;
    }
  int done = 0; // make disjoint
  for (int i = 0; i < nr; ++i) {
    rs[i] = max(rs[i], done);
    re[i] = max(re[i], rs[i]);
    done = re[i];
  }

  double v[2 * kFC]; // [0, kFC): abs coefficients, [kFC, 2 kFC): emi
#pragma unroll
  for (int n = 0; n < 2 * kFC; ++n) v[n] = 0.;
  const FastRec *frow = fast + (size_t)layer * n_sub;
  const double hw = (double)h;
  unsigned n_exp = 0;
  // The candidate ranges as ONE list (they are disjoint and sorted): chunks of 64 across the range ends.  Taken range by
  // range, every range ended in a partly filled chunk, and a chunk costs its ~230 (8-term series: ~110) instructions
  // whatever its lanes hold: 5.2 chunk bodies per box on config 2 for 168 expansions.
  int off[7];
  off[0] = 0;
  for (int i = 0; i < 6; ++i) off[i + 1] = off[i] + (i < nr ? re[i] - rs[i] : 0);
  const int total = off[6];
  {
    for (int base = 0; base < total; base += 64) {
      const int g = base + lane;
      if (g >= total) continue;
      int l = rs[0] + g;
#pragma unroll
      for (int i = 1; i < 6; ++i)
        if (i < nr && g >= off[i]) l = rs[i] + (g - off[i]);
      const FastRec r = frow[l];
      const int j1 = r.j1, il = r.il(), ir = r.ir();
      if (!ff_admissible(j1, il, ir, blo, bhi, thr2)) continue;
      if (M2L) {
        const bool displaced = l < fp.disp_lo_end || l >= fp.disp_hi_begin;
        if (!displaced && m2l_valid(0, b - ((j1 + kHalf - g_lo) >> 6), zreq, pms)) continue; // covered by a box pair
      } else if (!top && ff_admissible(j1, il, ir, plo, phi, thr2p)) {
        continue; // owned by a wider box
      }
      const int cls = classify(j1, il, ir, blo, bhi);
      if (COUNT) ++n_exp;
      // x (or -x) at the box centre blo + h - 1/2, and the half-width in x units
      const double xc = cls == 1 ? fma(0.5 * (double)(2 * (blo - j1) + W - 1), r.xstep, -r.xl)
                                 : fma(0.5 * (double)(2 * (blo - (j1 + ir - 1)) + W - 1), r.xstep, r.xr);
      const double e = hw * r.xstep;
      const double u0 = xc * xc, u1 = 2. * xc * e, u2 = e * e;
      const R1Coef rq = r1_of(r);
      const double n0 = fma(rq.b, u0, rq.a), n1 = rq.b * u1, n2 = rq.b * u2;
      const double d0 = fma(u0, fma(4., u0, rq.d), rq.c);
      const double d1 = u1 * fma(8., u0, rq.d);
      const double d2 = fma(u2, rq.d, 4. * fma(u1, u1, 2. * u0 * u2));
      const double d3 = 8. * u1 * u2, d4 = 4. * u2 * u2;
      const double r0 = fast_rcp<2>(d0);
      const double D1 = d1 * r0, D2 = d2 * r0, D3 = d3 * r0, D4 = d4 * r0;
      // series coefficients f_n, accumulated as they are produced (a window of four is live)
      double f0 = n0 * r0;
      double f1 = fma(-D1, f0, n1 * r0);
      double f2 = fma(-D1, f1, fma(-D2, f0, n2 * r0));
      double f3 = -fma(D1, f2, fma(D2, f1, D3 * f0));
      v[0] = fma(r.wabs, f0, v[0]); v[kFC + 0] = fma(r.wemi, f0, v[kFC + 0]);
      v[1] = fma(r.wabs, f1, v[1]); v[kFC + 1] = fma(r.wemi, f1, v[kFC + 1]);
      v[2] = fma(r.wabs, f2, v[2]); v[kFC + 2] = fma(r.wemi, f2, v[kFC + 2]);
      v[3] = fma(r.wabs, f3, v[3]); v[kFC + 3] = fma(r.wemi, f3, v[kFC + 3]);
      // Box-pair mode: the lines of the window bands are > 6000 points away -- the series converges like
      // (32/6000)^n, eight terms carry it to 1e-18 of the leading one.  The bands are disjoint runs of the sorted
      // list, so whole chunks take the short path.
      constexpr int kShort = kFC < 8 ? kFC : 8;
      const int d_abs = blo - (j1 + kHalf);
      if (M2L && __all((d_abs < 0 ? -d_abs : d_abs) > 4096)) {
#pragma unroll
        for (int n = 4; n < kShort; ++n) {
          const double fn = -fma(D1, f3, fma(D2, f2, fma(D3, f1, D4 * f0)));
          v[n] = fma(r.wabs, fn, v[n]);
          v[kFC + n] = fma(r.wemi, fn, v[kFC + n]);
          f0 = f1; f1 = f2; f2 = f3; f3 = fn;
        }
        continue;
      }
#pragma unroll
      for (int n = 4; n < kFC; ++n) {
        const double fn = -fma(D1, f3, fma(D2, f2, fma(D3, f1, D4 * f0)));
        v[n] = fma(r.wabs, fn, v[n]);
        v[kFC + n] = fma(r.wemi, fn, v[kFC + n]);
        f0 = f1; f1 = f2; f2 = f3; f3 = fn;
      }
    }
  }
  // Sum the 2*kFC coefficient sets over the lanes.  A butterfly per value would cost
  // 6 exchanges each (480 ds_bpermute per box); exchanging HALF of the values at every
  // step instead (the lanes with bit m set keep the upper half) needs 2*kFC - 1 in all
  // and leaves one finished sum per lane.
  lane_reduce<2 * kFC, 32>(v, lane);
  bool primary = true;
  const int n_out = lane_reduce_index<2 * kFC, 32>(lane, primary);
  if (primary)
    fp.coef[((size_t)layer * fp.n_boxes_total + fp.box_off[level] + b) * (2 * kFC) + n_out] = v[0];
  if (COUNT) count_add(cnt, kCntExpansions, n_exp, lane);
}

// The per-line scheme for SPARSE line sets (the per-level passes of the pair tables: 0.09-0.16 lines per grid point):
// sr_farfield_kernel gives a (box, level, layer) a wave whatever the box holds -- five or six lines in one chunk body
// of ~230 instructions at a tenth of its lanes, behind ~450 of range searches, reduction and indexing: 0.34 of the pass's
// 0.53 ms were that.  Here a wave takes a box for kFarRows LAYERS: lane = (line of the chunk, layer), kFarLines lines x
// kFarRows layers per chunk body (8 x 8 when this was written, 4 x 16 as built now); the candidate ranges are the union over the wave's layers (the exact admissibility tests
// run per lane with its layer's margins), the 2 kFC sums are reduced over the kFarLines lanes of a layer (log2 of them exchange
// steps instead of six).  Same expansions, same owner of every (line, box): the coefficients differ from
// sr_farfield_kernel<., false>'s by the summation order.
#ifndef SR_FAR_ROWS
#define SR_FAR_ROWS 16 // layers per wave x 64 / that many lines.  8 x 8 through round 6 (degree 22: 16 x 4 and 4 x 16 lost, tools/r04_ab.sh);
                       // at degree 19 a table build of 12 levels: 16 x 4 11.94 / 23.92 ms (1e5 / 2e5 lines), 8 x 8 12.27 / 24.42, 4 x 16 13.45 / 26.74
#endif
constexpr int kFarRows = SR_FAR_ROWS, kFarLines = 64 / kFarRows; // lanes: kFarRows layers x kFarLines lines of a chunk
static_assert(kFarRows == 4 || kFarRows == 8 || kFarRows == 16, "SR_FAR_ROWS");
template <int N, int M>
__device__ inline void lane_reduce_slots(int lane, int *idx) { // which of the N values v[i] holds after lane_reduce<N, M>
  if constexpr (M >= 1) {
    constexpr int half = N / 2;
    const bool up = (lane & M) != 0;
#pragma unroll
    for (int i = 0; i < half; ++i) idx[i] = up ? idx[i + half] : idx[i];
    if constexpr (N & 1) idx[half] = idx[2 * half];
    lane_reduce_slots<half + (N & 1), M / 2>(lane, idx);
  }
}
template <int N, int M>
constexpr int lane_reduce_left() { // values per lane after lane_reduce<N, M>
  if constexpr (M >= 1) return lane_reduce_left<N / 2 + (N & 1), M / 2>(); else return N;
}
// NW waves per workgroup, each with a box of its own (wid: the wave's work id, < n_boxes_total x layer groups)
template <bool COUNT, int NW>
__device__ __forceinline__ void farfield_rows_body(const FastRec *__restrict__ fast, const IcIndex ix,
                                                   const int *__restrict__ zmax, int n_sub, int g_lo, const FarParams &fp,
                                                   unsigned long long *__restrict__ cnt, const int wid) {
  const int grp = wid / fp.n_boxes_total; // group of kFarRows layers
  int idx = wid - grp * fp.n_boxes_total, level = fp.n_levels - 1;
  while (level > 0 && idx >= fp.box_count[level]) { // widest level first
    idx -= fp.box_count[level];
    --level;
  }
  const int b = idx;
  const int lane = threadIdx.x & 63, sub = lane % kFarLines, layer = grp * kFarRows + lane / kFarLines;
  const bool live = layer < fp.n_layers;
  const int lc = min(layer, fp.n_layers - 1);
  const int W = 64 << level, h = W >> 1;
  const int blo = g_lo + b * W, bhi = blo + W - 1;
  const int pm = fp.pm[lc];
  const int zm = min(zmax[lc], kHalf - 1);
  const int thr2 = ff_thr2(level, pm);
  const bool top = level == fp.n_levels - 1;
  const int W2 = 2 * W, plo = g_lo + (b >> 1) * W2, phi = plo + W2 - 1, thr2p = ff_thr2(level + 1, pm);
  // margins of the wave's layers: the candidate ranges must hold every layer's candidates
  int pm_min = pm, pm_max = pm, zm_max = zm;
#pragma unroll
  for (int m = kFarLines; m < 64; m <<= 1) {
    pm_min = min(pm_min, __shfl_xor(pm_min, m));
    pm_max = max(pm_max, __shfl_xor(pm_max, m));
    zm_max = max(zm_max, __shfl_xor(zm_max, m));
  }
  pm_min = __builtin_amdgcn_readfirstlane(pm_min);
  pm_max = __builtin_amdgcn_readfirstlane(pm_max);
  zm_max = __builtin_amdgcn_readfirstlane(zm_max);
  // (the ranges live in LDS: as local arrays their dynamic indices put them in scratch memory -- 32 bytes per lane, and
  // a wave that owns scratch is dispatched through the scratch ring)
  __shared__ int rs_all[NW][4], re_all[NW][4];
  int *rs = rs_all[threadIdx.x >> 6], *re = re_all[threadIdx.x >> 6];
  int nr;
  const int mid = blo + h, near_in = kTheta * h + pm_min;
  if (top) {
    const int a0 = lower_bound_ic(ix, blo - kHalf - 1), b0 = lower_bound_ic(ix, mid - near_in + 1 + 1);
    const int a1 = lower_bound_ic(ix, mid + near_in - 2), b1 = lower_bound_ic(ix, bhi + kHalf + 1 + 1);
    rs[0] = a0; re[0] = b0; rs[1] = a1; re[1] = b1;
    nr = 2;
  } else {
    const int bn = max(2 * kTheta * h + h + pm_max, zm_max + 3 * h) + 2;
    const int a0 = lower_bound_ic(ix, plo - kHalf), b0 = lower_bound_ic(ix, phi - (kHalf - 1) + 1 + 1); // window end inside the parent
    const int a1 = lower_bound_ic(ix, mid - bn - 1), b1 = lower_bound_ic(ix, mid - near_in + 1 + 1);   // left near band
    const int a2 = lower_bound_ic(ix, mid + near_in - 2), b2 = lower_bound_ic(ix, mid + bn + 1 + 1);   // right near band
    const int a3 = lower_bound_ic(ix, plo + kHalf - 1), b3 = lower_bound_ic(ix, phi + kHalf + 1 + 1);  // window start inside the parent
    rs[0] = a0; re[0] = b0; rs[1] = a1; re[1] = b1; rs[2] = a2; re[2] = b2; rs[3] = a3; re[3] = b3;
    nr = 4;
  }
  for (int i = 1; i < nr; ++i) // sort by start
    for (int k = i; k > 0 && rs[k] < rs[k - 1]; --k) {
      int t0 = rs[k]; rs[k] = rs[k - 1]; rs[k - 1] = t0;
      t0 = re[k]; re[k] = re[k - 1]; re[k - 1] = t0;
    }
  int done = 0; // make disjoint
  for (int i = 0; i < nr; ++i) {
    rs[i] = max(rs[i], done);
    re[i] = max(re[i], rs[i]);
    done = re[i];
  }
  // Adjacent ranges merge; the chunks are aligned on multiples of eight LINE INDICES, lane sub takes the lines with
  // l = sub (mod 8) in increasing order: which lane adds which line, and in which order, then depends on the layer's
  // own admissible lines alone -- not on the other layers of the wave, whose margins widen the ranges -- and a layer
  // stack processed in batches (sr_set_table_budget) gives the bits of the unbatched call.
  int nm = 0;
  for (int i = 0; i < nr; ++i) {
    if (re[i] <= rs[i]) continue;
    if (nm > 0 && rs[i] <= ((re[nm - 1] + kFarLines - 1) & ~(kFarLines - 1))) {
      re[nm - 1] = re[i]; // (the lines between the two ranges are no candidates of any layer: the exact tests reject them)
    } else {
      rs[nm] = rs[i]; re[nm] = re[i]; ++nm;
    }
  }

  // (Round 6 tried two phases -- the admissibility tests over all candidates first, every layer row queueing its
  // admissible lines in LDS, then eight queued lines of a row at a time: 333 -> 1109 us per sparse pass; the candidates
  // of a box are mostly its own already, the queue's round trip and the second record fetch were pure cost.
  // And two chunks at a time -- both records requested first, the two chains of 4 x kFC dependent fma interleaved, chunks
  // without an admissible lane skipped as here; the same doubles -- : 168 VGPRs, still three waves per SIMD, a table build
  // 13.0 -> 13.6 ms (1e5 x 1e5 x 80) and 25.2 -> 26.4 (2e5), same box: the chain's latency is not what the kernel waits for.)
  double v[2 * kFC];
#pragma unroll
  for (int n = 0; n < 2 * kFC; ++n) v[n] = 0.;
  const FastRec *frow = fast + (size_t)lc * n_sub;
  const double hw = (double)h;
  unsigned n_exp = 0;
  for (int i = 0; i < nm; ++i) {
  const int rsi = __builtin_amdgcn_readfirstlane(rs[i]), rei = __builtin_amdgcn_readfirstlane(re[i]);
  for (int l0 = rsi & ~(kFarLines - 1); l0 < rei; l0 += kFarLines) {
    const int l = l0 + sub;
    if (l < rsi || l >= rei || !live) continue;
    const FastRec r = frow[l];
    const int j1 = r.j1, il = r.il(), ir = r.ir();
    if (!ff_admissible(j1, il, ir, blo, bhi, thr2)) continue;
    if (!top && ff_admissible(j1, il, ir, plo, phi, thr2p)) continue; // owned by a wider box
    const int cls = classify(j1, il, ir, blo, bhi);
    if (COUNT) ++n_exp;
    const double xc = cls == 1 ? fma(0.5 * (double)(2 * (blo - j1) + W - 1), r.xstep, -r.xl)
                               : fma(0.5 * (double)(2 * (blo - (j1 + ir - 1)) + W - 1), r.xstep, r.xr);
    const double e = hw * r.xstep;
    const double u0 = xc * xc, u1 = 2. * xc * e, u2 = e * e;
    const R1Coef rq = r1_of(r);
    const double n0 = fma(rq.b, u0, rq.a), n1 = rq.b * u1, n2 = rq.b * u2;
    const double d0 = fma(u0, fma(4., u0, rq.d), rq.c);
    const double d1 = u1 * fma(8., u0, rq.d);
    const double d2 = fma(u2, rq.d, 4. * fma(u1, u1, 2. * u0 * u2));
    const double d3 = 8. * u1 * u2, d4 = 4. * u2 * u2;
    const double r0 = fast_rcp<2>(d0);
    const double D1 = d1 * r0, D2 = d2 * r0, D3 = d3 * r0, D4 = d4 * r0;
    double f0 = n0 * r0;
    double f1 = fma(-D1, f0, n1 * r0);
    double f2 = fma(-D1, f1, fma(-D2, f0, n2 * r0));
    double f3 = -fma(D1, f2, fma(D2, f1, D3 * f0));
    v[0] = fma(r.wabs, f0, v[0]); v[kFC + 0] = fma(r.wemi, f0, v[kFC + 0]);
    v[1] = fma(r.wabs, f1, v[1]); v[kFC + 1] = fma(r.wemi, f1, v[kFC + 1]);
    v[2] = fma(r.wabs, f2, v[2]); v[kFC + 2] = fma(r.wemi, f2, v[kFC + 2]);
    v[3] = fma(r.wabs, f3, v[3]); v[kFC + 3] = fma(r.wemi, f3, v[kFC + 3]);
#pragma unroll
    for (int n = 4; n < kFC; ++n) {
      const double fn = -fma(D1, f3, fma(D2, f2, fma(D3, f1, D4 * f0)));
      v[n] = fma(r.wabs, fn, v[n]);
      v[kFC + n] = fma(r.wemi, fn, v[kFC + n]);
      f0 = f1; f1 = f2; f2 = f3; f3 = fn;
    }
  }
  }
  // sums over the eight lanes of a layer (lane bits 4, 2, 1); every lane is left with kLeft finished values
  lane_reduce<2 * kFC, kFarLines / 2>(v, lane);
  constexpr int kLeft = lane_reduce_left<2 * kFC, kFarLines / 2>();
  int slot[2 * kFC];
#pragma unroll
  for (int n = 0; n < 2 * kFC; ++n) slot[n] = n;
  lane_reduce_slots<2 * kFC, kFarLines / 2>(lane, slot);
  if (live) {
    double *out = fp.coef + ((size_t)layer * fp.n_boxes_total + fp.box_off[level] + b) * (2 * kFC);
#pragma unroll
    for (int n = 0; n < kLeft; ++n) out[slot[n]] = v[n]; // (an unpaired value is held by two lanes: the same sum twice)
  }
  if (COUNT) count_add(cnt, kCntExpansions, n_exp, lane);
}
#ifndef SR_ROWS_WAVES_PER_EU
#define SR_ROWS_WAVES_PER_EU 0 // tuning knob: waves per SIMD to compile the sparse sets' kernels for (0: unconstrained, 138 VGPRs = 3; 4: 128 VGPRs + 48 B of scratch in the batch kernel: a table build 13.4 -> 15.1 ms)
#endif
#if SR_ROWS_WAVES_PER_EU > 0
#define SR_ROWS_ATTR __attribute__((amdgpu_waves_per_eu(SR_ROWS_WAVES_PER_EU)))
#else
#define SR_ROWS_ATTR
#endif
template <bool COUNT>
__global__ __launch_bounds__(64) SR_ROWS_ATTR void sr_farfield_rows_kernel(const FastRec *__restrict__ fast, IcIndex ix,
                                                              const int *__restrict__ zmax, int n_sub, int g_lo, FarParams fp,
                                                              unsigned long long *__restrict__ cnt) {
  farfield_rows_body<COUNT, 1>(fast, ix, zmax, n_sub, g_lo, fp, cnt, xcd_remap_groups((int)blockIdx.x, (int)gridDim.x, 16));
}
// ... of every sparse far-only pass of a table build in one launch (grid.y = item): one at a time these launches --
// eleven of 0.3 ms, latency-bound, each behind ~0.3 ms of host calls -- were 6 ms of a 13 ms build
#ifndef SR_ROWS_BATCH_WAVES
#define SR_ROWS_BATCH_WAVES 4
#endif
constexpr int kRowsBatchWaves = SR_ROWS_BATCH_WAVES; // waves (boxes) per workgroup of the batch kernel
__global__ __launch_bounds__(64 * kRowsBatchWaves) SR_ROWS_ATTR void sr_farfield_rows_batch_kernel(const FarBatchItem *__restrict__ items, const int *__restrict__ zmax,
                                                                    int g_lo, FarParams fp, int n_work) {
  const FarBatchItem it = items[blockIdx.y];
  fp.coef = it.coef;
  const int wid = xcd_remap_groups((int)blockIdx.x, (int)gridDim.x, 16 / kRowsBatchWaves) * kRowsBatchWaves + (int)(threadIdx.x >> 6);
  if (wid >= n_work) return; // (wave-uniform; the kernel has no block barrier)
  farfield_rows_body<false, kRowsBatchWaves>(it.fast, IcIndex{it.first, it.first_x0, it.first_n, it.line_lo, it.n_sub}, zmax, it.n_sub, g_lo, fp,
                                             nullptr, wid);
}

// ------------------------------------------------------------------------
// Box-pair far field (FarParams::m2l): multipole moments, upward pass, translations.
//
// Region 1 of a line is w g(x), g(x) = (a + b x^2)/(c + d x^2 + 4 x^4), x = xstep (j - j0) with j0 the (real)
// grid position where the reference's running x of that wing passes through zero.  At infinity
// g = sum_n e_(n-1) x^(-2n), e_0 = b/4, e_1 = (a - d e_0)/4, e_k = -(d e_(k-1) + c e_(k-2))/4, and about the
// centre C of a source box of half-width h (delta = j0 - C, J = j - C):
//   (J - delta)^(-2n) = sum_m binom(2n + m - 1, m) delta^m J^(-2n-m)
//   =>  g = sum_q M_q J^(-q),  M_q / (h^q (q-1)!) = sum_(2n+m=q) [e_(n-1) (xstep h)^(-2n)/(2n-1)!] [(delta/h)^m/m!]
// -- a convolution of two short series per line, summed over the lines of the box with lanes = lines.  The
// scaled moments mh_q of a parent box follow from its children (centres -+ h/2 from its own) by
//   mh_q = 2^-q sum_(m=0..q-2) (mhL_(q-m) (-1)^m + mhR_(q-m))/m!
// and the local coefficients (in t = (j - centre)/h, the polynomial each point evaluates) of a target box
// o boxes away by  L_n = sum_q T_o[n][q] mh_q,  T_o[n][q] = (-1)^n (q+n-1)!/n! (1/2o)^(q+n)  (host table).
// The two wings of a line use different anchors (xl from x(1), xr from x(ir): lineshape.f:462, 471; they differ
// by the rounding of the grid, ~1e-9 points), hence two sets of moments: side 0 serves targets right of the
// source box, side 1 targets left of it.
// ------------------------------------------------------------------------
__device__ constexpr double inv_factorial(int n) {
  double f = 1.0;
  for (int i = 2; i <= n; ++i) f *= (double)i;
  return 1.0 / f;
}
// One wave per (source box, layer); lane = (line % 32, side): the lower half-wave builds the right-going moments
// of 32 lines at a time, the upper half the left-going ones of the same lines, and each half sums its 2 kMQ values
// over its own 32 lanes.  (A box holds 64 +- 8 lines on config 2: with 64 lines per step and one wave per side the
// second step of most boxes ran nearly empty.)
// as SR_FAR_WAVES_PER_EU (154 VGPRs unconstrained, 104 B of scratch at 128).  0 (default): unconstrained.
#ifndef SR_S2M_WAVES_PER_EU
#define SR_S2M_WAVES_PER_EU 0
#endif
#if SR_S2M_WAVES_PER_EU > 0
#define SR_S2M_ATTR __attribute__((amdgpu_waves_per_eu(SR_S2M_WAVES_PER_EU)))
#else
#define SR_S2M_ATTR
#endif
// (one convolution per line about the MEAN of its two anchors + first-order corrections; round 3's one-per-side
// variant is in the history: S2M 0.375 vs 0.31 ms)
// Round 4: one moment set per line instead of one per side.  The two wings of a line use different anchors (the zero of
// the reference's running x: xl from x(1), xr from x(ir): lineshape.f:462, 471), eps ~ 1e-9 grid points apart (the
// rounding of the grid).  The moments are analytic in the anchor: M_q(delta +- eps / 2) = M_q(delta) +- (eps / 2) d M_q /
// d delta + O(eps^2 ~ 1e-18), and d m_q / d delta = m_(q-1) / h for the scaled per-line moments (the delta-series
// D_m = (delta / h)^m / m! shifts by one index).  So: ONE convolution per line about the mean anchor (lane = line, 64
// lines per step instead of 32 per side: a box holds 64 +- 8 lines -- 1.5 steps on average instead of 2.5), the sums
// v_q, and correction sums c_q = sum_i w_i (eps_i / 2 h) m_(q-1, i) for q = 3..6 only -- the correction of order q
// reaches a target J >= 192 points away with (h / J)^(q - 2) of a term that is 1e-11 of the pair's contribution to begin
// with: beyond q = 6 that is < 1e-15.  Written out: side 0 (right-going) = v + c, side 1 (left-going) = v - c: M2M and
// M2L read what they always read.
constexpr int kS2MCorr = 4; // corrected orders q = 3 .. 2 + kS2MCorr
template <bool COUNT>
__global__ __launch_bounds__(64) SR_S2M_ATTR void sr_s2m_kernel(const FastRec *__restrict__ fast, IcIndex ix, int n_sub, int g_lo,
                                                    FarParams fp, unsigned long long *__restrict__ cnt) {
  const int wid = xcd_remap(blockIdx.x, gridDim.x);
  const int layer = wid / fp.n_src[0], sb = wid - layer * fp.n_src[0];
  const int lane = threadIdx.x;
  const int s_lo = g_lo + (sb - kSrcPad) * 64; // first centre position of the box
  // A box wholly LEFT of the shard only ever serves targets to its right (side 0), one wholly right of it targets to
  // its left (side 1) -- and so do all its ancestors: the side nobody reads is written as zeros (sr_m2m_kernel skips it).
  const int only = s_lo + 64 <= g_lo ? 0 : (s_lo >= g_lo + fp.box_count[0] * 64 ? 1 : -1);
  const int l0 = lower_bound_ic(ix, s_lo), l1 = lower_bound_ic(ix, s_lo + 64);
  constexpr int NE = kFD / 2; // terms of the Laurent series
  double v[2 * kMQ];          // [0, kMQ): abs, [kMQ, 2 kMQ): emi -- about the mean anchor
  double c[2 * kS2MCorr];     // first-order anchor corrections of orders 3 .. 2 + kS2MCorr: abs, emi
#pragma unroll
  for (int n = 0; n < 2 * kMQ; ++n) v[n] = 0.;
#pragma unroll
  for (int n = 0; n < 2 * kS2MCorr; ++n) c[n] = 0.;
  const FastRec *frow = fast + (size_t)layer * n_sub;
  unsigned n_lines = 0;
  for (int base = l0; base < l1; base += 64) {
    const int l = base + lane;
    if (l >= l1 || l < fp.disp_lo_end || l >= fp.disp_hi_begin) continue;
    const FastRec r = frow[l];
    if (COUNT) n_lines += 2; // (line, side) pairs served
    constexpr double h = 32.0;
    const double sh = r.xstep * h;
    const double V = fast_rcp<2>(sh * sh);
    double E[NE]; // e_(n-1) (xstep h)^(-2n) / (2n-1)!
    {
      const R1Coef rq = r1_of(r);
      const double dV = 0.25 * rq.d * V, cV = 0.25 * rq.c * (V * V);
      E[0] = 0.25 * rq.b * V;
      E[1] = fma(0.25 * rq.a * V, V, -dV * E[0]);
#pragma unroll
      for (int k = 2; k < NE; ++k) E[k] = -fma(dV, E[k - 1], cV * E[k - 2]);
#pragma unroll
      for (int k = 1; k < NE; ++k) E[k] *= inv_factorial(2 * k + 1);
    }
    // zeros of the two running x, relative to the box centre s_lo + 31.5 (grid points)
    const double inv_xs = fast_rcp<2>(r.xstep);
    const double off_r = (double)(r.j1 + r.ir() - 1 - s_lo - 32) + (0.5 - r.xr * inv_xs);
    const double off_l = (double)(r.j1 - s_lo - 32) + (0.5 + r.xl * inv_xs);
    const double dt = (0.5 / h) * (off_r + off_l);   // mean anchor / h
    const double eh = (0.5 / h) * (off_r - off_l);   // eps / 2 h: side 0 sits at dt + eh, side 1 at dt - eh
    double D[kMQ]; // (delta/h)^m / m!, m = 0..kFD-2
    D[0] = 1.0;
#pragma unroll
    for (int m = 1; m < kMQ; ++m) D[m] = D[m - 1] * (dt * (1.0 / (double)m));
    const double wa_e = r.wabs * eh, we_e = r.wemi * eh;
#pragma unroll
    for (int q = 2; q <= kFD; ++q) {
      double mq = 0.;
#pragma unroll
      for (int n = 1; 2 * n <= q; ++n) mq = fma(E[n - 1], D[q - 2 * n], mq);
      v[q - 2] = fma(r.wabs, mq, v[q - 2]);
      v[kMQ + q - 2] = fma(r.wemi, mq, v[kMQ + q - 2]);
      if (q - 2 < kS2MCorr) { // d m_(q+1) / d delta = m_q / h
        c[q - 2] = fma(wa_e, mq, c[q - 2]);
        c[kS2MCorr + q - 2] = fma(we_e, mq, c[kS2MCorr + q - 2]);
      }
    }
  }
  // sums over the 64 lanes: the moments by halving exchanges (one finished value per lane), the few corrections by butterflies
  lane_reduce<kMQ, 32>(v, lane);
  lane_reduce<kMQ, 32>(v + kMQ, lane);
#pragma unroll
  for (int n = 0; n < 2 * kS2MCorr; ++n)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) c[n] += __shfl_xor(c[n], m);
  bool primary = true;
  const int n_out = lane_reduce_index<kMQ, 32>(lane, primary); // this lane holds order q = n_out + 2
  if (primary) {
    double ca = 0., ce = 0.; // correction of order q = n_out + 2: c index q - 3
#pragma unroll
    for (int i = 0; i < kS2MCorr; ++i) {
      ca = n_out == i + 1 ? c[i] : ca;
      ce = n_out == i + 1 ? c[kS2MCorr + i] : ce;
    }
    double *mo = fp.mom + ((size_t)(fp.src_off[0] + sb) * fp.n_layers + layer) * kMomPerBox + n_out;
    const bool z0 = only == 1, z1 = only == 0; // the side a halo box never serves
    mo[0] = z0 ? 0.0 : v[0] + ca;                       // side 0 (right-going), abs
    mo[kMQ] = z0 ? 0.0 : v[kMQ] + ce;                   //                       emi
    mo[2 * kMQ] = z1 ? 0.0 : v[0] - ca;                 // side 1 (left-going), abs
    mo[3 * kMQ] = z1 ? 0.0 : v[kMQ] - ce;
  }
  if (COUNT) count_add(cnt, kCntS2M, n_lines, lane);
}

// One thread per (parent box of level l, layer, side, weight); one launch per level, narrow to wide (round 3: one
// thread per widest-level subtree walked all its 15 boxes in sequence -- 500 waves with 15 dependent rounds of loads:
// 70 us alone on config 2, 130-210 us in the step's traces, where this kernel runs on an otherwise idle chip).
__global__ __launch_bounds__(64) void sr_m2m_kernel(FarParams fp, int l) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= fp.n_src[l] * fp.n_layers * 4) return;
  const int sw = tid & 3, rest = tid >> 2;
  const int layer = rest % fp.n_layers, p = rest / fp.n_layers;
  // halo boxes: the side nobody reads is not built (see sr_s2m_kernel); sw = side * 2 + weight
  if ((sw >> 1) == 1 ? ((p + 1) << l) <= kSrcPad : (p << l) - kSrcPad >= fp.box_count[0]) return;
  const double *cl = fp.mom + ((size_t)(fp.src_off[l - 1] + 2 * p) * fp.n_layers + layer) * kMomPerBox + sw * kMQ;
  const double *cr = cl + (size_t)fp.n_layers * kMomPerBox;
  double *par = fp.mom + ((size_t)(fp.src_off[l] + p) * fp.n_layers + layer) * kMomPerBox + sw * kMQ;
  double a[kMQ], b[kMQ];
#pragma unroll
  for (int q = 0; q < kMQ; ++q) {
    a[q] = cl[q];
    b[q] = cr[q];
  }
  double scale = 0.25; // 2^-q
#pragma unroll
  for (int q = 2; q <= kFD; ++q) {
    double s = 0.;
#pragma unroll
    for (int m = 0; m <= q - 2; ++m) {
      const double pair = (m & 1) ? b[q - m - 2] - a[q - m - 2] : b[q - m - 2] + a[q - m - 2];
      s = fma(pair, inv_factorial(m), s);
    }
    par[q - 2] = s * scale;
    scale *= 0.5;
  }
}

// Translations: L[pair][n] += sum_q mh[pair][q] T_o[q][n] is a small dense product per box offset o, the one
// place of this path with the shape of a matrix multiplication: v_mfma_f64_16x16x4_f64 with A = the moments of 16
// (target box, layer) pairs (rows) and B = the operator (16 of the kFC columns).  The matrix unit is no faster than
// the vector unit in fp64, but one lane-load of A and of B feeds 2 x 256 fma: the vector formulations were bound by
// their loads (lanes = pairs, operator by scalar loads: 0.9-1.6 ms on config 2; lane = (pair, n) with two
// accumulators: 63 loads per 42 fma, 1.0 ms).  Layout (cdna_hip_programming.md): A[row = lane & 15][k = lane >> 4],
// B[k = lane >> 4][col = lane & 15], D[row = (lane >> 4) + 4 r][col = lane & 15], r = 0..3.
// A pair that does not take part in an offset (its own validity, or it lies outside the level) enters with zeros.
// Stores the local coefficients of levels >= 1, adds to those the level-0 pass of sr_farfield_kernel stored.
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int kM2LSlices = kM2LRow / 16; // column tiles: all in one wave (the A loads are the cost, shared by them)
template <bool COUNT>
__global__ __launch_bounds__(64) void sr_m2l_kernel(const int *__restrict__ zmax, FarParams fp,
                                                    unsigned long long *__restrict__ cnt) {
  int chunk = blockIdx.x, level = 0;
  for (;;) {
    const int c = (fp.box_count[level] * fp.n_layers + 15) >> 4;
    if (chunk < c || level + 1 >= fp.n_levels) break;
    chunk -= c;
    ++level;
  }
  const int lane = threadIdx.x, lo = lane & 15, kq = lane >> 4;
  const int n_pairs = fp.box_count[level] * fp.n_layers;
  // the pair of this lane's A row
  const int idx = chunk * 16 + lo;
  const bool live = idx < n_pairs;
  const int t = live ? idx / fp.n_layers : 0, layer = live ? idx - t * fp.n_layers : 0;
  const int zreq = min(zmax[layer], kHalf - 1) + 2, pms = fp.pm_src[layer];
  const int W = 64 << level;
  const bool has_parent = level + 1 < fp.n_levels;
  const int a_max = (kHalf + 1) / W - 1;
  // offsets whose parent pair can be invalid: close ones (parent closer than its first valid offset, taken over
  // the wave's layers) and the ones at the window ends
  int near_hi = a_max, win_lo = a_max + 1;
  if (has_parent) {
    const int W2 = 2 * W;
    int a_min_p = m2l_first_offset(level + 1, zreq, pms);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) a_min_p = max(a_min_p, __shfl_xor(a_min_p, m));
    near_hi = __builtin_amdgcn_readfirstlane(min(a_max, 2 * a_min_p + 1));
    win_lo = max(near_hi + 1, 2 * (kHalf / W2 - 1));
  }
  unsigned n_pairs_on = 0; // COUNT: translations of this lane's pair (counted once per pair: kq == 0)
  v4d acc_a[kM2LSlices], acc_e[kM2LSlices];
#pragma unroll
  for (int sl = 0; sl < kM2LSlices; ++sl) acc_a[sl] = acc_e[sl] = v4d{0., 0., 0., 0.};
  const int pad = kSrcPad >> level;
  for (int a = 3; a <= a_max; a = (a == near_hi ? win_lo : a + 1)) {
    for (int sg = 0; sg < 2; ++sg) {
      const int o = sg ? -a : a, s = t - o;
      const int sidx = s + pad;
      const bool on = live && sidx >= 0 && sidx < fp.n_src[level] && m2l_valid(level, o, zreq, pms) &&
                      !(has_parent && m2l_valid(level + 1, (t >> 1) - (s >> 1), zreq, pms));
      if (!__any(on)) continue;
      if (COUNT) n_pairs_on += on && kq == 0;
      const double *m = fp.mom + ((size_t)(fp.src_off[level] + (on ? sidx : 0)) * fp.n_layers + layer) * kMomPerBox + sg * (2 * kMQ);
      const double *T = fp.tab + ((size_t)sg * kM2LOffsets + a) * (kM2LQ * kM2LRow) + lo;
#pragma unroll
      for (int ks = 0; ks < kM2LQ / 4; ++ks) {
        const int q = 4 * ks + kq;
        const bool use = on && q < kMQ;
        const double A_a = use ? m[q] : 0.0;
        const double A_e = use ? m[kMQ + q] : 0.0;
#pragma unroll
        for (int sl = 0; sl < kM2LSlices; ++sl) {
          const double B = T[q * kM2LRow + 16 * sl];
          acc_a[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(A_a, B, acc_a[sl], 0, 0, 0);
          acc_e[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(A_e, B, acc_e[sl], 0, 0, 0);
        }
      }
    }
  }
  if (COUNT) count_add(cnt, kCntM2L, n_pairs_on, lane);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int io = chunk * 16 + kq + 4 * r; // the pair of D's row
    if (io >= n_pairs) continue;
    const int to = io / fp.n_layers, lyo = io - to * fp.n_layers;
    double *c = fp.coef + ((size_t)lyo * fp.n_boxes_total + fp.box_off[level] + to) * (2 * kFC);
#pragma unroll
    for (int sl = 0; sl < kM2LSlices; ++sl) {
      const int n = 16 * sl + lo; // this lane's column of D
      if (n < kFC) {
        if (level == 0) {
          c[n] += acc_a[sl][r];
          c[kFC + n] += acc_e[sl][r];
        } else {
          c[n] = acc_a[sl][r];
          c[kFC + n] = acc_e[sl][r];
        }
      }
    }
  }
}

// Exact near field + evaluation of the far-field polynomials.  The scalar unit is shared by
// the CU's four SIMDs, so ownership tests and interval arithmetic run on the VALU, 64 candidate
// lines at a time (lane = line); ballots mark the lines with work; those are walked in ROWS: the
// wave's 64 lanes as 8 rows of 8, a row takes one line (its record by per-lane loads) and steps
// through that line's points 8 at a time.
//
// sr_abscoeff_near_wings_kernel (one 64-point slot per wave): every REGION-1 point of the
//   lines that no far-field level owns for the slot, in rows; the slots holding a window end or
//   start by per-line expansions and a lane scan (window_end_sum) -- plus one far-field
//   polynomial per level; writes abs/emi (or adds, when the zones kernel stored first).
// sr_abscoeff_near_zones_kernel (one 512-point LDS image per wave; stores or adds to abs/emi):
//   region 3 (~15 points per line) with lanes = lines in the chunk phase; regions 2 and 4 in
//   rows, sums by return-less LDS adds (one wave per image, program order: deterministic).
// ------------------------------------------------------------------------
__device__ inline void near_ranges(const IcIndex &ix, int wlo, int width, int zm,
                                   int rs[3], int re[3]) {
  // candidates as ranges of the sorted centre list, C <= A <= B by start; NOMINAL tile end
  // (the far-field kernel tests nominal boxes, which may reach past g_hi):
  //  C: window END inside the tile      ic+6504 in [wlo, whn]
  //  A: within zm of the tile           ic in [wlo-zm, whn+zm]
  //  B: window START inside the tile    ic-6505 in (wlo, whn]
  const int whn = wlo + width - 1;
  const int c0 = lower_bound_ic(ix, wlo - (kHalf - 1));
  const int c1 = lower_bound_ic(ix, whn - (kHalf - 1) + 1);
  const int a0 = lower_bound_ic(ix, wlo - zm);
  const int a1 = lower_bound_ic(ix, whn + zm + 1);
  const int b0 = lower_bound_ic(ix, wlo + kHalf + 1);
  const int b1 = lower_bound_ic(ix, whn + kHalf + 1);
  rs[0] = c0; re[0] = max(c1, c0);
  rs[1] = max(a0, re[0]); re[1] = max(a1, rs[1]);
  rs[2] = max(b0, re[1]); re[2] = max(b1, rs[2]);
}

// ------------------------------------------------------------------------
// Window ends.  Every line has two, and in the slot holding one it contributes region-1 values to the
// points up to (from) its last (first) window point only -- nothing beyond, exactly as the
// reference's 13010-point window.  Walked line by line these partial slots were half of the near
// wings kernel (64 ends + 64 starts per slot at one line per point).  They are 6441+ points from
// the line centre, where the wing is so smooth over 64 points (Taylor ratio 32/6441) that degree 5
// is exact to 2e-14 of the term; so: lanes = lines, each lane expands its own line about the slot
// centre, a suffix (ends) / prefix (starts) sum over the lanes -- the lines are sorted by window
// position -- gives for every cut-off position the polynomial of all lines still (already) inside
// their window, and each point evaluates the one for its own position.
// ------------------------------------------------------------------------
// DPP row move of a double (two dwords): CTRL 0x101..0x10f row_shl:n (lane i reads lane i + n of its row of
// 16), 0x111..0x11f row_shr:n; lanes whose source lies outside the row get 0.
template <int CTRL>
__device__ inline double dpp_move(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ inline double lane_value(double v, int src_lane) { // wave-uniform: the value lane src_lane holds
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src_lane),
                          __builtin_amdgcn_readlane(__double2loint(v), src_lane));
}
constexpr int kWE = 6; // series coefficients of the window-end expansions
__device__ inline void wing_series6(double xc, double e, const FastRec &r, double f[kWE]) {
  const double u0 = xc * xc, u1 = 2. * xc * e, u2 = e * e;
  const R1Coef rq = r1_of(r);
  const double n0 = fma(rq.b, u0, rq.a), n1 = rq.b * u1, n2 = rq.b * u2;
  const double d0 = fma(u0, fma(4., u0, rq.d), rq.c);
  const double d1 = u1 * fma(8., u0, rq.d);
  const double d2 = fma(u2, rq.d, 4. * fma(u1, u1, 2. * u0 * u2));
  const double d3 = 8. * u1 * u2, d4 = 4. * u2 * u2;
  const double r0 = fast_rcp<2>(d0);
  const double D1 = d1 * r0, D2 = d2 * r0, D3 = d3 * r0, D4 = d4 * r0;
  f[0] = n0 * r0;
  f[1] = fma(-D1, f[0], n1 * r0);
  f[2] = fma(-D1, f[1], fma(-D2, f[0], n2 * r0));
  f[3] = -fma(D1, f[2], fma(D2, f[1], D3 * f[0]));
#pragma unroll
  for (int n = 4; n < kWE; ++n) f[n] = -fma(D1, f[n - 1], fma(D2, f[n - 2], fma(D3, f[n - 3], D4 * f[n - 4])));
}
// Sum over the lanes' lines of (value at this lane's point, 0 outside the line's window) for one chunk
// of 64 lines.  ends: line of lane i is valid at points p <= pos_i; starts: at p >= pos_i; pos is
// non-decreasing over the lanes (lines sorted by window centre); lanes without such a line carry
// zero coefficients.  c: [0, kWE) abs, [kWE, 2 kWE) emi series coefficients of the lane's line.
__device__ inline void window_end_sum(bool ends, int pos, double c[2 * kWE], int lane, double &out_a, double &out_e) {
  // inclusive suffix (ends) or prefix (starts) sums over the lanes: inside each row of 16 lanes by DPP
  // row shifts (plain VALU moves, zero shifted in at the row end), then the totals of the other rows
  const int row = lane >> 4;
#pragma unroll
  for (int n = 0; n < 2 * kWE; ++n) {
    double v = c[n];
    if (ends) {
      v += dpp_move<0x101>(v); // row_shl:1  lane i <- lane i + 1
      v += dpp_move<0x102>(v);
      v += dpp_move<0x104>(v);
      v += dpp_move<0x108>(v);
      const double t1 = lane_value(v, 16), t2 = lane_value(v, 32), t3 = lane_value(v, 48); // row totals
      const double t23 = t2 + t3;
      v += row == 0 ? t1 + t23 : (row == 1 ? t23 : (row == 2 ? t3 : 0.0));
    } else {
      v += dpp_move<0x111>(v); // row_shr:1  lane i <- lane i - 1
      v += dpp_move<0x112>(v);
      v += dpp_move<0x114>(v);
      v += dpp_move<0x118>(v);
      const double t0 = lane_value(v, 15), t1 = lane_value(v, 31), t2 = lane_value(v, 47);
      const double t01 = t0 + t1;
      v += row == 3 ? t01 + t2 : (row == 2 ? t01 : (row == 1 ? t0 : 0.0));
    }
    c[n] = v;
  }
  // this lane as a POINT p = lane: ends: first line with pos >= p; starts: last line with pos <= p
  int lo = 0, hi = 64; // 65 possible answers: 7 halvings
#pragma unroll
  for (int it = 0; it < 7; ++it) {
    const int mid = min((lo + hi) >> 1, 63);
    const int v = __shfl(pos, mid);
    const bool right = lo < hi && (ends ? v < lane : v <= lane);
    const bool left = lo < hi && !right;
    lo = right ? mid + 1 : lo;
    hi = left ? mid : hi;
  }
  const int src = ends ? lo : lo - 1; // lo = number of lines with pos < p (ends) / pos <= p (starts)
  const bool any = ends ? src < 64 : src >= 0;
  const double t = (double)(2 * lane - 63) * (1.0 / 64);
  double pa = 0., pe = 0.;
#pragma unroll
  for (int n = kWE - 1; n >= 0; --n) {
    const double ca = __shfl(c[n], src & 63), ce = __shfl(c[kWE + n], src & 63);
    pa = fma(pa, t, any ? ca : 0.0);
    pe = fma(pe, t, any ? ce : 0.0);
  }
  out_a += pa;
  out_e += pe;
}

// One 64-point slot per wave (4 or 2 slots per wave measured slower: 2.53 / 2.31 ms on config 2 at the time).
// Per chunk of 64 candidate lines (lane = line): which lines have region-1 points in the slot that no far-field
// level owns; window ends / starts by expansion and scan; the rest in ROWS: eight lines at a time, each row of 8
// lanes evaluates ONE line at 8 points per step (8 steps cover the slot), the line's record in the row's registers
// by per-lane loads.  Before, every such line cost a scalar record fetch, a ballot walk and four specialised masked
// loops: the walk took 0.87 of the kernel's 1.34 ms for 0.27 ms worth of vector work.
template <bool COUNT>
__global__ __launch_bounds__(64) void sr_abscoeff_near_wings_kernel(
    const FastRec *__restrict__ fast, IcIndex ix, const int *__restrict__ zmax, int n_sub,
    int n_tiles, int g_lo, int g_hi, FarParams fp, int add, const double *__restrict__ z_abs,
    const double *__restrict__ z_emi, double *__restrict__ abs_out, double *__restrict__ emi_out,
    unsigned long long *__restrict__ cnt) {
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int layer = wg / n_tiles, tile = wg - layer * n_tiles;
  const int wlo = g_lo + tile * 64;
  const int whi = min(wlo + 64, g_hi) - 1;
  const int lane = threadIdx.x;
  const int pm = fp.pm[layer];
  const int thr0 = ff_thr2(0, pm);
  int rs[3], re[3];
  near_ranges(ix, wlo, 64, min(max(zmax[layer], kTheta * 32 + pm + 1), kHalf - 1), rs, re);

  constexpr int kRowLanes = 8, kRows = 8;
  const int row = lane / kRowLanes, col = lane % kRowLanes;
  double sum_a = 0., sum_e = 0.; // this lane's point wlo + lane: window ends, polynomials, and the rows' total
  double ra[kRows], rb[kRows];    // rows: partial sums (abs, emi) of point wlo + col + 8 s over the lines of this lane's row
#pragma unroll
  for (int q = 0; q < kRows; ++q) ra[q] = rb[q] = 0.;
  const FastRec *frow = fast + (size_t)layer * n_sub;
  unsigned n_r1 = 0, n_we = 0; // COUNT: region-1 evaluations of this lane, window-end expansions
#ifdef SR_DIAG_WINGS
  unsigned n_diag = 0;
#endif
  for (int rg = 0; rg < 3; ++rg) {
    for (int base = rs[rg]; base < re[rg]; base += 64) {
      const int lv = base + lane;
      // lane = line: does the line have region-1 points in this slot that no far-field level owns, and are they
      // cut by the window end / start only (then it can join the expansion + scan below)
      bool need = false, fastl = false;
      bool has_l = false, has_r = false; // region-1 points of the lane's line in this slot: left wing (k < il), right wing (k > ir)
      int pos = 64; // lanes without a line: never selected by the scan
      if (lv < re[rg]) {
        const int j1 = frow[lv].j1;
        const unsigned ilir = frow[lv].ilir;
        const int il = (int)(ilir & 0xffffu), ir = (int)(ilir >> 16), jN = j1 + (kImxsig - 1);
        // cut-off position of the scan: every line of the range takes part in it (the positions are
        // non-decreasing over the lanes), lines handled elsewhere with zero coefficients
        pos = rg == 0 ? min(max(jN - wlo, -1), 64) : min(max(j1 - wlo, 0), 64);
        if (jN >= wlo && j1 <= whi && !ff_admissible(j1, il, ir, wlo, wlo + 63, thr0)) {
          has_l = max(wlo, j1) <= min(whi, j1 + il - 2); // points with 1 <= k < il
          has_r = max(wlo, j1 + ir) <= min(whi, jN);     // points with ir < k <= 13010
          need = has_l || has_r;
          if (rg != 1 && classify(j1, il, ir, wlo, whi) == 0)
            fastl = rg == 0 ? (has_r && !has_l && j1 + ir <= wlo) : (has_l && !has_r && j1 + il - 2 >= wlo + 63);
        }
      }
      if (__ballot(need) == 0) continue;
      // window ends (range 0) / starts (range 2) of this slot, when there are enough of them
      if (rg != 1) {
        const bool ends = rg == 0;
        const unsigned long long fm = __ballot(fastl);
        if (__builtin_popcountll(fm) >= 12) {
          double c[2 * kWE];
#pragma unroll
          for (int n = 0; n < 2 * kWE; ++n) c[n] = 0.;
          if (fastl) {
            const FastRec r = frow[lv];
            // -x (starts: left wing) or x (ends: right wing) at the slot centre wlo + 31.5
            const double xc = ends ? fma(0.5 * (double)(2 * (wlo - (r.j1 + r.ir() - 1)) + 63), r.xstep, r.xr)
                                   : fma(0.5 * (double)(2 * (wlo - r.j1) + 63), r.xstep, -r.xl);
            double f[kWE];
            wing_series6(xc, 32.0 * r.xstep, r, f);
            if (COUNT) ++n_we;
#pragma unroll
            for (int n = 0; n < kWE; ++n) {
              c[n] = r.wabs * f[n];
              c[kWE + n] = r.wemi * f[n];
            }
            need = false; // done here
          }
          window_end_sum(ends, pos, c, lane, sum_a, sum_e);
          if (__ballot(need) == 0) continue;
        }
      }
      // rows: region 1 is k < il (running x from k = 1: x = (k - 1) xstep - xl, i.e. -x) or k > ir
      // (x = (k - ir) xstep + xr), inside the window and the grid (lineshape.f:461-477, last writer wins).
      // A row takes one (line, wing) ITEM: the wing's points in the slot are one run of window indices
      // [k_a, k_b], so a step's lane mask is one add + one compare and x one fma from the row's x at step 0.
      // (Round 2 walked lines, with the wing chosen per point and step: 6 instructions of tests for each of the
      // 8 steps, executed or not, 17 per executed step, 27 to deal the lines to the rows and 16 unconditional
      // accumulations: 183 per round of which 80 evaluated something -- tools/diag_counts.py: 2.2e6 rounds,
      // 4.7 executed steps per round, 48 of 64 lanes on in an executed step on config 2.)
      unsigned long long m_l = __ballot(need && has_l), m_r = __ballot(need && has_r);
      while (m_l | m_r) {
        int code = -1; // this row's item: line (index into the chunk) | wing << 6; -1: none left
#pragma unroll
        for (int q = 0; q < kRows; ++q) {
          int c_ = -1; // wave-uniform
          if (m_l) {
            c_ = __builtin_ctzll(m_l);
            asm("s_bitset0_b64 %0, %1" : "+s"(m_l) : "s"(c_)); // m_l &= m_l - 1 in one scalar instruction instead of three
          } else if (m_r) {
            const int cr = __builtin_ctzll(m_r);
            asm("s_bitset0_b64 %0, %1" : "+s"(m_r) : "s"(cr));
            c_ = cr | 64;
          }
          code = row == q ? c_ : code;
        }
#ifdef SR_DIAG_WINGS // diagnostic builds only: rounds of the row walk (in the window-end counter)
        if (COUNT) n_we += 1; // summed over the 64 lanes
#endif
        const bool live = code >= 0, rw = live && (code & 64);
        const FastRec &r = frow[base + (live ? (code & 63) : 0)];
        const R1Coef rq = r1_of(r);
        const double xstep = r.xstep, a = rq.a, b = rq.b, c = rq.c, d = rq.d, wa = r.wabs, we = r.wemi;
        const int j1 = r.j1, il = r.il(), ir = r.ir();
        const int kb0 = wlo - j1 + 1 + col;              // window index of this lane's point at step 0
        const int k_last = min(kImxsig, whi - j1 + 1);  // last window index inside the slot, the grid and the window
        const int k_a = rw ? ir + 1 : 1, k_b = rw ? k_last : min(il - 1, k_last);
        const unsigned n_on = live ? (unsigned)max(k_b - k_a + 1, 0) : 0u;
        const int t0 = kb0 - k_a;                        // step q is on for this lane: (unsigned)(t0 + 8 q) < n_on
        const double x0 = fma((double)(kb0 - (rw ? ir : 1)), xstep, rw ? r.xr : -r.xl);
#pragma unroll
        for (int q = 0; q < kRows; ++q) {
          // a branch per step: steps no lane of the wave needs are skipped
          if ((unsigned)(t0 + kRowLanes * q) < n_on) {
            double x = x0;
            if (q > 0) asm("v_fma_f64 %0, %1, %2, %3" : "=v"(x) : "s"((double)(kRowLanes * q)), "v"(xstep), "v"(x0)); // one instruction
            const double x2 = x * x;
            const double y = fma(x2, b, a) * fast_rcp<1>(fma(x2, fma(x2, 4.0, d), c));
            if (COUNT) ++n_r1;
#ifdef SR_DIAG_WINGS // executed steps (in the polynomial counter)
            if (COUNT) n_diag |= 1u << q;
#endif
            ra[q] = fma(wa, y, ra[q]);
            rb[q] = fma(we, y, rb[q]);
          }
        }
#ifdef SR_DIAG_WINGS
        if (COUNT) {
          unsigned any_ = n_diag & 0xffu;
#pragma unroll
          for (int m = 32; m >= 1; m >>= 1) any_ |= (unsigned)__shfl_xor((int)any_, m);
          n_diag = (n_diag & ~0xffu) + ((unsigned)__builtin_popcount(any_) << 8);
        }
#endif
      }
    }
  }
  // totals of the rows: point wlo + lane is (col, step) = (lane % 8, lane / 8): sum over the eight rows of that step
#pragma unroll
  for (int q = 0; q < kRows; ++q) {
    double ta = ra[q], te = rb[q];
#pragma unroll
    for (int m = kRowLanes; m < 64; m <<= 1) {
      ta += __shfl_xor(ta, m);
      te += __shfl_xor(te, m);
    }
    if (row == q) {
      sum_a += ta;
      sum_e += te;
    }
  }
  if (COUNT) {
    count_add(cnt, kCntRegion1, n_r1, lane);
    count_add(cnt, kCntWindowEnds, n_we, lane);
#ifdef SR_DIAG_WINGS
    count_add(cnt, kCntPolyPoints, lane == 0 ? n_diag >> 8 : 0u, lane);
#else
    count_add(cnt, kCntPolyPoints, (wlo + lane <= whi) ? (unsigned)fp.n_levels : 0u, lane);
#endif
  }
  // far field: one polynomial per level
  const double *cl = fp.coef + (size_t)layer * fp.n_boxes_total * (2 * kFC);
  for (int lv = 0; lv < fp.n_levels; ++lv) {
    const int W = 64 << lv;
    const int b = (wlo - g_lo) >> (6 + lv);
    const int blo = g_lo + b * W;
    const double t = (double)(2 * (wlo + lane - blo) - (W - 1)) * (1.0 / 64 / (double)(1 << lv)); // exact: W = 2^(6+lv)
    const double *c = cl + (size_t)(fp.box_off[lv] + b) * (2 * kFC);
    double pa = c[kFC - 1], pe = c[2 * kFC - 1];
#pragma unroll
    for (int n = kFC - 2; n >= 0; --n) {
      pa = fma3s(pa, t, c[n]);
      pe = fma3s(pe, t, c[kFC + n]);
    }
    sum_a += pa;
    sum_e += pe;
  }
  const size_t orow = (size_t)layer * (size_t)(g_hi - g_lo);
  const int j = wlo + lane;
  if (j <= whi) {
    if (z_abs) { // the zones kernel's sums sit in a buffer of their own (it runs decoupled from the caller's stream)
      abs_out[orow + (j - g_lo)] = z_abs[orow + (j - g_lo)] + sum_a;
      emi_out[orow + (j - g_lo)] = z_emi[orow + (j - g_lo)] + sum_e;
    } else if (add) { // the zones kernel ran first (overlapped with the far-field kernel) and stored its sums
      abs_out[orow + (j - g_lo)] += sum_a;
      emi_out[orow + (j - g_lo)] += sum_e;
    } else {
      abs_out[orow + (j - g_lo)] = sum_a;
      emi_out[orow + (j - g_lo)] = sum_e;
    }
  }
}

// One region-3/4 point of a line: the point's window index and the line's data, per lane.  Several lines can
// hit the same grid point in one step, hence LDS atomics (one wave per image, program order: the sums stay
// deterministic).
// Largest of a value that is uniform within each row of 8 lanes, as a scalar: eight v_readlane + scalar max.  The
// row loops then run on a scalar step counter; "while any lane has a step left" costs three VALU instructions per
// step however it is written (the ballot's mask goes through v_cndmask + v_cmp_ne).
__device__ inline int rows_max(int v) {
  int m = __builtin_amdgcn_readlane(v, 0);
#pragma unroll
  for (int r = 8; r < 64; r += 8) m = max(m, __builtin_amdgcn_readlane(v, r));
  return m;
}
struct CorePend {
  int k, base; // window index; element of the image = k + base
  double gc, x0, dwp, inv_dwp, ryf, ryf2, two_ryf, wa, we;
};
// cos_tier: see cos_tiered (wave-uniform, from the largest 2 ry rx of the rows' runs)
__device__ inline void core_eval4(const CorePend &P, bool on, const GridParams &gp, int cos_tier, double p6_vgpr, double *s_a,
                                  double *s_e) {
  if (on) {
    const WinX xf{gp.lin_start, gp.lin_delta, P.gc};
    const double d = fabs(xf(P.k) - P.x0);
    double rx = d * P.inv_dwp; // |x(k)-x0|/dw correctly rounded: one residual correction
    rx = fma(fma(-P.dwp, rx, d), P.inv_dwp, rx);
    const double y = core_region4_m(P.ryf, P.ryf2, P.two_ryf, (double)(float)rx, cos_tier, p6_vgpr);
    const int idx = P.k + P.base;
    atomicAdd(&s_a[idx], P.wa * y);
    atomicAdd(&s_e[idx], P.we * y);
  }
}
// WT points per wave (a multiple of 64): the wider the image, the fewer zones are cut in two by
// its ends (a zone is ~280 points), i.e. the fewer partially filled lane runs.
#ifndef SR_ZONES_ROW
#define SR_ZONES_ROW 8 // lanes per row of the region-2 / region-4 walk: eight lines at a time (16: 4.35, 8: 4.1, 4: 4.87, 32: 5.09 ms)
#endif
#ifndef SR_RMAX
#define SR_RMAX 0 // 1: the rows' step counts from the lanes = lines phase (s_rmax) instead of rows_max() per round
#endif
#ifndef SR_R2_SPLIT
#define SR_R2_SPLIT 1 // region 2: the left and the right run of a line in two loops
#endif
#ifndef SR_R2_XRUN
#define SR_R2_XRUN 1 // region 2: x by a running sum (see the row walk): 5.38 vs 5.42 ms per headline step
#endif
#if SR_RMAX
#define SR_RMAX_OR(lds_value, row_value) __builtin_amdgcn_readfirstlane(lds_value)
#else
#define SR_RMAX_OR(lds_value, row_value) rows_max(row_value)
#endif
// At least four waves per SIMD (<= 128 VGPRs): the kernel sits at 118-127; with 129 (one more feature in the rows)
// the 8-wave blocks of a small shard went from two per CU to one and the shard's kernel from 0.50 to 0.63 ms.
#ifndef SR_ZONES_WAVES_PER_EU
#define SR_ZONES_WAVES_PER_EU 4
#endif
#define SR_ZONES_ATTR __attribute__((amdgpu_waves_per_eu(SR_ZONES_WAVES_PER_EU)))
template <int WT, int NW, bool COUNT>
__global__ __launch_bounds__(64 * NW) SR_ZONES_ATTR void sr_abscoeff_near_zones_kernel(
    const FastRec *__restrict__ fast, const ColdRec *__restrict__ cold, IcIndex ix,
    const int *__restrict__ zmax, int n_sub, int n_groups, int g_lo, int g_hi, GridParams gp, int add,
    double *__restrict__ abs_out, double *__restrict__ emi_out, unsigned long long *__restrict__ cnt, int layer0) {
  // one private image per wave: abs, emi.  Point p of the group (window index k of a line: p = k + j1 - 1 - wlo) lives at
  // element p + 1, i.e. element k + (j1 - wlo): the hot loops' address arithmetic carries no "- 1" (it cost a v_add per
  // region-2 point: the offset field of ds_add cannot be negative)
  __shared__ double s_img[NW][2][WT + 2];
#ifdef SR_ZONES_PAD
  // Tuning knob (see SR_FAR_WAVES_PER_EU): the wave's register footprint rounded up to a whole quarter of the SIMD's
  // file (118 -> 128 VGPRs; four waves per SIMD either way, LDS allows no fifth).
  asm volatile("" ::: "v127");
#endif
  // per wave and region (2, 4): the chunk's lines with work there, compacted in lane order -- lane id and the two run
  // words -- written by the lanes = lines phase, read by the rows (dealing the lines to the rows with ballots and
  // selects cost 27 VALU instructions per round of eight lines, two bpermutes fetched the run words)
  __shared__ int s_item[NW][2][3][64];
  // longest run of each round of eight list entries: region-2 left, region-2 right, region 4 (both sides) -- the rows'
  // step counts, wave-uniform (eight v_readlane + maxima per round when the rows worked them out themselves)
#if SR_RMAX
  __shared__ int s_rmax[NW][3][8];
#endif
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int layer = layer0 + wg / n_groups, grp = wg - (layer - layer0) * n_groups; // layer0: the launch's first layer (layer chunks)
  const int wlo = g_lo + grp * WT;
  const int whi = min(wlo + WT, g_hi) - 1;
  // NW waves work on the same group and take its 64-line chunks in turn, each into its own image
  // (added up in wave order at the end: the result does not depend on timing).  For small shards,
  // where one wave per 512-point image leaves wave slots empty and 256-point images cut more zones.
  const int lane = NW == 1 ? threadIdx.x : threadIdx.x & 63, wave = NW == 1 ? 0 : threadIdx.x >> 6;
  double *const s_a = s_img[wave][0], *const s_e = s_img[wave][1];
  const int zm = min(zmax[layer], kHalf - 1);
#pragma unroll
  for (int p = 0; p < WT / 64; ++p) s_a[1 + lane + 64 * p] = s_e[1 + lane + 64 * p] = 0.;
  unsigned n_r2 = 0, n_r3 = 0, n_r4 = 0; // COUNT: evaluations of this lane per region
  const double p6_vgpr = vgpr_constant(SR_F32(.56419)); // see core_region4_m
  // lines whose zone [ic - zm, ic + zm] can meet the group
  const int l0 = lower_bound_ic(ix, wlo - zm), l1 = lower_bound_ic(ix, whi + zm + 1);
  const FastRec *frow = fast + (size_t)layer * n_sub;
  const ColdRec *crow = cold + (size_t)layer * n_sub;
  // Serpentine sweep: even groups take their 64-line chunks left to right, odd groups right to left.  Neighbouring
  // groups (consecutive work ids, same XCD, started together) share the lines within the zone width of their common
  // edge -- 45 % more record bytes than the tables hold; swept in opposite directions both reach the shared lines at
  // the same time (start or end of their sweep) and the second one finds them in L2 instead of HBM.
  const int n_chunks = (l1 - l0 + 63) >> 6;
  for (int ci = wave; ci < n_chunks; ci += NW) {
    const int base = l0 + 64 * ((grp & 1) ? n_chunks - 1 - ci : ci);
    const int lv = min(base + lane, l1 - 1);
    bool act;
    // runs of the lane's line inside this group, start | count << 16 (window indices <= 13010):
    // region 2 left / right (lineshape.f:503-522), region 4 left / right of the region-3 interval
    unsigned run2l, run2r, run4l, run4r;
    int n_items2, n_items4; // lines of this chunk with region-2 / region-4 work (wave-uniform)
    {
      // ---- lane = line: which lines have work here, the interval arithmetic of every line (the walk
      // below reads it back with v_readlane: done there it was ~70 scalar instructions per line), and
      // REGION 3 (lineshape.f:554-560), the ~15 points around the centre: all lanes step through
      // their own interval together (the intervals of neighbouring lines are equally long), the
      // line's data sit in the lane's registers and die before the walk; sums by ds_add_f64.  Walked
      // line by line with lanes = points these few points cost a whole 64-lane chunk per line (1.1
      // of 6.6 ms).
      const FastRec &r = frow[lv];
      const int j1 = r.j1, il = r.il(), ir = r.ir();
      const int zl = max(j1 + il - 1, j1), zh = min(j1 + ir - 1, j1 + (kImxsig - 1));
      act = base + lane < l1 && zl <= whi && zh >= wlo; // zone (inside the window) meets the group
      if (__ballot(act) == 0) continue;
      const ColdRec &z = crow[lv];
      const int k3lo = z.k3lo(), k3hi = z.k3hi();
      const int il2 = z.il2(), ir2 = z.ir2();
      const int k_lo = max(wlo - j1 + 1, 1), k_hi = min(whi - j1 + 1, kImxsig); // group & window, as k
      {
        const int a0 = max(il, k_lo), a1 = il < il2 ? min(il2, k_hi) : a0 - 1;
        const int b0 = max(ir2, k_lo), b1 = ir2 < ir ? min(ir, k_hi) : b0 - 1;
        run2l = (unsigned)a0 | ((unsigned)max(a1 - a0 + 1, 0) << 16);
        run2r = (unsigned)b0 | ((unsigned)max(b1 - b0 + 1, 0) << 16);
      }
      // the core (il2a, ir2a), lineshape.f:524-562: region 3 = [k3lo, k3hi], region 4 on both sides
      const int c_lo = max(((il2 == il) ? il - 1 : il2) + 1, k_lo), c_hi = min(((ir2 == ir) ? ir + 1 : ir2) - 1, k_hi);
      const bool has3 = k3lo <= k3hi;
      {
        const int a0 = c_lo, a1 = has3 ? min(k3lo - 1, c_hi) : c_hi;
        const int b0 = has3 ? max(k3hi + 1, c_lo) : c_hi + 1, b1 = c_hi;
        run4l = (unsigned)a0 | ((unsigned)max(a1 - a0 + 1, 0) << 16);
        run4r = (unsigned)b0 | ((unsigned)max(b1 - b0 + 1, 0) << 16);
      }
      const int e0 = max(k3lo, c_lo), e1 = min(k3hi, c_hi);
      const int n3 = (act && has3) ? max(e1 - e0 + 1, 0) : 0;
      const bool w2 = act && ((run2l >> 16) + (run2r >> 16)) > 0, w4 = act && ((run4l >> 16) + (run4r >> 16)) > 0;
      if (__any(n3 > 0 || w4)) {
        const WinX xf{gp.lin_start, gp.lin_delta, grid_at(gp, j1 + kHalf)};
        const double x0 = z.x0, dwp = z.dwp, inv_dwp = cold_inv_dwp(z.dwp), ryf = cold_ryf(z.ry), wa = r.wabs, we = r.wemi;
        const int ibase = j1 - wlo; // element = point + 1
        {
          // the cosine's argument range over the line's region-4 runs (cos_tiered), bits 30-31 of run4l (a count is
          // < 2^14): |Im c1| = 2 ry rx is largest at the outer ends of the two runs
          const double rx_max = fmax(fabs(xf((int)(run4l & 0xffffu)) - x0),
                                     fabs(xf((int)(run4r & 0xffffu) + (int)(run4r >> 16) - 1) - x0)) * inv_dwp * 1.001;
          const double ui_max = (ryf + ryf) * rx_max;
          run4l |= (ui_max < 6.5e-3 ? 2u : (ui_max < 0.78 ? 1u : 0u)) << 30;
        }
        for (int t = 0; __any(t < n3); ++t) {
          if (t < n3) {
            const int k = e0 + t;
            const double d = fabs(xf(k) - x0);
            double rx = d * inv_dwp; // |x(k)-x0|/dw correctly rounded: one residual correction
            rx = fma(fma(-dwp, rx, d), inv_dwp, rx);
            const double y = core_region3(ryf, (double)(float)(-rx));
            if (COUNT) ++n_r3;
            atomicAdd(&s_a[k + ibase], wa * y);
            atomicAdd(&s_e[k + ibase], we * y);
          }
        }
      }
      // the lines with region-2 / region-4 work, compacted in lane order
      const unsigned long long m2 = __ballot(w2), m4 = __ballot(w4);
      n_items2 = __builtin_popcountll(m2);
      n_items4 = __builtin_popcountll(m4);
#if SR_RMAX
      if (lane < 24) (&s_rmax[wave][0][0])[lane] = 0;
#endif
      if (w2) {
        const int at = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m2, 0u));
        s_item[wave][0][0][at] = lane;
        s_item[wave][0][1][at] = (int)run2l;
        s_item[wave][0][2][at] = (int)run2r;
#if SR_RMAX && SR_R2_SPLIT
        atomicMax(&s_rmax[wave][0][at >> 3], (int)(run2l >> 16));
        atomicMax(&s_rmax[wave][1][at >> 3], (int)(run2r >> 16));
#elif SR_RMAX
        atomicMax(&s_rmax[wave][0][at >> 3], (int)(run2l >> 16) + (int)(run2r >> 16));
#endif
      }
      if (w4) {
        const int at = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m4 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m4, 0u));
        s_item[wave][1][0][at] = lane;
        s_item[wave][1][1][at] = (int)run4l;
        s_item[wave][1][2][at] = (int)run4r;
#if SR_RMAX
        atomicMax(&s_rmax[wave][2][at >> 3], (int)((run4l >> 16) & 0x3fffu) + (int)(run4r >> 16));
#endif
      }
    }
    // ---- regions 2 and 4, kRows lines at a time: each row of kRowLanes lanes walks the points of ONE line,
    // kRowLanes per step, with the line's data in its own registers (per-lane loads of the record fields: the four
    // rows read four records).  History: one line at a time with 64 points per step left the last chunk of each
    // run mostly empty (a line has ~150 region-2 and ~72 region-4 points: 64 + 64 + 22 lanes, 71-78 % lane use)
    // and cost a scalar record fetch + ~40 scalar / vector instructions per line; packing region-4 points of
    // consecutive lines into full chunks fixed its lane use but not the per-line walk.  Rows: lane use 90-94 %, no
    // scalar walk, no pending state: sr_abscoeff_near_zones_kernel 4.9 -> 4.2 ms on config 2.
    {
      constexpr int kRowLanes = SR_ZONES_ROW, kRows = 64 / kRowLanes;
      const int row = lane / kRowLanes, col = lane % kRowLanes;
      // region 2 (lineshape.f:503-522): item t of a line: left k = a0 + t, x = xs2l - (k - il) xstep; right
      // k = b0 + (t - na), x = xs2r + (k - ir2) xstep.  Only x^2 enters, so both are t xstep + c with a per-line c.
      for (int g = 0; g < n_items2; g += kRows) {
        const int it = min(g + row, n_items2 - 1);
        const bool live = g + row < n_items2;
        const int ls_ = s_item[wave][0][0][it];
        const FastRec &r = frow[base + ls_];
        const ColdRec &z = crow[base + ls_];
        const unsigned ul = (unsigned)s_item[wave][0][1][it], ur = (unsigned)s_item[wave][0][2][it];
        const int a0 = (int)(ul & 0xffffu), na = (int)(ul >> 16), b0 = (int)(ur & 0xffffu), nb = (int)(ur >> 16);
        const double xstep = r.xstep, wa = r.wabs, we = r.wemi;
        double q2[8];
        region2_coef_fma(z.ry, q2); // rebuilt per line (ColdRec): 25 flops against ~20 steps of 21 instructions
        const int base_idx = r.j1 - wlo; // element = point + 1
        // the two runs one after the other, t counted from each run's first point (one loop over both with the
        // side chosen per point cost a compare and three selects per point: 25 -> 21 VALU instructions per point
        // for ~5 % more steps)
        auto run = [&](const int n, const int n_max, const double c, const int i0) {
          const int n_steps = (n_max + kRowLanes - 1) / kRowLanes; // wave-uniform: a scalar loop counter
#if SR_R2_XRUN
          // x of the lane's point by a running sum, eight steps of xstep at a time (one add instead of a conversion
          // and an fma per point; <= 20 steps of a run: <= 20 ulp of x, 1e-14 of the value at most)
          double xt = fma((double)col, xstep, c);
          const double xs8 = (double)kRowLanes * xstep;
#endif
#if SR_R2_XRUN
          // ... and ONE counter per lane: the byte offset of its point in the image, which is the loop's predicate too
          int ab = (col + i0) * 8;
          const int ab_end = (n + i0) * 8;
          for (int st = 0; st < n_steps; ++st, ab += 8 * kRowLanes) {
            if (ab < ab_end) {
              const double y = region2_val(q2, xt);
              xt += xs8;
              if (COUNT) ++n_r2;
              atomicAdd(reinterpret_cast<double *>(reinterpret_cast<char *>(s_a) + ab), wa * y); // return-less LDS adds: no wait for a read, runs of different lines overlap
              atomicAdd(reinterpret_cast<double *>(reinterpret_cast<char *>(s_e) + ab), we * y);
            }
          }
#else
          for (int st = 0; st < n_steps; ++st) {
            const int t = col + kRowLanes * st;
            if (t < n) {
              const double y = region2_val(q2, fma((double)t, xstep, c));
              if (COUNT) ++n_r2;
              atomicAdd(&s_a[t + i0], wa * y); // return-less LDS adds: no wait for a read, runs of different lines overlap
              atomicAdd(&s_e[t + i0], we * y);
            }
          }
#endif
        };
#if SR_R2_SPLIT
        run(live ? na : 0, SR_RMAX_OR(s_rmax[wave][0][g >> 3], live ? na : 0),
            fma((double)(a0 - r.il()), xstep, -z.xs2l), a0 + base_idx);
        run(live ? nb : 0, SR_RMAX_OR(s_rmax[wave][1][g >> 3], live ? nb : 0),
            fma((double)(b0 - z.ir2()), xstep, z.xs2r), b0 + base_idx);
#else
        {
          const int n = live ? na + nb : 0;
          const double c_left = fma((double)(a0 - r.il()), xstep, -z.xs2l);
          const double c_right = fma((double)(b0 - na - z.ir2()), xstep, z.xs2r);
          int i_left = a0 + base_idx, i_right = b0 - na + base_idx;
          asm volatile("" : "+v"(i_left), "+v"(i_right)); // keep the two sums: re-associated, they cost a v_add per point
          const int n_steps = (SR_RMAX_OR(s_rmax[wave][0][g >> 3], n) + kRowLanes - 1) / kRowLanes;
          for (int st = 0; st < n_steps; ++st) {
            const int t = col + kRowLanes * st;
            if (t < n) {
              const bool lf = t < na;
              const double y = region2_val(q2, fma((double)t, xstep, lf ? c_left : c_right));
              if (COUNT) ++n_r2;
              const int idx = t + (lf ? i_left : i_right);
              atomicAdd(&s_a[idx], wa * y);
              atomicAdd(&s_e[idx], we * y);
            }
          }
        }
#endif
      }
      // region 4 (lineshape.f:530-546), on both sides of the region-3 interval
      for (int g = 0; g < n_items4; g += kRows) {
        const int it = min(g + row, n_items4 - 1);
        const bool live = g + row < n_items4;
        const int ls_ = s_item[wave][1][0][it];
        const FastRec &r = frow[base + ls_];
        const ColdRec &z = crow[base + ls_];
        const unsigned ul = (unsigned)s_item[wave][1][1][it], ur = (unsigned)s_item[wave][1][2][it];
        const int a0 = (int)(ul & 0xffffu), na = (int)((ul >> 16) & 0x3fffu), b0 = (int)(ur & 0xffffu), nb = (int)(ur >> 16);
        const int n = live ? na + nb : 0;
        const int j1 = r.j1;
        CorePend P;
        P.base = j1 - wlo; // element = point + 1
        P.gc = grid_at(gp, j1 + kHalf);
        P.x0 = z.x0;
        P.dwp = z.dwp;
        P.inv_dwp = cold_inv_dwp(z.dwp);
        P.ryf = cold_ryf(z.ry);
        P.ryf2 = P.ryf * P.ryf;
        P.two_ryf = P.ryf + P.ryf;
        P.wa = r.wabs;
        P.we = r.wemi;
        // the cosine's tier of this round: the smallest of its lines' (from the lanes = lines phase)
        const int tier_ = live ? (int)(ul >> 30) : 2;
        const int cos_tier = __all(tier_ == 2) ? 2 : (__all(tier_ >= 1) ? 1 : 0);
        const int n_steps = (SR_RMAX_OR(s_rmax[wave][2][g >> 3], n) + kRowLanes - 1) / kRowLanes;
        for (int st = 0; st < n_steps; ++st) {
          const int t = col + kRowLanes * st;
          P.k = t < na ? a0 + t : b0 + (t - na);
          if (COUNT) n_r4 += t < n;
          core_eval4(P, t < n, gp, cos_tier, p6_vgpr, s_a, s_e);
        }
      }
    }
  }
  if (COUNT) {
    count_add(cnt, kCntRegion2, n_r2, lane);
    count_add(cnt, kCntRegion3, n_r3, lane);
    count_add(cnt, kCntRegion4, n_r4, lane);
  }
  if (NW > 1) __syncthreads();
  const size_t row = (size_t)layer * (size_t)(g_hi - g_lo);
  for (int p = threadIdx.x; p < WT; p += 64 * NW) {
    const int j = wlo + p;
    if (j <= whi) {
      double ta = s_img[0][0][p + 1], te = s_img[0][1][p + 1];
#pragma unroll
      for (int w = 1; w < NW; ++w) {
        ta += s_img[w][0][p + 1];
        te += s_img[w][1][p + 1];
      }
      if (add) {
        abs_out[row + (j - g_lo)] += ta;
        emi_out[row + (j - g_lo)] += te;
      } else {
        abs_out[row + (j - g_lo)] = ta;
        emi_out[row + (j - g_lo)] = te;
      }
    }
  }
}


// ------------------------------------------------------------------------
// The multi-channel pass (level pair tables, G-coefficient tables): every line ONCE, each of its three weighted
// contributions to the spectrum of the level it belongs to.
//
// The per-level route evaluates every line twice (in its upper level's pass and in its lower level's) and pays the
// fixed cost of a (slot, layer) wave twelve times -- a 9 %-of-the-lines pass cost 1.08 ms where its lines' share of a
// full pass is 0.48 (3.1-3.4 coefficient-op equivalents for twelve levels).  Here the near-field
// kernels walk the FULL sorted list once, exactly as the folded op does, and the accumulation goes to an LDS image
// with one plane per output spectrum: wabs -> plane (lev_lo, o_lo), wemi -> (lev_up, o_up_e), w3 -> (lev_up, o_up_a)
// (McChannels).  An image of n_ch x 256 points is shared by the NW waves of a workgroup (return-less ds_add_f64 from
// all of them: the sums' order of addition depends on timing at the 1e-16 level, unlike the one-wave images of the folded op).
// The far field stays per level (its translations and polynomials are linear per output spectrum anyway): far-only
// passes of the level sub-linesets leave their coefficients for sr_wings_mc_kernel's polynomial stage.
// ------------------------------------------------------------------------
// ------------------------------------------------------------------------
// Local-to-local translation (the downward pass of the hierarchy).  A box's local expansion IS a polynomial of degree
// kFD over the box, so its restriction to a child box is the same polynomial re-centred: with t_parent = s/2 + t_child/2
// (s = -1 left, +1 right child)  child_n += sum_{m >= n} 2^-m C(m, n) s^(m-n) parent_m  -- exact algebra, no truncation,
// operator entries exact in fp64 (an integer below 2^53 times a power of two).  After it the level-0 coefficients hold
// every level's far field and a point evaluates ONE polynomial per output instead of one per hierarchy level.  For the
// multi-channel pass (n_far x 2 outputs per point: 5 x 12 x 2 Horner chains of 22 fma per point were 2.3 of
// sr_wings_mc_kernel's 5.2 ms); the folded op keeps its five polynomials (176 fma of a ~4100-instruction wave).
// One wave per (layer, widest box) of a far pass: the box's tree of 31 coefficient sets in LDS, levels top down.
// ------------------------------------------------------------------------
__global__ __launch_bounds__(64) void sr_l2l_kernel(double *__restrict__ coef_one, const FarBatchItem *__restrict__ items, FarParams fp,
                                                    const double *__restrict__ tab) {
  double *const coef = items ? items[blockIdx.z].coef : coef_one; // one pass, or item blockIdx.z of a batch
  constexpr int kBlk = 2 * kFC, kTop = kMaxFarLevels - 1;
  constexpr int kTree = (2 << kTop) - 1; // boxes of the tree under one widest box: 1 + 2 + ... + 2^kTop
  __shared__ double s_c[kTree * kBlk];
  const int layer = blockIdx.y, top = blockIdx.x, lane = threadIdx.x;
  double *cl = coef + (size_t)layer * fp.n_boxes_total * kBlk;
  const int lq = min(lane, kBlk - 1), ch = lq / kFC, n = lq - ch * kFC; // lanes 0 .. 2 kFC - 1: (output, coefficient)
  // this lane's two operator rows in registers (zeros left of the diagonal: the inner loop needs no bounds); in LDS the
  // 8.5 KB table held a CU to eight of these waves (first build: 2.2 ms for 0.3 ms of work)
  double T0[kFC], T1[kFC];
#pragma unroll
  for (int m = 0; m < kFC; ++m) {
    T0[m] = tab[(size_t)n * kFC + m];
    T1[m] = tab[((size_t)kFC + n) * kFC + m];
  }
  // tree slot of box j (0-based within this top box) of level lv: the levels stored widest first
  auto slot = [&](int lv, int j) { return ((1 << (kTop - lv)) - 1) + j; };
  for (int lv = kTop; lv >= 0; --lv) {
    const int nb = 1 << (kTop - lv);
    for (int j = 0; j < nb; ++j) {
      const int b = top * nb + j;
      if (lane < kBlk) s_c[slot(lv, j) * kBlk + lane] = b < fp.box_count[lv] ? cl[(size_t)(fp.box_off[lv] + b) * kBlk + lane] : 0.0;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int lv = kTop; lv >= 1; --lv) {
    const int nb = 1 << (kTop - lv);
    for (int j = 0; j < nb; ++j) {
      const double *par = s_c + slot(lv, j) * kBlk + ch * kFC;
      double *c0 = s_c + slot(lv - 1, 2 * j) * kBlk + lq, *c1 = c0 + kBlk;
      double a0 = *c0, a1 = *c1;
#pragma unroll
      for (int m = 0; m < kFC; ++m) {
        const double pm = par[m];
        a0 = fma(T0[m], pm, a0);
        a1 = fma(T1[m], pm, a1);
      }
      if (lane < kBlk) {
        *c0 = a0;
        *c1 = a1;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  const int nb0 = 1 << kTop;
  for (int j = 0; j < nb0; ++j) {
    const int b = top * nb0 + j;
    if (lane < kBlk && b < fp.box_count[0]) cl[(size_t)(fp.box_off[0] + b) * kBlk + lane] = s_c[slot(0, j) * kBlk + lane];
  }
}
void l2l_table_host(double *tab) { // [2: left, right child][n][m]
  for (int sg = 0; sg < 2; ++sg)
    for (int n = 0; n < kFC; ++n)
      for (int m = 0; m < kFC; ++m) {
        long double v = 0.0L;
        if (m >= n) {
          v = 1.0L;
          for (int k = 1; k <= n; ++k) v = v * (long double)(m - n + k) / (long double)k; // C(m, n)
          for (int k = 0; k < m; ++k) v *= 0.5L;
          if (sg == 0 && ((m - n) & 1)) v = -v;
        }
        tab[((size_t)sg * kFC + n) * kFC + m] = (double)v;
      }
}
int launch_l2l(double *coef, int n_layers, const FarParams &fp, const double *tab, hipStream_t st) {
  if (n_layers <= 0) return 0;
  hipLaunchKernelGGL(sr_l2l_kernel, dim3((unsigned)fp.box_count[kMaxFarLevels - 1], (unsigned)n_layers), dim3(64), 0, st, coef,
                     (const FarBatchItem *)nullptr, fp, tab);
  return (int)hipGetLastError();
}
// tables, per-line expansions at every level and the downward pass of n_items sparse far-only passes, three launches
int launch_far_batch(const FarBatchItem *items, int n_items, int max_n_sub, const LayersDev &A, const GridParams &gp, const int *zmax,
                     int g_lo, const FarParams &fp, const double *l2l_tab, hipStream_t st) {
  if (n_items <= 0 || A.n_layers <= 0 || max_n_sub <= 0) return 0;
  hipLaunchKernelGGL(sr_prep_batch_kernel, dim3((unsigned)((max_n_sub + kPrepBlock - 1) / kPrepBlock), (unsigned)A.n_layers, (unsigned)n_items),
                     dim3(kPrepBlock), 0, st, items, A, gp);
  const unsigned gx = (unsigned)(fp.n_boxes_total * ((A.n_layers + kFarRows - 1) / kFarRows));
  hipLaunchKernelGGL(sr_farfield_rows_batch_kernel, dim3((gx + kRowsBatchWaves - 1) / kRowsBatchWaves, (unsigned)n_items), dim3(64 * kRowsBatchWaves), 0,
                     st, items, zmax, g_lo, fp, (int)gx);
  hipLaunchKernelGGL(sr_l2l_kernel, dim3((unsigned)fp.box_count[kMaxFarLevels - 1], (unsigned)A.n_layers, (unsigned)n_items), dim3(64), 0, st,
                     (double *)nullptr, items, fp, l2l_tab);
  return (int)hipGetLastError();
}

#if SR_FASTREC64
__device__ inline int mc_base(const McChannels &mc, int level, int off, int plane) { return (mc.stride * level + off) * plane; }

// The two stages of a 64-point slot's near wings (sr_wings_mc_kernel):
// img = plane 0 at the slot's first point, `plane` doubles between planes; the slot's n_waves waves (this one: `wave`) take
// its candidate chunks and far passes in turn.
__device__ __forceinline__ void wings_mc_rows(const FastRec *__restrict__ frow, const int *__restrict__ lev_up,
                                              const int *__restrict__ lev_lo, const IcIndex &ix, const int zm_near, const int wlo,
                                              const int whi, const int thr0, const McChannels &mc, double *const img,
                                              const int plane, const int lane, const int wave, const int n_waves) {
  int rs[3], re[3];
  near_ranges(ix, wlo, 64, zm_near, rs, re);
  constexpr int kRowLanes = 8, kRows = 8;
  const int row = lane / kRowLanes, col = lane % kRowLanes;
  int turn = 0; // chunks of all three ranges dealt to the waves in turn
  for (int rg = 0; rg < 3; ++rg) {
    for (int base = rs[rg]; base < re[rg]; base += 64) {
      if ((turn++) % n_waves != wave) continue;
      const int lv = base + lane;
      bool has_l = false, has_r = false; // region-1 points of the lane's line in this slot that no far-field level owns
      if (lv < re[rg]) {
        const int j1 = frow[lv].j1;
        const unsigned ilir = frow[lv].ilir;
        const int il = (int)(ilir & 0xffffu), ir = (int)(ilir >> 16), jN = j1 + (kImxsig - 1);
        if (jN >= wlo && j1 <= whi && !ff_admissible(j1, il, ir, wlo, wlo + 63, thr0)) {
          has_l = max(wlo, j1) <= min(whi, j1 + il - 2); // points with 1 <= k < il
          has_r = max(wlo, j1 + ir) <= min(whi, jN);     // points with ir < k <= 13010
        }
      }
      unsigned long long m_l = __ballot(has_l), m_r = __ballot(has_r);
      while (m_l | m_r) {
        int code = -1; // this row's item: line (index into the chunk) | wing << 6; -1: none left
#pragma unroll
        for (int q = 0; q < kRows; ++q) {
          int c_ = -1; // wave-uniform
          if (m_l) {
            c_ = __builtin_ctzll(m_l);
            asm("s_bitset0_b64 %0, %1" : "+s"(m_l) : "s"(c_));
          } else if (m_r) {
            const int cr = __builtin_ctzll(m_r);
            asm("s_bitset0_b64 %0, %1" : "+s"(m_r) : "s"(cr));
            c_ = cr | 64;
          }
          code = row == q ? c_ : code;
        }
        const bool live = code >= 0, rw = live && (code & 64);
        const int li = base + (live ? (code & 63) : 0);
        const FastRec &r = frow[li];
        const R1Coef rq = r1_of(r);
        const double xstep = r.xstep, a = rq.a, b = rq.b, c = rq.c, d = rq.d, wa = r.wabs, we = r.wemi, w3 = r.w3;
        const int j1 = r.j1, il = r.il(), ir = r.ir();
        const int lu = lev_up[li], ll = lev_lo[li];
        double *const p_lo = img + (mc.stride * ll + mc.o_lo) * plane + col, *const p_ue = img + (mc.stride * lu + mc.o_up_e) * plane + col,
                     *const p_ua = img + (mc.stride * lu + mc.o_up_a) * plane + col;
        const int kb0 = wlo - j1 + 1 + col;              // window index of this lane's point at step 0
        const int k_last = min(kImxsig, whi - j1 + 1);  // last window index inside the slot, the grid and the window
        const int k_a = rw ? ir + 1 : 1, k_b = rw ? k_last : min(il - 1, k_last);
        const unsigned n_on = live ? (unsigned)max(k_b - k_a + 1, 0) : 0u;
        const int t0 = kb0 - k_a;                        // step q is on for this lane: (unsigned)(t0 + 8 q) < n_on
        const double x0 = fma((double)(kb0 - (rw ? ir : 1)), xstep, rw ? r.xr : -r.xl);
        // (The rows take the slot's eight 8-point groups in lockstep -- row r and row r' add to the same eight doubles of a
        // plane when their lines share a level, as most do with the ground state.  Taking them in ROTATED order, row r at
        // group (q + r) mod 8, removes that same-address conflict: built -- a table build 13.02 (lockstep) vs 13.17 ms (rotated), three
        // alternating runs: the three more integer instructions per step cost more than the conflicts.)
#pragma unroll
        for (int q = 0; q < kRows; ++q) {
          if ((unsigned)(t0 + kRowLanes * q) < n_on) {
            double x = x0;
            if (q > 0) asm("v_fma_f64 %0, %1, %2, %3" : "=v"(x) : "s"((double)(kRowLanes * q)), "v"(xstep), "v"(x0));
            const double x2 = x * x;
            const double y = fma(x2, b, a) * fast_rcp<1>(fma(x2, fma(x2, 4.0, d), c));
            atomicAdd(&p_lo[kRowLanes * q], wa * y); // return-less LDS adds: rows of different lines may hit the same point
            atomicAdd(&p_ue[kRowLanes * q], we * y);
            atomicAdd(&p_ua[kRowLanes * q], w3 * y);
          }
        }
      }
    }
  }
  // far field: one polynomial per (far pass, hierarchy level), added to the pass's two planes at this lane's point.
  // The coefficients of a pass -- kMaxFarLevels x 2 kFC doubles, one block per hierarchy level -- come in by four
  // coalesced vector loads (lane = coefficient), go through an LDS staging row and are read back as broadcasts; the next
  // pass's loads are in flight while this one is evaluated.  (Round 6, first build: wave-uniform scalar loads, one
  // (pass, level) block of 92 SGPRs at a time, each waited for before its 44 fma -- 60 dependent round trips per wave
  // and most of the kernel's 4.9 ms.)
}
__device__ __forceinline__ void wings_mc_polys(const McFarPass *__restrict__ far, const int n_far, const FarParams &fp, const int layer,
                                               const int g_lo, const int wlo, double *const img, const int plane,
                                               double *const s_cf, const int lane, const int wave, const int n_waves) {
  {
    constexpr int kEvalLevels = 1; // sr_l2l_kernel has folded the wider levels into level 0
    constexpr int kBlk = 2 * kFC, kTot = kEvalLevels * kBlk;
    constexpr int kLoads = (kTot + 63) / 64; // loads per lane that cover a pass
    // s_cf: [64 kLoads] staging row of this wave
    double tt[kMaxFarLevels];
    int cidx[kLoads]; // element of the pass's coefficient table this lane fetches in load j (-1: none)
#pragma unroll
    for (int lv = 0; lv < kMaxFarLevels; ++lv) {
      const int W = 64 << lv;
      const int blo = g_lo + ((wlo - g_lo) >> (6 + lv)) * W;
      tt[lv] = (double)(2 * (wlo + lane - blo) - (W - 1)) * (1.0 / 64 / (double)(1 << lv)); // exact: W = 2^(6+lv)
    }
#pragma unroll
    for (int q = 0; q < kLoads; ++q) {
      const int e = lane + 64 * q, lv = e / kBlk;
      cidx[q] = e < kTot ? (fp.box_off[min(lv, kMaxFarLevels - 1)] + ((wlo - g_lo) >> (6 + lv))) * kBlk + (e - lv * kBlk) : -1;
    }
    const size_t lrow = (size_t)layer * fp.n_boxes_total * kBlk;
    auto next_pass = [&](int f) { // first pass >= f of this wave (passes f = wave, wave + kMcWingWaves, ...) with coefficients
      while (f < n_far && !far[f].coef) f += n_waves;
      return f;
    };
    double reg[kLoads];
    auto fetch = [&](int f) {
      const double *cl = far[f].coef + lrow;
#pragma unroll
      for (int q = 0; q < kLoads; ++q) reg[q] = cidx[q] >= 0 ? cl[cidx[q]] : 0.0;
    };
    int f = next_pass(wave);
    if (f < n_far) fetch(f);
    while (f < n_far) {
      __builtin_amdgcn_wave_barrier(); // (the previous pass's broadcast reads are done: LDS is in order per wave)
#pragma unroll
      for (int q = 0; q < kLoads; ++q) s_cf[lane + 64 * q] = reg[q];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int ch_a = far[f].ch_a, ch_e = far[f].ch_e;
      const int fn = next_pass(f + n_waves);
      if (fn < n_far) fetch(fn);
      // one Horner chain (kFC broadcast reads, then kFC - 1 fma) at a time: unrolled over levels and outputs the
      // compiler hoisted all 230 reads of a pass (256 VGPRs, or 171 spills at 128)
      double sa = 0., se = 0.;
#pragma unroll 1
      for (int h = 0; h < 2 * kEvalLevels; ++h) {
        const int lv = h >> 1;
        const double *c = s_cf + h * kFC; // block lv: [abs kFC | emi kFC]
        double t = tt[0];
#pragma unroll
        for (int q = 1; q < kMaxFarLevels; ++q) t = lv == q ? tt[q] : t;
        double cv[kFC];
#pragma unroll
        for (int n = 0; n < kFC; ++n) cv[n] = c[n];
        double pv = cv[kFC - 1];
#pragma unroll
        for (int n = kFC - 2; n >= 0; --n) pv = fma(pv, t, cv[n]);
        if (h & 1) se += pv; else sa += pv;
      }
      if (ch_a >= 0) img[ch_a * plane + lane] += sa;
      if (ch_e >= 0) img[ch_e * plane + lane] += se;
      f = fn;
    }
  }
}

// (Round 6 also built this kernel FUSED with the near wings and the polynomials -- wings_mc_rows / wings_mc_polys on the
// same image, two waves per 64-point slot, the tables stored once instead of stored, read and stored again: 13.5 vs 13.1
// ms per build of the pair tables, 24.0 vs 21.5 for the three ctypes.  The wings stages ran at this kernel's 16 waves per
// CU instead of their own 24, behind barriers, and the zones part no longer ran beside the far passes.  Removed.)
template <int WT, int NW>
__global__ __launch_bounds__(64 * NW) SR_ZONES_ATTR void sr_zones_mc_kernel(
    const FastRec *__restrict__ fast, const ColdRec *__restrict__ cold, const int *__restrict__ lev_up,
    const int *__restrict__ lev_lo, IcIndex ix, const int *__restrict__ zmax, int n_sub, int n_groups, int g_lo, int g_hi,
    GridParams gp, McChannels mc, double *__restrict__ out, int n_rows_total, int row0) {
  extern __shared__ double s_dyn[];
  constexpr int kPlane = WT + 2;             // point p of the group lives at element p + 1 of its plane (see the folded kernel)
  double *const s_img = s_dyn;               // [n_ch][kPlane]
  int *const s_item_all = reinterpret_cast<int *>(s_dyn + (size_t)mc.n_ch * kPlane); // [NW][2][4][64]
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int layer = wg / n_groups, grp = wg - layer * n_groups;
  const int wlo = g_lo + grp * WT;
  const int whi = min(wlo + WT, g_hi) - 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int *const s_item = s_item_all + wave * (2 * 4 * 64);
  auto item = [&](int region, int field, int at) -> int & { return s_item[(region * 4 + field) * 64 + at]; };
  const int zm = min(zmax[layer], kHalf - 1);
  for (int e = threadIdx.x; e < mc.n_ch * kPlane; e += 64 * NW) s_img[e] = 0.;
  __syncthreads();
  const double p6_vgpr = vgpr_constant(SR_F32(.56419)); // see core_region4_m
  const int l0 = lower_bound_ic(ix, wlo - zm), l1 = lower_bound_ic(ix, whi + zm + 1);
  const FastRec *frow = fast + (size_t)layer * n_sub;
  const ColdRec *crow = cold + (size_t)layer * n_sub;
  const int n_chunks = (l1 - l0 + 63) >> 6;
  for (int ci = wave; ci < n_chunks; ci += NW) {
    const int base = l0 + 64 * ((grp & 1) ? n_chunks - 1 - ci : ci); // serpentine sweep, see the folded kernel
    const int lv = min(base + lane, l1 - 1);
    bool act;
    unsigned run2l, run2r, run4l, run4r;
    int n_items2, n_items4;
    {
      // ---- lane = line: interval arithmetic of every line, region 3 (see sr_abscoeff_near_zones_kernel)
      const FastRec &r = frow[lv];
      const int j1 = r.j1, il = r.il(), ir = r.ir();
      const int zl = max(j1 + il - 1, j1), zh = min(j1 + ir - 1, j1 + (kImxsig - 1));
      act = base + lane < l1 && zl <= whi && zh >= wlo;
      if (__ballot(act) == 0) continue;
      const ColdRec &z = crow[lv];
      const int k3lo = z.k3lo(), k3hi = z.k3hi();
      const int il2 = z.il2(), ir2 = z.ir2();
      const int k_lo = max(wlo - j1 + 1, 1), k_hi = min(whi - j1 + 1, kImxsig);
      {
        const int a0 = max(il, k_lo), a1 = il < il2 ? min(il2, k_hi) : a0 - 1;
        const int b0 = max(ir2, k_lo), b1 = ir2 < ir ? min(ir, k_hi) : b0 - 1;
        run2l = (unsigned)a0 | ((unsigned)max(a1 - a0 + 1, 0) << 16);
        run2r = (unsigned)b0 | ((unsigned)max(b1 - b0 + 1, 0) << 16);
      }
      const int c_lo = max(((il2 == il) ? il - 1 : il2) + 1, k_lo), c_hi = min(((ir2 == ir) ? ir + 1 : ir2) - 1, k_hi);
      const bool has3 = k3lo <= k3hi;
      {
        const int a0 = c_lo, a1 = has3 ? min(k3lo - 1, c_hi) : c_hi;
        const int b0 = has3 ? max(k3hi + 1, c_lo) : c_hi + 1, b1 = c_hi;
        run4l = (unsigned)a0 | ((unsigned)max(a1 - a0 + 1, 0) << 16);
        run4r = (unsigned)b0 | ((unsigned)max(b1 - b0 + 1, 0) << 16);
      }
      const int e0 = max(k3lo, c_lo), e1 = min(k3hi, c_hi);
      const int n3 = (act && has3) ? max(e1 - e0 + 1, 0) : 0;
      const bool w2 = act && ((run2l >> 16) + (run2r >> 16)) > 0, w4 = act && ((run4l >> 16) + (run4r >> 16)) > 0;
      // the three planes of the lane's line, as element offsets, packed for the rows: 3 x 10 bits of channel index
      const int lu = lev_up[lv], ll = lev_lo[lv];
      const int ch_lo = mc.stride * ll + mc.o_lo, ch_ue = mc.stride * lu + mc.o_up_e, ch_ua = mc.stride * lu + mc.o_up_a;
      const int chans = ch_lo | (ch_ue << 10) | (ch_ua << 20);
      if (__any(n3 > 0 || w4)) {
        const WinX xf{gp.lin_start, gp.lin_delta, grid_at(gp, j1 + kHalf)};
        const double x0 = z.x0, dwp = z.dwp, inv_dwp = cold_inv_dwp(z.dwp), ryf = cold_ryf(z.ry);
        const double wa = r.wabs, we = r.wemi, w3 = r.w3;
        const int ibase = j1 - wlo; // element = point + 1
        {
          const double rx_max = fmax(fabs(xf((int)(run4l & 0xffffu)) - x0),
                                     fabs(xf((int)(run4r & 0xffffu) + (int)(run4r >> 16) - 1) - x0)) * inv_dwp * 1.001;
          const double ui_max = (ryf + ryf) * rx_max;
          run4l |= (ui_max < 6.5e-3 ? 2u : (ui_max < 0.78 ? 1u : 0u)) << 30;
        }
        double *const p_lo = s_img + ch_lo * kPlane + ibase, *const p_ue = s_img + ch_ue * kPlane + ibase,
                     *const p_ua = s_img + ch_ua * kPlane + ibase;
        for (int t = 0; __any(t < n3); ++t) {
          if (t < n3) {
            const int k = e0 + t;
            const double d = fabs(xf(k) - x0);
            double rx = d * inv_dwp;
            rx = fma(fma(-dwp, rx, d), inv_dwp, rx);
            const double y = core_region3(ryf, (double)(float)(-rx));
            atomicAdd(&p_lo[k], wa * y);
            atomicAdd(&p_ue[k], we * y);
            atomicAdd(&p_ua[k], w3 * y);
          }
        }
      }
      const unsigned long long m2 = __ballot(w2), m4 = __ballot(w4);
      n_items2 = __builtin_popcountll(m2);
      n_items4 = __builtin_popcountll(m4);
      if (w2) {
        const int at = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m2, 0u));
        item(0, 0, at) = lane;
        item(0, 1, at) = (int)run2l;
        item(0, 2, at) = (int)run2r;
        item(0, 3, at) = chans;
      }
      if (w4) {
        const int at = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m4 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m4, 0u));
        item(1, 0, at) = lane;
        item(1, 1, at) = (int)run4l;
        item(1, 2, at) = (int)run4r;
        item(1, 3, at) = chans;
      }
    }
    {
      constexpr int kRowLanes = SR_ZONES_ROW, kRows = 64 / kRowLanes;
      const int row = lane / kRowLanes, col = lane % kRowLanes;
      // region 2 (lineshape.f:503-522), eight lines at a time (see the folded kernel)
      for (int g = 0; g < n_items2; g += kRows) {
        const int it = min(g + row, n_items2 - 1);
        const bool live = g + row < n_items2;
        const int ls_ = item(0, 0, it);
        const FastRec &r = frow[base + ls_];
        const ColdRec &z = crow[base + ls_];
        const unsigned ul = (unsigned)item(0, 1, it), ur = (unsigned)item(0, 2, it);
        const int chans = item(0, 3, it);
        const int a0 = (int)(ul & 0xffffu), na = (int)(ul >> 16), b0 = (int)(ur & 0xffffu), nb = (int)(ur >> 16);
        const double xstep = r.xstep, wa = r.wabs, we = r.wemi, w3 = r.w3;
        double q2[8];
        region2_coef_fma(z.ry, q2);
        const int base_idx = r.j1 - wlo; // element = point + 1
        // byte offsets of the line's second and third plane relative to its first
        const int d_ue = (((chans >> 10) & 0x3ff) - (chans & 0x3ff)) * (kPlane * 8), d_ua = ((chans >> 20) - (chans & 0x3ff)) * (kPlane * 8);
        char *const plane0 = reinterpret_cast<char *>(s_img + (chans & 0x3ff) * kPlane);
        auto run = [&](const int n, const int n_max, const double c, const int i0) {
          const int n_steps = (n_max + kRowLanes - 1) / kRowLanes;
          double xt = fma((double)col, xstep, c);
          const double xs8 = (double)kRowLanes * xstep;
          int ab = (col + i0) * 8;
          const int ab_end = (n + i0) * 8;
          for (int st = 0; st < n_steps; ++st, ab += 8 * kRowLanes) {
            if (ab < ab_end) {
              const double y = region2_val(q2, xt);
              xt += xs8;
              atomicAdd(reinterpret_cast<double *>(plane0 + ab), wa * y);
              atomicAdd(reinterpret_cast<double *>(plane0 + ab + d_ue), we * y);
              atomicAdd(reinterpret_cast<double *>(plane0 + ab + d_ua), w3 * y);
            }
          }
        };
        run(live ? na : 0, rows_max(live ? na : 0), fma((double)(a0 - r.il()), xstep, -z.xs2l), a0 + base_idx);
        run(live ? nb : 0, rows_max(live ? nb : 0), fma((double)(b0 - z.ir2()), xstep, z.xs2r), b0 + base_idx);
      }
      // region 4 (lineshape.f:530-546), on both sides of the region-3 interval
      for (int g = 0; g < n_items4; g += kRows) {
        const int it = min(g + row, n_items4 - 1);
        const bool live = g + row < n_items4;
        const int ls_ = item(1, 0, it);
        const FastRec &r = frow[base + ls_];
        const ColdRec &z = crow[base + ls_];
        const unsigned ul = (unsigned)item(1, 1, it), ur = (unsigned)item(1, 2, it);
        const int chans = item(1, 3, it);
        const int a0 = (int)(ul & 0xffffu), na = (int)((ul >> 16) & 0x3fffu), b0 = (int)(ur & 0xffffu), nb = (int)(ur >> 16);
        const int n = live ? na + nb : 0;
        const int j1 = r.j1;
        const int ebase = j1 - wlo; // element = point + 1
        const double gc = grid_at(gp, j1 + kHalf), x0 = z.x0, dwp = z.dwp, inv_dwp = cold_inv_dwp(z.dwp);
        const double ryf = cold_ryf(z.ry), ryf2 = ryf * ryf, two_ryf = ryf + ryf;
        const double wa = r.wabs, we = r.wemi, w3 = r.w3;
        double *const p_lo = s_img + (chans & 0x3ff) * kPlane + ebase, *const p_ue = s_img + ((chans >> 10) & 0x3ff) * kPlane + ebase,
                     *const p_ua = s_img + (chans >> 20) * kPlane + ebase;
        const int tier_ = live ? (int)(ul >> 30) : 2;
        const int cos_tier = __all(tier_ == 2) ? 2 : (__all(tier_ >= 1) ? 1 : 0);
        const int n_steps = (rows_max(n) + kRowLanes - 1) / kRowLanes;
        const WinX xf{gp.lin_start, gp.lin_delta, gc};
        for (int st = 0; st < n_steps; ++st) {
          const int t = col + kRowLanes * st;
          const int k = t < na ? a0 + t : b0 + (t - na);
          if (t < n) {
            const double d = fabs(xf(k) - x0);
            double rx = d * inv_dwp; // |x(k)-x0|/dw correctly rounded: one residual correction
            rx = fma(fma(-dwp, rx, d), inv_dwp, rx);
            const double y = core_region4_m(ryf, ryf2, two_ryf, (double)(float)rx, cos_tier, p6_vgpr);
            atomicAdd(&p_lo[k], wa * y);
            atomicAdd(&p_ue[k], we * y);
            atomicAdd(&p_ua[k], w3 * y);
          }
        }
      }
    }
  }
  __syncthreads();
  // the image's planes to their spectra: out [n_ch][n_rows_total][n_pts], stores (sr_wings_mc_kernel adds)
  const size_t n_pts = (size_t)(g_hi - g_lo);
  const int n_here = whi - wlo + 1;
  for (int c = wave; c < mc.n_ch; c += NW) {
    double *o = out + ((size_t)c * n_rows_total + (size_t)(row0 + layer)) * n_pts + (size_t)(wlo - g_lo);
    for (int p = lane; p < n_here; p += 64) o[p] = s_img[c * kPlane + p + 1];
  }
}

// wave-uniform loads of far-field coefficients through the constant address space: scalar loads whatever LDS traffic
// precedes them (profiles/r05_wings_experiments.md: behind any LDS store the compiler turned these loads, through a flat
// pointer, into 24 vector loads per level)
typedef const double __attribute__((address_space(4))) *kconst_ptr;
__device__ inline kconst_ptr as_kconst(const double *p) { return (kconst_ptr)(uintptr_t)p; }

// sr_wings_mc_kernel (one 64-point slot per wave, as sr_abscoeff_near_wings_kernel): every region-1 point of the lines
// that no far-field level owns for the slot -- window ends and starts included: the per-channel suffix sums of the
// folded kernel's window-end scan would be n_ch scans -- in rows, summed into the wave's LDS image [n_ch][64]; then one
// far-field polynomial per (far pass, hierarchy level) from the coefficients the level sub-linesets' far-only passes
// left; ADDS to out (sr_zones_mc_kernel stored).
__global__ __launch_bounds__(64 * kMcWingWaves) void sr_wings_mc_kernel(
    const FastRec *__restrict__ fast, const int *__restrict__ lev_up, const int *__restrict__ lev_lo, IcIndex ix,
    const int *__restrict__ zmax, int n_sub, int n_tiles, int g_lo, int g_hi, FarParams fp, McChannels mc,
    const McFarPass *__restrict__ far, int n_far, double *__restrict__ out, int n_rows_total, int row0) {
  extern __shared__ double s_dyn[];
  double *const s_img = s_dyn; // [n_ch][64]
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int layer = wg / n_tiles, tile = wg - layer * n_tiles;
  const int wlo = g_lo + tile * 64;
  const int whi = min(wlo + 64, g_hi) - 1;
  // kMcWingWaves waves share the slot's image and take its candidate chunks, far passes and output planes in turn: one
  // wave per slot with its 12 KB image kept a CU to 11 of these latency-bound waves (24 of the folded kernel's fit)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pm = fp.pm[layer];
  for (int c = wave; c < mc.n_ch; c += kMcWingWaves) s_img[c * 64 + lane] = 0.;
  __syncthreads();
  wings_mc_rows(fast + (size_t)layer * n_sub, lev_up, lev_lo, ix, min(max(zmax[layer], kTheta * 32 + pm + 1), kHalf - 1), wlo, whi,
                ff_thr2(0, pm), mc, s_img, 64, lane, wave, kMcWingWaves);
  __syncthreads(); // the rows' sums of every wave are in the image
  wings_mc_polys(far, n_far, fp, layer, g_lo, wlo, s_img, 64, s_img + mc.n_ch * 64 + wave * 64, lane, wave, kMcWingWaves);
  __syncthreads(); // (a pass's two planes are its wave's alone, but the output planes below are dealt by index)
  const int j = wlo + lane;
  if (j <= whi) {
    const size_t n_pts = (size_t)(g_hi - g_lo);
    double *o = out + (size_t)(row0 + layer) * n_pts + (size_t)(j - g_lo);
    const size_t cstride = (size_t)n_rows_total * n_pts;
    for (int c = wave; c < mc.n_ch; c += kMcWingWaves) o[c * cstride] += s_img[c * 64 + lane];
  }
}

// Image width by what fits: two workgroups of kMcImage points per CU when their planes allow (24 planes x 258 doubles =
// 49.5 KB), else 128-point images (36 planes -- the three ctypes of twelve levels -- are 74 KB at 256 points: ONE workgroup,
// eight waves per CU; at 128 points 37 KB: two).  Up to 146 planes (64 levels' pair tables).
size_t zones_mc_lds(int n_ch, int wt) { return sizeof(double) * (size_t)n_ch * (wt + 2) + sizeof(int) * (size_t)kMcWaves * 2 * 4 * 64; }
int zones_mc_image(int n_ch) {
  if (2 * zones_mc_lds(n_ch, kMcImage) <= (size_t)160 * 1024) return kMcImage;
  return zones_mc_lds(n_ch, 128) <= (size_t)160 * 1024 ? 128 : 0;
}
template <int WT>
static int launch_zones_mc_wt(const FastRec *fast, const ColdRec *cold, const int *lev_up, const int *lev_lo, const IcIndex &ix,
                              const int *zmax, int n_sub, int n_layers, int g_lo, int g_hi, const GridParams &gp, const McChannels &mc,
                              double *out, int n_rows_total, int row0, hipStream_t st) {
  constexpr int NW = kMcWaves;
  const int n_t = (g_hi - g_lo + WT - 1) / WT;
  // (more than 64 KB of dynamic LDS needs the attribute once per kernel and DEVICE: a process that moves to another
  // device sets it there too)
  static unsigned long long attr_set = 0;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!(attr_set >> (dev & 63) & 1ull)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sr_zones_mc_kernel<WT, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set |= 1ull << (dev & 63);
  }
  hipLaunchKernelGGL((sr_zones_mc_kernel<WT, NW>), dim3((unsigned)(n_t * n_layers)), dim3(64 * NW), zones_mc_lds(mc.n_ch, WT), st, fast, cold,
                     lev_up, lev_lo, ix, zmax, n_sub, n_t, g_lo, g_hi, gp, mc, out, n_rows_total, row0);
  return (int)hipGetLastError();
}
int launch_zones_mc(const FastRec *fast, const ColdRec *cold, const int *lev_up, const int *lev_lo, const IcIndex &ix,
                    const int *zmax, int n_sub, int n_layers, int g_lo, int g_hi, const GridParams &gp, const McChannels &mc,
                    double *out, int n_rows_total, int row0, hipStream_t st) {
  if (g_hi <= g_lo || n_layers <= 0) return 0;
  const int wt = zones_mc_image(mc.n_ch);
  if (wt == kMcImage)
    return launch_zones_mc_wt<kMcImage>(fast, cold, lev_up, lev_lo, ix, zmax, n_sub, n_layers, g_lo, g_hi, gp, mc, out, n_rows_total, row0, st);
  if (wt == 128)
    return launch_zones_mc_wt<128>(fast, cold, lev_up, lev_lo, ix, zmax, n_sub, n_layers, g_lo, g_hi, gp, mc, out, n_rows_total, row0, st);
  return (int)hipErrorInvalidValue;
}

size_t wings_mc_lds(int n_ch) { return sizeof(double) * ((size_t)n_ch * 64 + 64 * kMcWingWaves); }
int launch_wings_mc(const FastRec *fast, const int *lev_up, const int *lev_lo, const IcIndex &ix, const int *zmax, int n_sub,
                    int n_layers, int g_lo, int g_hi, const FarParams &fp, const McChannels &mc, const McFarPass *far, int n_far,
                    double *out, int n_rows_total, int row0, hipStream_t st) {
  if (g_hi <= g_lo || n_layers <= 0) return 0;
  const int n_g1 = (g_hi - g_lo + 63) / 64;
  const size_t lds = wings_mc_lds(mc.n_ch); // the image + the polynomial stage's staging rows
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  static unsigned long long attr_set = 0; // (per kernel and device, see launch_zones_mc_wt)
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!(attr_set >> (dev & 63) & 1ull)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sr_wings_mc_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set |= 1ull << (dev & 63);
  }
  hipLaunchKernelGGL(sr_wings_mc_kernel, dim3((unsigned)(n_g1 * n_layers)), dim3(64 * kMcWingWaves), lds, st, fast, lev_up, lev_lo, ix, zmax, n_sub,
                     n_g1, g_lo, g_hi, fp, mc, far, n_far, out, n_rows_total, row0);
  return (int)hipGetLastError();
}

#else // 80-byte records carry no third weight: mc_pass never takes this route then (SR_ERR_UNSUPPORTED)
int zones_mc_image(int) { return 0; }
size_t wings_mc_lds(int) { return 0; }
int launch_zones_mc(const FastRec *, const ColdRec *, const int *, const int *, const IcIndex &, const int *, int, int, int, int,
                    const GridParams &, const McChannels &, double *, int, int, hipStream_t) { return (int)hipErrorNotSupported; }
int launch_wings_mc(const FastRec *, const int *, const int *, const IcIndex &, const int *, int, int, int, int, const FarParams &,
                    const McChannels &, const McFarPass *, int, double *, int, int, hipStream_t) { return (int)hipErrorNotSupported; }
#endif

// a += za, e += ze (small shards: the zones kernel's private result joins the wings kernel's)
__global__ __launch_bounds__(256) void sr_add2_kernel(double *__restrict__ a, const double *__restrict__ za,
                                                      double *__restrict__ e, const double *__restrict__ ze, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    a[i] += za[i];
    e[i] += ze[i];
  }
}
int launch_add2(double *a, const double *za, double *e, const double *ze, size_t n, hipStream_t st) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(sr_add2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, za, e, ze, n);
  return (int)hipGetLastError();
}

int launch_farfield(const FastRec *fast, const IcIndex &ix, const int *zmax, int n_sub, int n_layers, int g_lo,
                    int g_hi, const FarParams &fp, unsigned long long *cnt, hipStream_t st) {
  if (g_hi <= g_lo || n_layers <= 0) return 0;
  if (!fp.m2l && fp.rows) { // sparse line set: a box for kFarRows layers per wave
    const dim3 gr((unsigned)(fp.n_boxes_total * ((n_layers + kFarRows - 1) / kFarRows)));
    if (cnt)
      hipLaunchKernelGGL(sr_farfield_rows_kernel<true>, gr, dim3(64), 0, st, fast, ix, zmax, n_sub, g_lo, fp, cnt);
    else
      hipLaunchKernelGGL(sr_farfield_rows_kernel<false>, gr, dim3(64), 0, st, fast, ix, zmax, n_sub, g_lo, fp, cnt);
    return (int)hipGetLastError();
  }
  const dim3 grid((unsigned)((fp.m2l ? fp.box_count[0] : fp.n_boxes_total) * n_layers));
#define SR_FAR(C, M) hipLaunchKernelGGL((sr_farfield_kernel<C, M>), grid, dim3(64), 0, st, fast, ix, zmax, n_sub, g_lo, g_hi, fp, cnt)
  if (fp.m2l) {
    if (cnt) SR_FAR(true, true); else SR_FAR(false, true);
  } else {
    if (cnt) SR_FAR(true, false); else SR_FAR(false, false);
  }
#undef SR_FAR
  return (int)hipGetLastError();
}

int launch_m2l(const FastRec *fast, const IcIndex &ix, const int *zmax, int n_sub, int n_layers, int g_lo, int g_hi,
               const FarParams &fp, unsigned long long *cnt, hipStream_t st, int which) {
  if (g_hi <= g_lo || n_layers <= 0) return 0;
  if (which & 1) {
  const dim3 g1((unsigned)(fp.n_src[0] * n_layers));
  if (cnt)
    hipLaunchKernelGGL(sr_s2m_kernel<true>, g1, dim3(64), 0, st, fast, ix, n_sub, g_lo, fp, cnt);
  else
    hipLaunchKernelGGL(sr_s2m_kernel<false>, g1, dim3(64), 0, st, fast, ix, n_sub, g_lo, fp, cnt);
  for (int l = 1; l < fp.n_levels; ++l) { // upward pass, level by level
    const int n2 = fp.n_src[l] * n_layers * 4;
    hipLaunchKernelGGL(sr_m2m_kernel, dim3((unsigned)((n2 + 63) / 64)), dim3(64), 0, st, fp, l);
  }
  }
  if (!(which & 2)) return (int)hipGetLastError();
  int chunks = 0;
  for (int lv = 0; lv < fp.n_levels; ++lv) chunks += (fp.box_count[lv] * n_layers + 15) / 16;
  if (cnt)
    hipLaunchKernelGGL(sr_m2l_kernel<true>, dim3((unsigned)chunks), dim3(64), 0, st, zmax, fp, cnt);
  else
    hipLaunchKernelGGL(sr_m2l_kernel<false>, dim3((unsigned)chunks), dim3(64), 0, st, zmax, fp, cnt);
  return (int)hipGetLastError();
}

// Translation operator of the box-pair far field, [o > 0 | o < 0][|o|][q - 2 < kM2LQ][n < kM2LRow]:
// (-1)^n (q + n - 1)!/n! r^(q+n), r = 1/(2o) (see sr_m2l_kernel); long double, rounded once.
void m2l_table_host(double *tab) {
  for (size_t i = 0; i < (size_t)2 * kM2LOffsets * kM2LQ * kM2LRow; ++i) tab[i] = 0.0; // padding rows / columns
  for (int sg = 0; sg < 2; ++sg)
    for (int a = 0; a < kM2LOffsets; ++a)
      for (int q = 2; q <= kFD; ++q) {
        double *row = tab + (((size_t)sg * kM2LOffsets + a) * kM2LQ + (q - 2)) * kM2LRow;
        if (a == 0) continue;
        const long double r = (sg ? -1.0L : 1.0L) / (2.0L * a);
        long double v = 1.0L;
        for (int k = 2; k < q; ++k) v *= k;        // (q-1)!
        for (int k = 0; k < q; ++k) v *= r;        // r^q
        row[0] = (double)v;
        for (int n = 1; n < kFC; ++n) {
          v *= -r * (long double)(q + n - 1) / (long double)n;
          row[n] = (double)v;
        }
      }
}

#ifndef SR_ZONES_WT
#define SR_ZONES_WT 512
#endif
constexpr int kZoneImage = SR_ZONES_WT; // grid points per LDS image of sr_abscoeff_near_zones_kernel (a multiple of 64)
template <int NW, bool COUNT>
static void launch_zones(dim3 gz, const FastRec *fast, const ColdRec *cold, const IcIndex &ix, const int *zmax,
                         int n_sub, int n_t, int g_lo, int g_hi, const GridParams &gp, int add, double *abs_out,
                         double *emi_out, unsigned long long *cnt, hipStream_t st) {
  // (layer chunks, one launch each, were measured and lost: profiles/r04_timeline_zones_chunks8.txt)
  hipLaunchKernelGGL((sr_abscoeff_near_zones_kernel<kZoneImage, NW, COUNT>), gz, dim3(64 * NW), 0, st, fast, cold, ix, zmax, n_sub,
                     n_t, g_lo, g_hi, gp, add, abs_out, emi_out, cnt, 0);
}

int launch_near(int part, int add, const FastRec *fast, const ColdRec *cold, const IcIndex &ix, const int *zmax,
                int n_sub, int n_layers, int g_lo, int g_hi, const GridParams &gp, const FarParams &fp,
                double *abs_out, double *emi_out, unsigned long long *cnt, hipStream_t st, const double *z_abs,
                const double *z_emi) {
  if (g_hi <= g_lo || n_layers <= 0) return 0;
  const int n_groups = (g_hi - g_lo + kGroup - 1) / kGroup;
  const dim3 grid((unsigned)(n_groups * n_layers));
  if (part == 1) {
    // slots per wave: 4 (256-point groups) measured 2.53 ms on config 2, 2: 2.31 ms, 1: see DESIGN.md
    const int n_g1 = (g_hi - g_lo + 63) / 64;
    if (cnt)
      hipLaunchKernelGGL(sr_abscoeff_near_wings_kernel<true>, dim3((unsigned)(n_g1 * n_layers)), dim3(64), 0, st,
                         fast, ix, zmax, n_sub, n_g1, g_lo, g_hi, fp, add, z_abs, z_emi, abs_out, emi_out, cnt);
    else
      hipLaunchKernelGGL(sr_abscoeff_near_wings_kernel<false>, dim3((unsigned)(n_g1 * n_layers)), dim3(64), 0, st,
                         fast, ix, zmax, n_sub, n_g1, g_lo, g_hi, fp, add, z_abs, z_emi, abs_out, emi_out, cnt);
  } else {
    // Image width: wider images cut fewer zones in two (fewer (line, group) pairs: 7.1 -> 6.7 ms on
    // 1e5 points x 80 layers with 512 instead of 256) as long as the waves still fill the chip
    // several times over (1024: 9.3 ms, too few waves and 16 KB LDS each).
    const long waves512 = (long)((g_hi - g_lo + kZoneImage - 1) / kZoneImage) * n_layers;
    const int n_t = (g_hi - g_lo + kZoneImage - 1) / kZoneImage;
    const dim3 gz((unsigned)(n_t * n_layers));
#define SR_ZONES(NW)                                                                                         \
  (cnt ? launch_zones<NW, true>(gz, fast, cold, ix, zmax, n_sub, n_t, g_lo, g_hi, gp, add, abs_out, emi_out, cnt, st) \
       : launch_zones<NW, false>(gz, fast, cold, ix, zmax, n_sub, n_t, g_lo, g_hi, gp, add, abs_out, emi_out, cnt, st))
    if (waves512 >= 3 * 4096)
      SR_ZONES(1);
    else if (waves512 >= 3 * 2048)
      SR_ZONES(2);
    else if (waves512 >= 3 * 512)
      // Judged by the PIPELINED step (zones beside the far-field and wings kernels), not by the kernel alone: on
      // config 2 (round 3) 1/8 shard, 2000 images: NW = 1 / 2 / 4 / 8 -> 1.065 / 0.928 / 0.870 / 0.923 ms per step
      // (alone: 0.765 / 0.550 / 0.520 / 0.518) -- round 2 took 8 below 3072 images; 1/4 shard, 3920 images:
      // 1.554 / 1.588 / 1.589 / 1.602; 1/2 shard: 2.848 / 2.881 / 2.841 / 2.900; whole grid, 15680: 5.49 / 5.52
      // (alone 3.53 / 3.34: two waves per image have the shorter tail, but beside the other kernels the tail is
      // filled anyway).  One wave from 3072 images up looked as good on config 2 but left a sparse 20000-point
      // shard (0.3 lines per point, 3300 images) at 2.8 instead of 1.5 ms: it is the WAVES that must fill the chip.
      // 5, 6 and 7 waves (14 chunks per image = 2 x 7) measured 0.68 / 0.81 / 0.59 ms alone on the 1/8 shard.
      SR_ZONES(4);
    else
      SR_ZONES(8);
#undef SR_ZONES
  }
  return (int)hipGetLastError();
}

int launch_prep(const LinesDev &L, const LayersDev &A, const GridParams &gp, const WeightMode &W, int line_lo,
                int n_sub, int cold_lo, int cold_hi, FastRec *fast, ColdRec *cold, hipStream_t st) {
  if (n_sub <= 0 || A.n_layers <= 0) return 0;
  dim3 grid((n_sub + kPrepBlock - 1) / kPrepBlock, A.n_layers);
  hipLaunchKernelGGL(sr_prep_kernel, grid, dim3(kPrepBlock), 0, st, L, A, gp, W, line_lo, n_sub, cold_lo, cold_hi, fast,
                     cold);
  return (int)hipGetLastError();
}

int abscoeff_tile_points(int variant) { return 64 * (variant == 4 ? 4 : 8); }

int launch_abscoeff(int variant, int which, const FastRec *fast, const ColdRec *cold, const IcIndex &ix,
                    const int *zmax, int n_sub, int n_layers, int g_lo, int g_hi, const GridParams &gp,
                    double *abs_out, double *emi_out, hipStream_t st) {
  if (g_hi <= g_lo || n_layers <= 0) return 0;
  if (which == 0) {
    const int tp = abscoeff_tile_points(variant);
    const int n_tiles = (g_hi - g_lo + tp - 1) / tp;
    dim3 grid((unsigned)(n_tiles * n_layers));
    if (variant == 4)
      hipLaunchKernelGGL((sr_abscoeff_wings_kernel<4>), grid, dim3(64), 0, st, fast, ix, n_sub, n_tiles,
                         g_lo, g_hi, abs_out, emi_out);
    else
      hipLaunchKernelGGL((sr_abscoeff_wings_kernel<8>), grid, dim3(64), 0, st, fast, ix, n_sub, n_tiles,
                         g_lo, g_hi, abs_out, emi_out);
  } else {
    const int n_groups = (g_hi - g_lo + kGroup - 1) / kGroup;
    dim3 grid((unsigned)(n_groups * n_layers));
    hipLaunchKernelGGL(sr_abscoeff_cores_kernel, grid, dim3(64), 0, st, fast, cold, ix, zmax, n_sub, n_groups,
                       g_lo, g_hi, gp, abs_out, emi_out);
  }
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------
// radiance recursion (build's own definition, see include/spectrobot_hip.h)
// ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sr_radiance_kernel(const double *__restrict__ abs_c,
                                                          const double *__restrict__ emi_c,
                                                          int n_pts, int /*n_rays: gridDim.y*/,
                                                          const int *__restrict__ seg_off,
                                                          const int *__restrict__ seg_layer,
                                                          const double *__restrict__ seg_col,
                                                          int init_from_rad, double *__restrict__ rad) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int ray = blockIdx.y;
  if (j >= n_pts) return;
  double I = init_from_rad ? rad[(size_t)ray * n_pts + j] : 0.0;
  const int s0 = seg_off[ray], s1 = seg_off[ray + 1];
  // The recursion is sequential in the segments and a 1-ray launch has only ~1.5 waves per SIMD,
  // so the coefficient loads of kB segments are issued together ahead of their use (one memory
  // latency per kB segments instead of one per segment: 147 -> ~40 us for 160 segments x 1e5 points)
  constexpr int kB = 8;
  for (int sb = s0; sb < s1; sb += kB) {
    double a[kB], e[kB], u[kB];
#pragma unroll
    for (int t = 0; t < kB; ++t) {
      const int s = min(sb + t, s1 - 1);
      const size_t o = (size_t)seg_layer[s] * n_pts + j;
      a[t] = abs_c[o];
      e[t] = emi_c[o];
      u[t] = seg_col[s];
    }
#pragma unroll
    for (int t = 0; t < kB; ++t) {
      if (sb + t < s1) {
        const Atten A = attenuation(a[t] * u[t]);
        I = I * A.t + (e[t] * u[t]) * A.f;
      }
    }
  }
  rad[(size_t)ray * n_pts + j] = I;
}

// Radiance and its derivatives with respect to NP retrieval parameters the absorber columns
// depend on linearly (VMR profile parameters, spect_main_module.py:319-375: col_s = sum_p
// dcol[s][p] * x_p).  Forward sensitivity of the same recursion:
//   d/dcol [I e^-tau + emi col (1 - e^-tau)/tau] = e^-tau (emi - abs I)
//   J_p <- J_p e^-tau + e^-tau (emi - abs I_prev) dcol[s][p];   I <- I e^-tau + src
// Thread = (point, ray, chunk of NP parameters).
template <int NP>
__global__ __launch_bounds__(256) void sr_radiance_jac_kernel(const double *__restrict__ abs_c,
                                                              const double *__restrict__ emi_c, int n_pts,
                                                              int /*n_rays: gridDim.y*/, const int *__restrict__ seg_off,
                                                              const int *__restrict__ seg_layer,
                                                              const double *__restrict__ seg_col,
                                                              const double *__restrict__ dcol, int n_par,
                                                              double *__restrict__ rad, double *__restrict__ jac) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int ray = blockIdx.y, p0 = blockIdx.z * NP;
  if (j >= n_pts) return;
  double I = 0.0, J[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) J[q] = 0.0;
  const int s0 = seg_off[ray], s1 = seg_off[ray + 1];
  constexpr int kB = 8; // coefficient loads of kB segments ahead of the recursion, as in sr_radiance_kernel
  for (int sb = s0; sb < s1; sb += kB) {
    double av[kB], ev[kB];
#pragma unroll
    for (int t = 0; t < kB; ++t) {
      const size_t o = (size_t)seg_layer[min(sb + t, s1 - 1)] * n_pts + j;
      av[t] = abs_c[o];
      ev[t] = emi_c[o];
    }
#pragma unroll
    for (int k = 0; k < kB; ++k) {
      const int s = sb + k;
      if (s < s1) {
        const double u = seg_col[s], a = av[k], e = ev[k];
        const Atten A = attenuation(a * u);
        const double t = A.t, src = (e * u) * A.f;
        const double g = t * (e - a * I);
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          const double d = (p0 + q < n_par) ? dcol[(size_t)s * n_par + p0 + q] : 0.0;
          J[q] = fma(J[q], t, g * d);
        }
        I = I * t + src;
      }
    }
  }
  if (blockIdx.z == 0) rad[(size_t)ray * n_pts + j] = I;
#pragma unroll
  for (int q = 0; q < NP; ++q)
    if (p0 + q < n_par) jac[((size_t)ray * n_par + p0 + q) * n_pts + j] = J[q];
}

// Radiance derivatives with respect to ONE scalar per layer (its temperature, ...) through the
// layer's own coefficients: dabs[k][j], demi[k][j] = d(abs, emi of layer k)/d(parameter of layer k).
// Forward sensitivity of the recursion I <- I t + src, t = e^-tau, src = emi u f(tau), f = (1 - e^-tau)/tau:
//   dt = -u t dabs,  dsrc = u f demi + emi u^2 f'(tau) dabs,  f' = (tau e^-tau - (1 - e^-tau))/tau^2
//   J_k <- J_k t + [seg_layer == k] (I_prev dt + dsrc).
// Thread = (point, ray, chunk of NP layers); build's definition like the recursion itself.
template <int NP>
__global__ __launch_bounds__(256) void sr_radiance_jac_layer_kernel(
    const double *__restrict__ abs_c, const double *__restrict__ emi_c, const double *__restrict__ dabs,
    const double *__restrict__ demi, int n_pts, int n_layers, const int *__restrict__ seg_off,
    const int *__restrict__ seg_layer, const double *__restrict__ seg_col, double *__restrict__ jac) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int ray = blockIdx.y, p0 = blockIdx.z * NP;
  if (j >= n_pts) return;
  double I = 0.0, J[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) J[q] = 0.0;
  const int s0 = seg_off[ray], s1 = seg_off[ray + 1];
  for (int s = s0; s < s1; ++s) {
    const int k = seg_layer[s];
    const size_t o = (size_t)k * n_pts + j;
    const double u = seg_col[s], a = abs_c[o], e = emi_c[o];
    const double tau = a * u;
    const Atten A = attenuation(tau);
    const double t = A.t, em1 = A.em1, f = A.f;
    const bool thin = A.thin;
    const double src = (e * u) * f;
    if (k >= p0 && k < p0 + NP) { // this segment's layer is one of this thread's parameters
      const double fp = thin ? -0.5 : (tau * t - em1) * (A.rtau * A.rtau);
      const double da = dabs[o], de = demi[o];
      const double d = I * (-u * t * da) + u * f * de + e * u * u * fp * da;
#pragma unroll
      for (int q = 0; q < NP; ++q) J[q] = fma(J[q], t, (k == p0 + q) ? d : 0.0);
    } else {
#pragma unroll
      for (int q = 0; q < NP; ++q) J[q] *= t;
    }
    I = I * t + src;
  }
#pragma unroll
  for (int q = 0; q < NP; ++q)
    if (p0 + q < n_layers) jac[((size_t)ray * n_layers + p0 + q) * n_pts + j] = J[q];
}

// ------------------------------------------------------------------------
// Device LOS pipeline (the build's own definition standing in for the absent sbm
// LineOfSight.calc_radtran_steps / radtran_fast, call sites spect_main_module.py:2746-2767, 2834-2845).
//   sr_los_columns_kernel   Curtis-Godson columns of every (gas | profile parameter, segment):
//                           curgod_fort_2 (curgods.f:24-45) over the segment's LOS sample points
//   sr_limb_kernel          recursion of a ray batch over n_gas gases with the call-site options
//                           solo_absorption / initial_intensity (Planck) (radtran_3D_ch4.py:297-315)
//   sr_limb_jac_kernel      + derivatives w.r.t. VMR-profile parameters (columns linear in them)
//   sr_limb_jac_layer_kernel  + derivatives w.r.t. one scalar per layer acting through the coefficients
//   (both: forward sensitivities, NP parameters per thread; many parameters / layers: sr_limb_adjoint_kernel below)
// Per segment s of a ray, layer k = seg_layer[s], columns u_g = col[g][s]:
//   tau = sum_g abs_g[k] u_g,  E = sum_g emi_g[k] u_g,  t = e^-tau,  f = (1 - t)/tau
//   I <- I t + E f          (E f dropped with solo_absorption)
// ------------------------------------------------------------------------
__global__ void sr_los_columns_kernel(const double *__restrict__ nd, const double *__restrict__ x,
                                      const double *__restrict__ prof, // [n_prof][n_pt]: vmr per gas, then weights per parameter
                                      const double *__restrict__ scale, // [n_prof]
                                      const int *__restrict__ pt_off, int n_seg, int n_pt,
                                      double *__restrict__ col) {      // [n_prof][n_seg]
  const int s = blockIdx.x * blockDim.x + threadIdx.x, q = blockIdx.y;
  if (s >= n_seg) return;
  const double *vmr = prof + (size_t)q * n_pt;
  double acc = 0.0;
  for (int i = pt_off[s]; i < pt_off[s + 1] - 1; ++i) { // curgods.f:33-43
    const double dx = x[i + 1] - x[i];
    const double A = nd[i] * vmr[i];
    const double B = nd[i] * (vmr[i + 1] - vmr[i]) / dx;
    const double fu = nd[i + 1] / nd[i];
    const double D = log(fu) / dx;
    acc = acc + (A * D * (fu - 1.) + B * fu * (D * dx - 1.) + B) / (D * D);
  }
  col[(size_t)q * n_seg + s] = scale[q] * acc;
}

// The VMRs of the retrieved gases at a resident batch's sample points from the parameter vector of a retrieval:
// prof[g][i] = sum over the parameters p of gas g of x_p w_p[i], w_p = prof[n_gas + p] the parameter's mask at the
// sample points.  (A profile that is sum_p mask_p x_p on the altitude levels, interpolated linearly to the sample
// points, is the same sum of the interpolated masks: spect_main_module.py LinearProfile_1D.profile.)  Gases without
// parameters keep their row.
// The parameter vector travels as a kernel argument (up to kVmrParArg values: no staging copy in front of a retrieval
// iteration's first kernel); longer ones in blocks of kVmrParArg, each block adding to the row the first one started.
struct ParVec { double x[kVmrParArg]; };
__global__ void sr_los_vmr_from_params_kernel(double *__restrict__ prof, int n_gas, int n_par, int n_pt,
                                              const int *__restrict__ par_gas, ParVec xv, int p0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, g = blockIdx.y;
  if (i >= n_pt) return;
  bool any = false;
  for (int p = 0; p < p0; ++p) any = any || par_gas[p] == g; // (an earlier block started the row)
  double v = any ? prof[(size_t)g * n_pt + i] : 0.0;
  for (int p = p0; p < min(n_par, p0 + kVmrParArg); ++p)
    if (par_gas[p] == g) {
      v = v + xv.x[p - p0] * prof[(size_t)(n_gas + p) * n_pt + i];
      any = true;
    }
  if (any) prof[(size_t)g * n_pt + i] = v;
}
int launch_los_vmr_from_params(double *prof, int n_gas, int n_par, int n_pt, const int *par_gas, const double *x_host, hipStream_t st) {
  if (n_pt <= 0 || n_gas <= 0 || n_par <= 0) return 0;
  for (int p0 = 0; p0 < n_par; p0 += kVmrParArg) {
    ParVec xv;
    for (int p = 0; p < kVmrParArg; ++p) xv.x[p] = p0 + p < n_par ? x_host[p0 + p] : 0.0;
    hipLaunchKernelGGL(sr_los_vmr_from_params_kernel, dim3((n_pt + 255) / 256, n_gas), dim3(256), 0, st, prof, n_gas, n_par, n_pt,
                       par_gas, xv, p0);
  }
  return (int)hipGetLastError();
}

int launch_los_columns(const double *nd, const double *x, const double *prof, const double *scale, const int *pt_off,
                       int n_seg, int n_pt, int n_prof, double *col, hipStream_t st) {
  if (n_seg <= 0 || n_prof <= 0) return 0;
  hipLaunchKernelGGL(sr_los_columns_kernel, dim3((n_seg + 63) / 64, n_prof), dim3(64), 0, st, nd, x, prof, scale, pt_off,
                     n_seg, n_pt, col);
  return (int)hipGetLastError();
}

__device__ inline double limb_initial(const LimbOpts &o, const double *rad, size_t at, int j) {
  if (o.init_mode == 1) return rad[at];
  if (o.init_mode == 2) { // Calc_BB, spect_classes.py:1886
    const double nu = o.w0 + (double)(o.g_lo + j) * o.gstep;
    return 2 * kHcgs * (kCcgs * kCcgs) * (nu * nu * nu) / (exp(kC2 * nu / o.t_init) - 1);
  }
  return 0.0;
}

// Block -> (point block, ray) of the ray-batch kernels.  With the rays on the grid's slow axis the resident blocks
// work on one or two rays at a time and every ray streams the coefficient tables from HBM again (64 rays: 4.0 GB
// fetched for 0.13 GB of tables, the kernel bound by that).  Here the point blocks are dealt round-robin to the
// eight XCDs (workgroup ids are) and ALL rays of a point block follow each other on one XCD: the 80 layers x 2 KB
// of a point block are read from HBM once and by the other rays from that XCD's L2.
__device__ inline bool limb_block(int n_pb, int n_rays, int &pb, int &ray) {
  const int id = blockIdx.x, q = id >> 3;
  ray = q % n_rays;
  pb = (id & 7) + 8 * (q / n_rays);
  return pb < n_pb;
}
__host__ inline unsigned limb_grid(int n_pb, int n_rays) { return (unsigned)(((n_pb + 7) / 8) * 8) * (unsigned)n_rays; }

template <int NG>
__global__ __launch_bounds__(256) void sr_limb_kernel(const double *__restrict__ abs_c, const double *__restrict__ emi_c,
                                                      int n_pts, int n_layers, const int *__restrict__ seg_off,
                                                      const int *__restrict__ seg_layer, const double *__restrict__ col,
                                                      LimbOpts o, int n_rays, double *__restrict__ rad) {
  int pb, ray;
  if (!limb_block((n_pts + 255) / 256, n_rays, pb, ray)) return;
  const int j = pb * 256 + threadIdx.x;
  if (j >= n_pts) return;
  double I = limb_initial(o, rad, (size_t)ray * n_pts + j, j);
  const int s0 = seg_off[ray], s1 = seg_off[ray + 1];
  constexpr int kB = NG == 1 ? 8 : (NG == 2 ? 4 : 2); // segments whose loads are issued together (see sr_radiance_kernel)
  const size_t gstride = (size_t)n_layers * n_pts;
  for (int sb = s0; sb < s1; sb += kB) {
    double a[kB][NG], e[kB][NG], u[kB][NG];
#pragma unroll
    for (int t = 0; t < kB; ++t) {
      const int s = min(sb + t, s1 - 1);
      const size_t ofs = (size_t)seg_layer[s] * n_pts + j;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        a[t][g] = abs_c[g * gstride + ofs];
        e[t][g] = emi_c[g * gstride + ofs];
        u[t][g] = col[(size_t)g * o.n_seg_total + s];
      }
    }
#pragma unroll
    for (int t = 0; t < kB; ++t) {
      if (sb + t < s1) {
        double tau = a[t][0] * u[t][0], E = e[t][0] * u[t][0];
#pragma unroll
        for (int g = 1; g < NG; ++g) {
          tau = tau + a[t][g] * u[t][g];
          E = E + e[t][g] * u[t][g];
        }
        const Atten A = attenuation(tau);
        I = I * A.t + (o.solo_absorption ? 0.0 : E * A.f);
      }
    }
  }
  rad[(size_t)ray * n_pts + j] = I;
}

// The same recursion for SMALL launches (one ray on a 1/8 spectral shard: 49 blocks on 256 CUs), where the run time
// is the latency of one thread's chain of 160 dependent segments: the segments of a ray are cut into kLimbParts
// consecutive stretches, one wave per stretch and 64 points; a stretch is the affine map I -> I T + S (T = product of
// its transmissions, S its own emission attenuated by what follows inside the stretch), and maps compose in order.
constexpr int kLimbParts = 4;
template <int NG>
__global__ __launch_bounds__(64 * kLimbParts) void sr_limb_split_kernel(
    const double *__restrict__ abs_c, const double *__restrict__ emi_c, int n_pts, int n_layers,
    const int *__restrict__ seg_off, const int *__restrict__ seg_layer, const double *__restrict__ col, LimbOpts o,
    int n_rays, double *__restrict__ rad) {
  __shared__ double s_T[kLimbParts][64], s_S[kLimbParts][64];
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  int pb, ray;
  if (!limb_block((n_pts + 63) / 64, n_rays, pb, ray)) return; // block-uniform: no thread reaches the barrier
  const int j = pb * 64 + lane;
  const int jj = min(j, n_pts - 1);
  const int r0 = seg_off[ray], r1 = seg_off[ray + 1], n = r1 - r0;
  const int s0 = r0 + (int)(((long)n * part) / kLimbParts), s1 = r0 + (int)(((long)n * (part + 1)) / kLimbParts);
  constexpr int kB = NG == 1 ? 8 : (NG == 2 ? 4 : 2);
  const size_t gstride = (size_t)n_layers * n_pts;
  double T = 1.0, S = 0.0;
  for (int sb = s0; sb < s1; sb += kB) {
    double a[kB][NG], e[kB][NG], u[kB][NG];
#pragma unroll
    for (int t = 0; t < kB; ++t) {
      const int s = min(sb + t, s1 - 1);
      const size_t ofs = (size_t)seg_layer[s] * n_pts + jj;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        a[t][g] = abs_c[g * gstride + ofs];
        e[t][g] = emi_c[g * gstride + ofs];
        u[t][g] = col[(size_t)g * o.n_seg_total + s];
      }
    }
#pragma unroll
    for (int t = 0; t < kB; ++t) {
      if (sb + t < s1) {
        double tau = a[t][0] * u[t][0], E = e[t][0] * u[t][0];
#pragma unroll
        for (int g = 1; g < NG; ++g) {
          tau = tau + a[t][g] * u[t][g];
          E = E + e[t][g] * u[t][g];
        }
        const Atten A = attenuation(tau);
        T = T * A.t;
        S = S * A.t + (o.solo_absorption ? 0.0 : E * A.f);
      }
    }
  }
  s_T[part][lane] = T;
  s_S[part][lane] = S;
  __syncthreads();
  if (part == 0 && j < n_pts) {
    double I = limb_initial(o, rad, (size_t)ray * n_pts + j, j);
#pragma unroll
    for (int p = 0; p < kLimbParts; ++p) I = I * s_T[p][lane] + s_S[p][lane];
    rad[(size_t)ray * n_pts + j] = I;
  }
}

// Derivatives w.r.t. NP parameters per thread; parameter p belongs to gas par_gas[p] and moves its columns
// linearly, d u_g[s] / d x_p = dcol[p][s] (profile parameters of RetParam / LinearProfile, smm:319-375):
//   d tau = a_g D,  d(E f) = e_g D f + E f'(tau) a_g D,  f' = (tau t - (1 - t))/tau^2
//   J_p <- J_p t + (-I t a_g + e_g f + E f' a_g) D      (single gas: t (e - a I) D)
template <int NG, int NP>
__global__ __launch_bounds__(256) void sr_limb_jac_kernel(
    const double *__restrict__ abs_c, const double *__restrict__ emi_c, int n_pts, int n_layers,
    const int *__restrict__ seg_off, const int *__restrict__ seg_layer, const double *__restrict__ col,
    const double *__restrict__ dcol, const int *__restrict__ par_gas, int n_par, LimbOpts o,
    double *__restrict__ rad, double *__restrict__ jac) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, ray = blockIdx.y, p0 = blockIdx.z * NP;
  if (j >= n_pts) return;
  double I = limb_initial(o, rad, (size_t)ray * n_pts + j, j), J[NP];
  int pg[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    J[q] = 0.0;
    pg[q] = p0 + q < n_par ? par_gas[p0 + q] : 0;
  }
  const int s0 = seg_off[ray], s1 = seg_off[ray + 1];
  const size_t gstride = (size_t)n_layers * n_pts;
  for (int s = s0; s < s1; ++s) {
    const size_t ofs = (size_t)seg_layer[s] * n_pts + j;
    double a[NG], e[NG];
    double tau = 0.0, E = 0.0;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      a[g] = abs_c[g * gstride + ofs];
      e[g] = emi_c[g * gstride + ofs];
      const double u = col[(size_t)g * o.n_seg_total + s];
      tau = g == 0 ? a[g] * u : tau + a[g] * u;
      E = g == 0 ? e[g] * u : E + e[g] * u;
    }
    const Atten A = attenuation(tau);
    const double t = A.t, em1 = A.em1, f = A.f;
    const bool thin = A.thin;
    const double fp = thin ? -0.5 : (tau * t - em1) * (A.rtau * A.rtau);
    const double src = o.solo_absorption ? 0.0 : E * f;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      double ag = a[0], eg = e[0];
#pragma unroll
      for (int g = 1; g < NG; ++g) {
        ag = pg[q] == g ? a[g] : ag;
        eg = pg[q] == g ? e[g] : eg;
      }
      const double D = p0 + q < n_par ? dcol[(size_t)(p0 + q) * o.n_seg_total + s] : 0.0;
      const double dsrc = o.solo_absorption ? 0.0 : fma(E * fp, ag, eg * f);
      J[q] = fma(J[q], t, (dsrc - I * t * ag) * D);
    }
    I = I * t + src;
  }
  if (blockIdx.z == 0) rad[(size_t)ray * n_pts + j] = I;
#pragma unroll
  for (int q = 0; q < NP; ++q)
    if (p0 + q < n_par) jac[((size_t)ray * n_par + p0 + q) * n_pts + j] = J[q];
}

// Derivatives w.r.t. ONE scalar per layer acting through the layer's coefficients (temperature, ...):
// dabs / demi: [NG][n_layers][n_pts].  See sr_radiance_jac_layer_kernel for the recursion.
template <int NG, int NP>
__global__ __launch_bounds__(256) void sr_limb_jac_layer_kernel(
    const double *__restrict__ abs_c, const double *__restrict__ emi_c, const double *__restrict__ dabs,
    const double *__restrict__ demi, int n_pts, int n_layers, const int *__restrict__ seg_off,
    const int *__restrict__ seg_layer, const double *__restrict__ col, LimbOpts o, double *__restrict__ jac) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, ray = blockIdx.y, p0 = blockIdx.z * NP;
  if (j >= n_pts) return;
  double I = limb_initial(o, jac, 0, j), J[NP]; // init_mode 1 is refused by the host for this kernel
#pragma unroll
  for (int q = 0; q < NP; ++q) J[q] = 0.0;
  const int s0 = seg_off[ray], s1 = seg_off[ray + 1];
  const size_t gstride = (size_t)n_layers * n_pts;
  for (int s = s0; s < s1; ++s) {
    const int k = seg_layer[s];
    const size_t ofs = (size_t)k * n_pts + j;
    const bool mine = k >= p0 && k < p0 + NP;
    double tau = 0.0, E = 0.0, dtau = 0.0, dE = 0.0;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const double u = col[(size_t)g * o.n_seg_total + s];
      const double av = abs_c[g * gstride + ofs], ev = emi_c[g * gstride + ofs];
      tau = g == 0 ? av * u : tau + av * u;
      E = g == 0 ? ev * u : E + ev * u;
      if (mine) {
        dtau = fma(dabs[g * gstride + ofs], u, dtau);
        dE = fma(demi[g * gstride + ofs], u, dE);
      }
    }
    const Atten A = attenuation(tau);
    const double t = A.t, em1 = A.em1, f = A.f;
    const bool thin = A.thin;
    const double src = o.solo_absorption ? 0.0 : E * f;
    if (mine) {
      const double fp = thin ? -0.5 : (tau * t - em1) * (A.rtau * A.rtau);
      const double d = -I * t * dtau + (o.solo_absorption ? 0.0 : dE * f + E * fp * dtau);
#pragma unroll
      for (int q = 0; q < NP; ++q) J[q] = fma(J[q], t, (k == p0 + q) ? d : 0.0);
    } else {
#pragma unroll
      for (int q = 0; q < NP; ++q) J[q] *= t;
    }
    I = I * t + src;
  }
#pragma unroll
  for (int q = 0; q < NP; ++q)
    if (p0 + q < n_layers) jac[((size_t)ray * n_layers + p0 + q) * n_pts + j] = J[q];
}

// ------------------------------------------------------------------------
// Radiances and BOTH kinds of Jacobian of a ray batch in ONE pass over each ray (round 3).
//
// I_final = sum_s src_s T(s+1 .. end) + I_0 T(all), so the derivative with respect to anything that acts through
// segment s alone is that segment's own sensitivity times the transmission of everything behind it:
//   w_tau = (-I_prev t + E f') T_after,   w_E = f T_after          (d I_final / d tau_s,  / d E_s)
//   per-layer scalar (temperature):   J_k += w_tau dtau_k u + w_E dE_k u            over the segments in layer k
//   column parameter p of gas g:      J_p += (w_tau a_g + w_E e_g) dcol[p][s]       over the segments p touches
// T_after = exp(-(tau_total - tau(0..s))): a first sweep adds up tau_total (no exp; error-free two-sums, see the
// kernel), the second runs the recursion.
// The forward-sensitivity kernels carry NP accumulators through the whole recursion and repeat it per block of NP
// parameters: 80 per-layer VMR parameters cost five recursions of 160 x (60 + 16 x 4) instructions; here one of
// 160 x ~100.  What remains is the traffic of the Jacobian rows (configs[3]: 2 x 1 GB per set of 8 rays), so the
// host plans the accumulation per ray (SegProg):
//   * a layer's (parameter's) FIRST touch in the ray stores, later ones add: no memset, no read of zeros;
//     rows the ray never touches are zero-filled by the kernel;
//   * a parameter touched by consecutive segments (a level's VMR acts on the shells above and below it) is
//     carried in one of four registers across them and written once per run: a limb path touches a level on its
//     way down and again on its way up -- one store, one read-add-store instead of four accesses.
// Everything in a SegProg is wave-uniform (a block works on one ray): scalar loads and scalar branches.
// ------------------------------------------------------------------------
constexpr int kAdjEnt = 4; // column parameters a segment may touch (more: the forward-sensitivity kernel is used)
struct __attribute__((aligned(16))) SegProg {
  int layer;            // row of the coefficient tables
  int flags;            // bit 0: first touch of the per-layer Jacobian row in this ray (store), else add
  int n_ent;
  int jrow;             // row of the per-layer Jacobian (= layer, or the altitude layer of a 3-D path's step)
  int ent_p[kAdjEnt];   // parameter
  int ent_gf[kAdjEnt];  // gas | slot << 8 | flags << 16: bit 0 run starts here (no carry in), bit 1 run ends here
                        // (write), bit 2 first write of the parameter in this ray (store)
  double u[4];          // column of every gas
  double dc[kAdjEnt];   // d col / d x_p of the entries
};
static_assert(sizeof(SegProg) == 112, "SegProg layout");

// The host's plan (layer, flags, entries) + the columns the device computed -> one record per segment.
__global__ void sr_adj_pack_kernel(const int *__restrict__ plan, // [n_seg][4 + 2 kAdjEnt] ints: layer, flags, n_ent, 0, ent_p, ent_gf
                                   const double *__restrict__ col, int n_gas, int n_seg, SegProg *__restrict__ out) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_seg) return;
  const int *pl = plan + (size_t)s * (4 + 2 * kAdjEnt);
  SegProg r;
  r.layer = pl[0]; r.flags = pl[1]; r.n_ent = pl[2]; r.jrow = pl[3];
#pragma unroll
  for (int i = 0; i < kAdjEnt; ++i) {
    r.ent_p[i] = pl[4 + i];
    r.ent_gf[i] = pl[4 + kAdjEnt + i];
    r.dc[i] = i < r.n_ent ? col[(size_t)(n_gas + r.ent_p[i]) * n_seg + s] : 0.0;
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) r.u[g] = g < n_gas ? col[(size_t)g * n_seg + s] : 0.0;
  out[s] = r;
}

template <int NG, bool LAYER, bool PAR>
__global__ __launch_bounds__(256) void sr_limb_adjoint_kernel(
    const double *__restrict__ abs_c, const double *__restrict__ emi_c, const double *__restrict__ dabs,
    const double *__restrict__ demi, int n_pts, int n_layers, int n_jrows, const int *__restrict__ seg_off,
    const SegProg *__restrict__ prog, const int *__restrict__ zero_off, const int *__restrict__ zero_row, int n_par,
    LimbOpts o, double *__restrict__ rad, double *__restrict__ jac_layer, double *__restrict__ jac_par) {
  // (rays on the slow grid axis: this kernel is bound by its Jacobian writes; with the limb_block order of the
  // radiance kernels -- all rays of a point block on one XCD -- it measured 2.21 instead of 2.11 ms on configs[3])
  const int j = blockIdx.x * blockDim.x + threadIdx.x, ray = blockIdx.y;
  if (j >= n_pts) return;
  const int s0 = seg_off[ray], s1 = seg_off[ray + 1];
  const size_t gstride = (size_t)n_layers * n_pts;
  // rows this ray never touches: row < n_jrows: per-layer Jacobian row, else parameter row - n_jrows
  for (int q = zero_off[ray]; q < zero_off[ray + 1]; ++q) {
    const int row = zero_row[q];
    if (row < n_jrows) {
      if (LAYER) jac_layer[((size_t)ray * n_jrows + row) * n_pts + j] = 0.0;
    } else if (PAR) {
      jac_par[((size_t)ray * n_par + (row - n_jrows)) * n_pts + j] = 0.0;
    }
  }
  // Optical depth of the segments not yet passed, as an unevaluated sum rem + rem_lo (two-sum: every addition's
  // rounding error is kept).  The first sweep adds the segments' tau up, the recursion takes the SAME values off again
  // one by one: at a line centre the path's total can be 1e9 while the last segments' own depth is 0.1 -- carried in
  // one double the difference would be good to 1e-7 only, and with it the transmission behind the near-side layers.
  double rem = 0.0, rem_lo = 0.0;
  for (int s = s0; s < s1; ++s) {
    const SegProg &P = prog[s];
    const size_t ofs = (size_t)P.layer * n_pts + j;
    double tau = 0.0;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const double ag = abs_c[g * gstride + ofs];
      tau = g == 0 ? ag * P.u[g] : tau + ag * P.u[g];   // as the recursion below forms it
    }
    const double sm = rem + tau, bb = sm - rem;
    rem_lo += (rem - (sm - bb)) + (tau - bb);
    rem = sm;
  }
  double I = limb_initial(o, rad, (size_t)ray * n_pts + j, j);
  double slot[4] = {0., 0., 0., 0.};
  constexpr int kB = NG == 1 ? 4 : 2; // segments whose coefficient loads are issued together
  for (int sb = s0; sb < s1; sb += kB) {
    double a[kB][NG], e[kB][NG], da[kB][NG], de[kB][NG];
#pragma unroll
    for (int t = 0; t < kB; ++t) {
      const int s = min(sb + t, s1 - 1);
      const size_t ofs = (size_t)prog[s].layer * n_pts + j;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        a[t][g] = abs_c[g * gstride + ofs];
        e[t][g] = emi_c[g * gstride + ofs];
        if (LAYER) {
          da[t][g] = dabs[g * gstride + ofs];
          de[t][g] = demi[g * gstride + ofs];
        }
      }
    }
#pragma unroll
    for (int t = 0; t < kB; ++t) {
      const int s = sb + t;
      if (s >= s1) break;
      const SegProg &P = prog[s];
      double tau = 0.0, E = 0.0, dtau = 0.0, dE = 0.0;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const double u = P.u[g];
        tau = g == 0 ? a[t][g] * u : tau + a[t][g] * u;
        E = g == 0 ? e[t][g] * u : E + e[t][g] * u;
        if (LAYER) {
          dtau = fma(da[t][g], u, dtau);
          dE = fma(de[t][g], u, dE);
        }
      }
      const Atten A = attenuation(tau);
      const double fp = A.thin ? -0.5 : (tau * A.t - A.em1) * (A.rtau * A.rtau);
      {
        const double sm = rem - tau, bb = sm - rem;
        rem_lo += (rem - (sm - bb)) + (-tau - bb);
        rem = sm;
      }
      const double Ta = exp_bounded(fmin(fmax(-(rem + rem_lo), -700.0), 700.0)); // transmission of everything behind this segment
      const double w_tau = (o.solo_absorption ? -I * A.t : fma(E, fp, -I * A.t)) * Ta;
      const double w_E = o.solo_absorption ? 0.0 : A.f * Ta;
      if (LAYER) {
        const double d = fma(w_tau, dtau, w_E * dE);
        double *out = jac_layer + ((size_t)ray * n_jrows + P.jrow) * n_pts + j;
        if (P.flags & 1) *out = d; else *out += d;
      }
      if (PAR) {
        for (int i = 0; i < P.n_ent; ++i) {
          const int gf = P.ent_gf[i], g = gf & 0xff, sl = (gf >> 8) & 0xff, fl = gf >> 16;
          double ag = a[t][0], eg = e[t][0];
#pragma unroll
          for (int q = 1; q < NG; ++q) {
            ag = g == q ? a[t][q] : ag;
            eg = g == q ? e[t][q] : eg;
          }
          double v = fma(w_tau, ag, w_E * eg) * P.dc[i];
          if (!(fl & 1)) v += sl == 0 ? slot[0] : (sl == 1 ? slot[1] : (sl == 2 ? slot[2] : slot[3]));
          if (fl & 2) {
            double *out = jac_par + ((size_t)ray * n_par + P.ent_p[i]) * n_pts + j;
            if (fl & 4) *out = v; else *out += v;
          } else {
            if (sl == 0) slot[0] = v; else if (sl == 1) slot[1] = v; else if (sl == 2) slot[2] = v; else slot[3] = v;
          }
        }
      }
      I = I * A.t + (o.solo_absorption ? 0.0 : E * A.f);
    }
  }
  if (rad) rad[(size_t)ray * n_pts + j] = I;
}

// ------------------------------------------------------------------------
// The same recursion, LAYER-SYNCHRONOUS over a batch of NR rays (round 4).  sr_limb_adjoint_kernel gives every ray
// its own threads, and every ray streams the four coefficient tables from HBM again, twice (the optical-depth sweep
// and the recursion): configs[3], 8 rays: 8.4 GB fetched + 3.5 GB written per set for 2.6 GB of algorithmic bytes, the
// kernel bound by that (profiles/r03_config3_pmc_limb_kernels.txt).  In a 1-D atmosphere all rays walk the SAME
// coefficient rows, down to their tangent shells and up again: a thread here owns one grid point for NR rays, the host
// lists the shells in that order (`sched`: per visit the layer and every ray's segment there, -1 = the ray is not in
// this shell on this side), the shell's coefficients are loaded ONCE per visit and every ray that crosses it takes its
// step -- per ray exactly the operations of sr_limb_adjoint_kernel in exactly its order, so the results are bit for
// bit the same.  State per ray: I, the two-sum of the remaining optical depth, four carry slots.
// ------------------------------------------------------------------------
template <int NG, bool LAYER, bool PAR, int NR>
__global__ __launch_bounds__(256) void sr_limb_adjoint_sync_kernel(
    const double *__restrict__ abs_c, const double *__restrict__ emi_c, const double *__restrict__ dabs,
    const double *__restrict__ demi, int n_pts, int n_layers, int n_jrows, const SegProg *__restrict__ prog,
    const int *__restrict__ zero_off, const int *__restrict__ zero_row, int n_par, LimbOpts o,
    const int *__restrict__ sched, // [n_batches][n_visits][1 + NR]: layer, segment of each ray of the batch (or -1)
    int n_visits, int n_rays, double *__restrict__ rad, double *__restrict__ jac_layer, double *__restrict__ jac_par) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, batch = blockIdx.y;
  if (j >= n_pts) return;
  const int ray0 = batch * NR;
  const int *sc = sched + (size_t)batch * n_visits * (1 + NR);
  const size_t gstride = (size_t)n_layers * n_pts;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int ray = ray0 + r;
    if (ray >= n_rays) continue;
    for (int q = zero_off[ray]; q < zero_off[ray + 1]; ++q) { // rows this ray never touches
      const int row = zero_row[q];
      if (row < n_jrows) {
        if (LAYER) jac_layer[((size_t)ray * n_jrows + row) * n_pts + j] = 0.0;
      } else if (PAR) {
        jac_par[((size_t)ray * n_par + (row - n_jrows)) * n_pts + j] = 0.0;
      }
    }
  }
  double rem[NR], rem_lo[NR], I[NR], slot[NR][4];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    rem[r] = rem_lo[r] = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) slot[r][q] = 0.0;
  }
  // sweep 1: the optical depth of every ray's path, two-sums (see sr_limb_adjoint_kernel)
  for (int v = 0; v < n_visits; ++v) {
    const int *sv = sc + (size_t)v * (1 + NR);
    const size_t ofs = (size_t)sv[0] * n_pts + j;
    double a[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) a[g] = abs_c[g * gstride + ofs];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int sg = sv[1 + r];
      if (sg < 0) continue; // wave-uniform
      const SegProg &P = prog[sg];
      double tau = 0.0;
#pragma unroll
      for (int g = 0; g < NG; ++g) tau = g == 0 ? a[g] * P.u[g] : tau + a[g] * P.u[g];
      const double sm = rem[r] + tau, bb = sm - rem[r];
      rem_lo[r] += (rem[r] - (sm - bb)) + (tau - bb);
      rem[r] = sm;
    }
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) I[r] = ray0 + r < n_rays ? limb_initial(o, rad, (size_t)(ray0 + r) * n_pts + j, j) : 0.0;
  // sweep 2: the recursion, shell by shell
  for (int v = 0; v < n_visits; ++v) {
    const int *sv = sc + (size_t)v * (1 + NR);
    const size_t ofs = (size_t)sv[0] * n_pts + j;
    double a[NG], e[NG], da[NG], de[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      a[g] = abs_c[g * gstride + ofs];
      e[g] = emi_c[g * gstride + ofs];
      if (LAYER) {
        da[g] = dabs[g * gstride + ofs];
        de[g] = demi[g * gstride + ofs];
      }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int sg = sv[1 + r];
      if (sg < 0) continue;
      const int ray = ray0 + r;
      const SegProg &P = prog[sg];
      double tau = 0.0, E = 0.0, dtau = 0.0, dE = 0.0;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const double u = P.u[g];
        tau = g == 0 ? a[g] * u : tau + a[g] * u;
        E = g == 0 ? e[g] * u : E + e[g] * u;
        if (LAYER) {
          dtau = fma(da[g], u, dtau);
          dE = fma(de[g], u, dE);
        }
      }
      const Atten A = attenuation(tau);
      const double fp = A.thin ? -0.5 : (tau * A.t - A.em1) * (A.rtau * A.rtau);
      {
        const double sm = rem[r] - tau, bb = sm - rem[r];
        rem_lo[r] += (rem[r] - (sm - bb)) + (-tau - bb);
        rem[r] = sm;
      }
      const double Ta = exp_bounded(fmin(fmax(-(rem[r] + rem_lo[r]), -700.0), 700.0));
      const double w_tau = (o.solo_absorption ? -I[r] * A.t : fma(E, fp, -I[r] * A.t)) * Ta;
      const double w_E = o.solo_absorption ? 0.0 : A.f * Ta;
      if (LAYER) {
        const double d = fma(w_tau, dtau, w_E * dE);
        double *out = jac_layer + ((size_t)ray * n_jrows + P.jrow) * n_pts + j;
        if (P.flags & 1) *out = d; else *out += d;
      }
      if (PAR) {
        for (int i = 0; i < P.n_ent; ++i) {
          const int gf = P.ent_gf[i], g = gf & 0xff, sl = (gf >> 8) & 0xff, fl = gf >> 16;
          double ag = a[0], eg = e[0];
#pragma unroll
          for (int q = 1; q < NG; ++q) {
            ag = g == q ? a[q] : ag;
            eg = g == q ? e[q] : eg;
          }
          double val = fma(w_tau, ag, w_E * eg) * P.dc[i];
          if (!(fl & 1)) val += sl == 0 ? slot[r][0] : (sl == 1 ? slot[r][1] : (sl == 2 ? slot[r][2] : slot[r][3]));
          if (fl & 2) {
            double *out = jac_par + ((size_t)ray * n_par + P.ent_p[i]) * n_pts + j;
            if (fl & 4) *out = val; else *out += val;
          } else {
            if (sl == 0) slot[r][0] = val; else if (sl == 1) slot[r][1] = val; else if (sl == 2) slot[r][2] = val; else slot[r][3] = val;
          }
        }
      }
      I[r] = I[r] * A.t + (o.solo_absorption ? 0.0 : E * A.f);
    }
  }
  if (rad) {
#pragma unroll
    for (int r = 0; r < NR; ++r)
      if (ray0 + r < n_rays) rad[(size_t)(ray0 + r) * n_pts + j] = I[r];
  }
}

int launch_limb_adjoint_sync(const double *abs_c, const double *emi_c, const double *dabs, const double *demi, int n_pts,
                             int n_layers, int n_jrows, int n_rays, const SegProg *prog, const int *zero_off,
                             const int *zero_row, int n_par, const LimbOpts &o, const int *sched, int n_visits, double *rad,
                             double *jac_layer, double *jac_par, hipStream_t st) {
  if (n_pts <= 0 || n_rays <= 0 || n_visits <= 0) return 0;
  const dim3 grid((n_pts + 255) / 256, (n_rays + kAdjSyncRays - 1) / kAdjSyncRays);
#define SR_AS(NG, L, P) hipLaunchKernelGGL((sr_limb_adjoint_sync_kernel<NG, L, P, kAdjSyncRays>), grid, dim3(256), 0, st, abs_c, emi_c, \
                                           dabs, demi, n_pts, n_layers, n_jrows, prog, zero_off, zero_row, n_par, o, sched, n_visits,   \
                                           n_rays, rad, jac_layer, jac_par)
#define SR_AS3(NG)                                                     \
  do {                                                                 \
    if (jac_layer && jac_par) SR_AS(NG, true, true);                   \
    else if (jac_layer) SR_AS(NG, true, false);                        \
    else SR_AS(NG, false, true);                                       \
  } while (0)
  switch (o.n_gas) { case 1: SR_AS3(1); break; case 2: SR_AS3(2); break; case 3: SR_AS3(3); break; default: SR_AS3(4); break; }
#undef SR_AS3
#undef SR_AS
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------
// The FOLDED one-pass kernel (round 4): a limb ray crosses every shell above its tangent point twice, and both one-pass
// kernels above touch the shell's Jacobian row twice -- a store on the far side, a read-add-store on the near side --
// and stream the coefficient tables along the path, 2 x 80 visits per sweep.  Here a thread walks the SHELLS once per
// sweep, from the outermost inwards, and takes a ray's far-side and near-side segment of a shell TOGETHER:
//   far side, in path order:   I_f <- I_f t + E f  (the recursion itself), the transmission behind the segment from
//                              the two-sum of the remaining optical depth (as sr_limb_adjoint_kernel);
//   near side, from the observer inwards:  Tn = prod t of the near-side segments outside this one (the transmission
//                              behind it, no subtraction), cs = sum of E f Tn over them (what they contribute to the
//                              observed radiance), and what ENTERS the segment, seen at the observer, is the rest:
//                              I_in t Tn = I_obs - cs(including this segment) -- I_obs from sweep 1,
//                              I_obs = I_f(tangent) Tn(all) + cs(all).
// The weights are those of sr_limb_adjoint_kernel, w_tau = (E f' - I_in t) Ta, w_E = f Ta, the two segments' values
// of a row are added in registers and stored ONCE; a column parameter's touches on both sides fall into the same run
// of shells and are carried in one register.  Per set of configs[3]: 80 visits x (2 + 4) table rows per ray batch
// instead of 160 x (1 + 4), 1 access per Jacobian value instead of 1.7.  I_obs and cs are carried as error-free sums
// of two doubles, so that the difference keeps the relative accuracy of what is still to come (see sweep 2).
// Segments of one shell with the same columns (a 1-D limb path is symmetric) share one attenuation().
// ------------------------------------------------------------------------
#ifndef SR_FOLD_XCD
#define SR_FOLD_XCD 1 // all rays of a point block on one XCD, one after the other (limb_block): 1.22 vs 1.28 ms per configs[3] set
#endif
#ifndef SR_FOLD_WAVES
#define SR_FOLD_WAVES 4 // waves per SIMD the register allocation of the folded kernel aims at
#endif
struct __attribute__((aligned(16))) FoldRec { // one ray in one shell
  int layer;            // row of the coefficient tables (1-D atmospheres: of both segments; 3-D paths: of the far-side one)
  int has;              // bit 0: far-side segment, bit 1: near-side segment, bit 2: both, with the same columns (and rows)
  int n_ent;
  int layer_n;          // row of the near-side segment's coefficients (3-D paths: every LOS step has its own; else = layer)
  int jrow, pad[3];     // row of the per-layer Jacobian (the shell)
  int ent_p[kAdjEnt], ent_gf[kAdjEnt]; // as SegProg
  double u_f[4], u_n[4];               // columns of the two segments
  double dc_f[kAdjEnt], dc_n[kAdjEnt]; // d col / d x_p of the entries, per side (0: that side does not touch it)
};
static_assert(sizeof(FoldRec) == 192, "FoldRec layout");

// plan: [n_rec][kFoldPlanInts] ints: layer, far segment, near segment (-1: none), n_ent, ent_p, ent_gf, layer_n, jrow
__global__ void sr_fold_pack_kernel(const int *__restrict__ plan, const double *__restrict__ col, int n_gas, int n_seg, int n_rec,
                                    FoldRec *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rec) return;
  const int *pl = plan + (size_t)i * kFoldPlanInts;
  const int sf = pl[1], sn = pl[2];
  FoldRec r;
  r.layer = pl[0]; r.n_ent = pl[3]; r.layer_n = pl[12]; r.jrow = pl[13]; r.pad[0] = r.pad[1] = r.pad[2] = 0;
  bool same = sf >= 0 && sn >= 0 && r.layer == r.layer_n;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    r.u_f[g] = g < n_gas && sf >= 0 ? col[(size_t)g * n_seg + sf] : 0.0;
    r.u_n[g] = g < n_gas && sn >= 0 ? col[(size_t)g * n_seg + sn] : 0.0;
    same = same && fabs(r.u_f[g] - r.u_n[g]) <= 2e-15 * fabs(r.u_f[g]);
  }
  // The two segments of a shell on a symmetric path have the same column; the column integration leaves them a few
  // ulp apart in half of the cases (<= 4.4e-16 relative on the configs[3] rays): within 2e-15 they are taken as ONE
  // value, and the segments share their attenuation.
  if (same)
#pragma unroll
    for (int g = 0; g < 4; ++g) r.u_n[g] = r.u_f[g];
#pragma unroll
  for (int e = 0; e < kAdjEnt; ++e) {
    r.ent_p[e] = pl[4 + e];
    r.ent_gf[e] = pl[4 + kAdjEnt + e];
    r.dc_f[e] = e < r.n_ent && sf >= 0 ? col[(size_t)(n_gas + r.ent_p[e]) * n_seg + sf] : 0.0;
    r.dc_n[e] = e < r.n_ent && sn >= 0 ? col[(size_t)(n_gas + r.ent_p[e]) * n_seg + sn] : 0.0;
  }
  r.has = (sf >= 0 ? 1 : 0) | (sn >= 0 ? 2 : 0) | (same ? 4 : 0);
  out[i] = r;
}

// TWO: the two segments of a shell read different coefficient rows (3-D paths: a row per LOS step; one ray per thread)
template <int NG, bool LAYER, bool PAR, int NR, bool TWO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SR_FOLD_WAVES, SR_FOLD_WAVES))) void sr_limb_adjoint_fold_kernel(
    const double *__restrict__ abs_c, const double *__restrict__ emi_c, const double *__restrict__ dabs,
    const double *__restrict__ demi, int n_pts, int n_layers, int n_jrows, const FoldRec *__restrict__ rec, // [n_batches][n_visits][NR]
    const int *__restrict__ zero_off, const int *__restrict__ zero_row, int n_par, LimbOpts o, int n_visits, int n_rays,
    double *__restrict__ rad, double *__restrict__ jac_layer, double *__restrict__ jac_par) {
  static_assert(!TWO || NR == 1, "rows per ray: one ray per thread");
#if SR_FOLD_XCD
  int pb, batch; // all batches of a point block on one XCD, one after the other (limb_block)
  if (!limb_block((n_pts + 255) / 256, (n_rays + NR - 1) / NR, pb, batch)) return;
  const int j = pb * 256 + threadIdx.x;
#else
  const int j = blockIdx.x * blockDim.x + threadIdx.x, batch = blockIdx.y;
#endif
  if (j >= n_pts) return;
  const int ray0 = batch * NR;
  const FoldRec *rc = rec + (size_t)batch * n_visits * NR;
  const size_t gstride = (size_t)n_layers * n_pts;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int ray = ray0 + r;
    if (ray >= n_rays) continue;
    for (int q = zero_off[ray]; q < zero_off[ray + 1]; ++q) { // rows this ray never touches
      const int row = zero_row[q];
      if (row < n_jrows) {
        if (LAYER) jac_layer[((size_t)ray * n_jrows + row) * n_pts + j] = 0.0;
      } else if (PAR) {
        jac_par[((size_t)ray * n_par + (row - n_jrows)) * n_pts + j] = 0.0;
      }
    }
  }
  double If[NR], rem[NR], rem_lo[NR], Tn[NR], cs[NR], cs_lo[NR], Iobs[NR], Iobs_lo[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    If[r] = ray0 + r < n_rays ? limb_initial(o, rad, (size_t)(ray0 + r) * n_pts + j, j) : 0.0;
    rem[r] = rem_lo[r] = cs[r] = cs_lo[r] = 0.0;
    Tn[r] = 1.0;
  }
  auto two_sum_add = [](double &hi, double &lo, double x) {
    const double sm = hi + x, bb = sm - hi;
    lo += (hi - (sm - bb)) + (x - bb);
    hi = sm;
  };
  // sweep 1: the observed radiance and the path's optical depth
  for (int v = 0; v < n_visits; ++v) {
    const FoldRec *rv = rc + (size_t)v * NR;
    const size_t ofs = (size_t)rv[0].layer * n_pts + j;
    double a[NG], e[NG], a2[NG], e2[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      a[g] = abs_c[g * gstride + ofs];
      e[g] = emi_c[g * gstride + ofs];
      if (TWO) {
        const size_t ofs2 = (size_t)rv[0].layer_n * n_pts + j;
        a2[g] = abs_c[g * gstride + ofs2];
        e2[g] = emi_c[g * gstride + ofs2];
      } else {
        a2[g] = a[g];
        e2[g] = e[g];
      }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const FoldRec &R = rv[r];
      if (!R.has) continue; // wave-uniform
      Atten A;
      if (R.has & 1) {
        double tau = 0.0, E = 0.0;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          tau = g == 0 ? a[g] * R.u_f[g] : tau + a[g] * R.u_f[g];
          E = g == 0 ? e[g] * R.u_f[g] : E + e[g] * R.u_f[g];
        }
        two_sum_add(rem[r], rem_lo[r], tau);
        A = attenuation(tau);
        If[r] = If[r] * A.t + (o.solo_absorption ? 0.0 : E * A.f);
      }
      if (R.has & 2) {
        double tau = 0.0, E = 0.0;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          tau = g == 0 ? a2[g] * R.u_n[g] : tau + a2[g] * R.u_n[g];
          E = g == 0 ? e2[g] * R.u_n[g] : E + e2[g] * R.u_n[g];
        }
        two_sum_add(rem[r], rem_lo[r], tau);
        if (!(R.has & 4)) A = attenuation(tau);
        two_sum_add(cs[r], cs_lo[r], (o.solo_absorption ? 0.0 : E * Tn[r]) * A.f);
        Tn[r] *= A.t;
      }
    }
  }
  double slot[NR][4];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    // I_obs = I_f Tn + cs as an unevaluated sum of two doubles (exact product, two-sums): see sweep 2
    const double p_hi = If[r] * Tn[r], p_lo = fma(If[r], Tn[r], -p_hi);
    Iobs[r] = cs[r];
    Iobs_lo[r] = cs_lo[r] + p_lo;
    two_sum_add(Iobs[r], Iobs_lo[r], p_hi);
    If[r] = ray0 + r < n_rays ? limb_initial(o, rad, (size_t)(ray0 + r) * n_pts + j, j) : 0.0; // (rad is written at the very end)
    cs[r] = cs_lo[r] = 0.0;
    Tn[r] = 1.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) slot[r][q] = 0.0;
  }
  // sweep 2: the weights, shell by shell
  for (int v = 0; v < n_visits; ++v) {
    const FoldRec *rv = rc + (size_t)v * NR;
    const size_t ofs = (size_t)rv[0].layer * n_pts + j;
    double a[NG], e[NG], da[NG], de[NG], a2[NG], e2[NG], da2[NG], de2[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      a[g] = abs_c[g * gstride + ofs];
      e[g] = emi_c[g * gstride + ofs];
      if (LAYER) {
        da[g] = dabs[g * gstride + ofs];
        de[g] = demi[g * gstride + ofs];
      }
      if (TWO) {
        const size_t ofs2 = (size_t)rv[0].layer_n * n_pts + j;
        a2[g] = abs_c[g * gstride + ofs2];
        e2[g] = emi_c[g * gstride + ofs2];
        if (LAYER) {
          da2[g] = dabs[g * gstride + ofs2];
          de2[g] = demi[g * gstride + ofs2];
        }
      } else {
        a2[g] = a[g];
        e2[g] = e[g];
        if (LAYER) {
          da2[g] = da[g];
          de2[g] = de[g];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const FoldRec &R = rv[r];
      if (!R.has) continue;
      const int ray = ray0 + r;
      Atten A;
      double fp = 0.0, d = 0.0, wt_f = 0.0, we_f = 0.0, wt_n = 0.0, we_n = 0.0;
      if (R.has & 1) {
        double tau = 0.0, E = 0.0, dtau = 0.0, dE = 0.0;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const double u = R.u_f[g];
          tau = g == 0 ? a[g] * u : tau + a[g] * u;
          E = g == 0 ? e[g] * u : E + e[g] * u;
          if (LAYER) {
            dtau = fma(da[g], u, dtau);
            dE = fma(de[g], u, dE);
          }
        }
        A = attenuation(tau);
        fp = A.thin ? -0.5 : (tau * A.t - A.em1) * (A.rtau * A.rtau);
        two_sum_add(rem[r], rem_lo[r], -tau);
        const double Ta = exp_bounded(fmin(fmax(-(rem[r] + rem_lo[r]), -700.0), 700.0)); // everything behind: the inner far side and the near side
        wt_f = (o.solo_absorption ? -If[r] * A.t : fma(E, fp, -If[r] * A.t)) * Ta;
        we_f = o.solo_absorption ? 0.0 : A.f * Ta;
        if (LAYER) d = fma(wt_f, dtau, we_f * dE);
        If[r] = If[r] * A.t + (o.solo_absorption ? 0.0 : E * A.f);
      }
      if (R.has & 2) {
        double tau = 0.0, E = 0.0, dtau = 0.0, dE = 0.0;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const double u = R.u_n[g];
          tau = g == 0 ? a2[g] * u : tau + a2[g] * u;
          E = g == 0 ? e2[g] * u : E + e2[g] * u;
          if (LAYER) {
            dtau = fma(da2[g], u, dtau);
            dE = fma(de2[g], u, dE);
          }
        }
        if (!(R.has & 4)) {
          A = attenuation(tau);
          fp = A.thin ? -0.5 : (tau * A.t - A.em1) * (A.rtau * A.rtau);
        }
        const double ETn = o.solo_absorption ? 0.0 : E * Tn[r];
        // I_in t Tn = I_obs - cs.  Both are carried as error-free sums (two doubles) of the SAME rounded terms -- the
        // operations of sweep 1, bit for bit -- so the difference is the sum of the terms still to come, I_f Tn(all) and
        // the inner segments' E f Tn, to THEIR relative accuracy: in one double it was good to eps x I_obs only, which
        // at a point whose tangent shells are hidden behind tau ~ 25 left 4e-22 on a value of 1e-25 (configs[3]: 1e-9
        // of a row's largest value)
        two_sum_add(cs[r], cs_lo[r], ETn * A.f);
        const double X = (Iobs[r] - cs[r]) + (Iobs_lo[r] - cs_lo[r]);
        wt_n = fma(ETn, fp, -X); // E f' Tn - I_in t Tn
        we_n = o.solo_absorption ? 0.0 : A.f * Tn[r];
        if (LAYER) d += fma(wt_n, dtau, we_n * dE);
        Tn[r] *= A.t;
      }
      if (LAYER) jac_layer[((size_t)ray * n_jrows + R.jrow) * n_pts + j] = d;
      if (PAR) {
        for (int i = 0; i < R.n_ent; ++i) {
          const int gf = R.ent_gf[i], g = gf & 0xff, sl = (gf >> 8) & 0xff, fl = gf >> 16;
          double ag = a[0], eg = e[0], ag2 = a2[0], eg2 = e2[0];
#pragma unroll
          for (int q = 1; q < NG; ++q) {
            ag = g == q ? a[q] : ag;
            eg = g == q ? e[q] : eg;
            ag2 = g == q ? a2[q] : ag2;
            eg2 = g == q ? e2[q] : eg2;
          }
          double val = fma(fma(wt_f, ag, we_f * eg), R.dc_f[i], fma(wt_n, ag2, we_n * eg2) * R.dc_n[i]);
          if (!(fl & 1)) val += sl == 0 ? slot[r][0] : (sl == 1 ? slot[r][1] : (sl == 2 ? slot[r][2] : slot[r][3]));
          if (fl & 2) {
            double *out = jac_par + ((size_t)ray * n_par + R.ent_p[i]) * n_pts + j;
            if (fl & 4) *out = val; else *out += val;
          } else {
            if (sl == 0) slot[r][0] = val; else if (sl == 1) slot[r][1] = val; else if (sl == 2) slot[r][2] = val; else slot[r][3] = val;
          }
        }
      }
    }
  }
  if (rad) {
#pragma unroll
    for (int r = 0; r < NR; ++r)
      if (ray0 + r < n_rays) rad[(size_t)(ray0 + r) * n_pts + j] = Iobs[r] + Iobs_lo[r];
  }
}

// ------------------------------------------------------------------------
// The folded recursion for FEW, BROAD column parameters (the retrieval loop of configs[4]: 7 profile parameters whose
// masks cover the whole path): the path-order forward-sensitivity kernel carries four derivatives through the recursion
// and repeats it per block of four, segment by segment; here a shell's coefficients are loaded once for its two
// segments and every parameter (up to kFoldDensePar) is carried along in ONE sweep: sr_limb_fold_sens_lds_kernel below.
// ------------------------------------------------------------------------
// the instrument step's device scratch (launch_lowres): the bands' weight table and point ranges, then partial sums
struct LowresScratch {
  double *W;   // [n_bands][n_pts]
  int *range;  // [n_bands][2]
  double *Wt;  // [band tiles][n_pts][16]: the same weights, band-minor in tiles of 16 (zero columns beyond n_bands)
  double *part;
};
static LowresScratch lowres_layout(void *scratch, int n_pts, int n_bands) {
  LowresScratch L;
  L.W = static_cast<double *>(scratch);
  L.range = reinterpret_cast<int *>(L.W + (size_t)n_bands * n_pts); // (before the partial sums: their size follows n_rays)
  L.Wt = reinterpret_cast<double *>(L.range + 2 * (size_t)n_bands + 2);
  L.part = L.Wt + (size_t)((n_bands + 15) / 16) * 16 * n_pts;
  return L;
}
struct __attribute__((aligned(16))) FoldDense { // one ray in one shell
  int layer, has, pad0, pad1;                // has: as FoldRec
  double u_f[4], u_n[4];
  double dc_f[kFoldDensePar], dc_n[kFoldDensePar]; // d col / d x_p of the two segments
};
static_assert(sizeof(FoldDense) == 16 + 64 + 16 * kFoldDensePar, "FoldDense layout");
struct ParGas { int g[kFoldDensePar]; };

// plan: [n_rec][4] ints: layer, far segment, near segment (-1: none), 0
__global__ void sr_fold_dense_pack_kernel(const int *__restrict__ plan, const double *__restrict__ col, int n_gas, int n_par,
                                          int n_seg, int n_rec, FoldDense *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rec) return;
  const int *pl = plan + (size_t)i * 4;
  const int sf = pl[1], sn = pl[2];
  FoldDense r;
  r.layer = pl[0]; r.pad0 = r.pad1 = 0;
  bool same = sf >= 0 && sn >= 0;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    r.u_f[g] = g < n_gas && sf >= 0 ? col[(size_t)g * n_seg + sf] : 0.0;
    r.u_n[g] = g < n_gas && sn >= 0 ? col[(size_t)g * n_seg + sn] : 0.0;
    same = same && fabs(r.u_f[g] - r.u_n[g]) <= 2e-15 * fabs(r.u_f[g]);
  }
  // The two segments of a shell on a symmetric path have the same column; the column integration leaves them a few
  // ulp apart in half of the cases (<= 4.4e-16 relative on the configs[3] rays): within 2e-15 they are taken as ONE
  // value, and the segments share their attenuation.
  if (same)
#pragma unroll
    for (int g = 0; g < 4; ++g) r.u_n[g] = r.u_f[g];
#pragma unroll
  for (int p = 0; p < kFoldDensePar; ++p) {
    r.dc_f[p] = p < n_par && sf >= 0 ? col[(size_t)(n_gas + p) * n_seg + sf] : 0.0;
    r.dc_n[p] = p < n_par && sn >= 0 ? col[(size_t)(n_gas + p) * n_seg + sn] : 0.0;
  }
  r.has = (sf >= 0 ? 1 : 0) | (sn >= 0 ? 2 : 0) | (same ? 4 : 0);
  out[i] = r;
}

// ONE sweep (round 5): forward sensitivities in fold order.  The far side carries dI_f / dx_p through the recursion
// (dI_f' = dI_f t + (w_tau a_g + w_E e_g) dc_far,p); the near side is visited from the observer inwards, so the
// transmission T_n between a segment and the observer is known when the segment is reached, and its logarithmic
// derivative D_p = sum a_g dc_near,p over the segments already passed is carried beside it:
//   d cs_p += T_n (f e_g + E f' a_g) dc_near,p - E f T_n D_p,   D_p += a_g dc_near,p,
//   d I_obs / d x_p = T_n (dI_f,p - I_f D_p) + d cs_p           (I_obs = I_f T_n + cs).
// Round 4's kernel ran two sweeps with one accumulator per parameter (sr_limb_adjoint_fold_kernel's quantities: 10 023
// VALU instructions per wave on configs[4] -- an attenuation() per shell and sweep, the exponential of the remaining
// optical depth, four error-free sums); here three accumulators per parameter, 4 300-5 000 instructions, and no
// difference of nearly equal sums anywhere: every term is a product of the quantities the radiance itself is made of.
// A parameter's gas is wave-uniform: the choice between the gases' weights is a scalar branch (the asm statements keep
// the compiler from turning it into per-lane selects, two v_cndmask per double).  profiles/r05_fold_sens_ab.txt.

// attenuation() with the polynomial's coefficients in SGPRs (the same operations, bit for bit): for a kernel whose
// registers are its accumulators (20 VGPRs of constants otherwise)
__device__ inline Atten attenuation_sc(double tau) {
  const double x = -tau;
  const double n = rint(x * 0x1.71547652b82fep+0);
  double r = fma(-n, 0x1.62e42fefa39efp-1, x);
  r = fma(-n, 0x1.abc9e3b39803fp-56, r);
  double p = fma3vs(r, 0x1.ade156a5dcb37p-26, 0x1.28af3fca7ab0cp-22);
  p = fma3s(r, p, 0x1.71dee623fde64p-19);
  p = fma3s(r, p, 0x1.a01997c89e6b0p-16);
  p = fma3s(r, p, 0x1.a01a014761f6ep-13);
  p = fma3s(r, p, 0x1.6c16c1852b7b0p-10);
  p = fma3s(r, p, 0x1.1111111122322p-7);
  p = fma3s(r, p, 0x1.55555555502a1p-5);
  p = fma3s(r, p, 0x1.5555555555511p-3);
  p = fma3s(r, p, 0x1.000000000000bp-1);
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

// The one-sweep kernel with the ray's records in LDS.  With the records read by scalar loads (the first one-sweep version)
// a wave waits for its 208-byte record (the scalar cache holds 16 KB, the rays resident on a CU stream 50 KB through
// it), THEN for the coefficients whose address the record gives, then computes: 46 % issue-busy at four waves per SIMD.
// Here a block copies its ray's records to LDS in chunks of kSensChunk shells; headers are read one shell ahead, so
// the next shell's coefficients are in flight during this shell's arithmetic, and the column derivatives arrive
// as LDS broadcasts under the attenuation.
constexpr int kSensChunk = 64;
constexpr int kSensRecD = 26; // doubles per FoldDense
static_assert(sizeof(FoldDense) == kSensRecD * 8, "FoldDense in doubles");
// BANDS (round 6): the instrument bands in the epilogue.  A retrieval iteration wants the band integrals of the block's
// eight or nine spectra (radiance + derivatives), not the spectra: S[q][b] = sum_j val_q(j) W_b(j) over the block's 256
// points is a [9 x 256] x [256 x n_bands] product -- a wave's values go through LDS and it multiplies its own 64 points
// with v_mfma_f64_16x16x4 (A = the values, rows padded to 16; B = a 16-band tile of the band-minor weight table) and
// stores ONE partial sum per (spectrum, band) and wave -- the 69 MB of spectra a configs[4] iteration wrote and
// sr_lowres_apply_kernel read again (57 us of a 350 us iteration) never exist.  (First version: per band a product per
// value and a lane reduction, 26 ds_bpermute each; configs[4]'s 14 bands all cover its whole grid: +27 us on the kernel's
// 260.)  part: [n_rays (1 + n_par)][64-point slots][band tiles][16], rows as sr_retrieval_forward_dev orders them;
// sr_lowres_sum_blocks_kernel adds a band's slots.
struct FoldBands {
  const double *Wt;  // [band tiles][n_pts][16]
  const int *range;  // [n_bands][2]
  double *part;
  int n_bands;
};
constexpr int kBandVals = kFoldDensePar + 1;  // spectra per ray: the radiance and its derivatives
constexpr int kBandRow = 68;                  // doubles per row of a wave's tile of values (64 + padding: rows 8 banks apart)
template <int NG, bool BANDS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NG <= 2 ? 4 : 3, NG <= 2 ? 4 : 3))) void sr_limb_fold_sens_lds_kernel(
    const double *__restrict__ abs_c, const double *__restrict__ emi_c, int n_pts, int n_layers,
    const FoldDense *__restrict__ rec, // [n_rays][n_visits]
    int n_par, ParGas pg, LimbOpts o, int n_visits, int n_rays, double *__restrict__ rad, double *__restrict__ jac_par,
    FoldBands bd) {
  // the ray's records + a row of zeros; BANDS: + every wave's tile of values
  constexpr int kRecDoubles = kSensChunk * kSensRecD + kFoldDensePar;
  __shared__ double lrec[kRecDoubles + (BANDS ? 4 * kBandVals * kBandRow : 0)];
  int pb, ray; // all rays of a point block on one XCD, one after the other
  if (!limb_block((n_pts + 255) / 256, n_rays, pb, ray)) return; // (block-uniform)
  const int j0 = pb * 256 + threadIdx.x;
  const bool live = j0 < n_pts;
  const int j = live ? j0 : n_pts - 1; // (the block's barriers need every thread)
  const double *src = reinterpret_cast<const double *>(rec + (size_t)ray * n_visits);
  const size_t gstride = (size_t)n_layers * n_pts;
  double If = limb_initial(o, rad, (size_t)ray * n_pts + j, j), Tn = 1.0, cs = 0.0;
  double dIf[kFoldDensePar], dcs[kFoldDensePar], Dn[kFoldDensePar];
  int gmask = 0;
#pragma unroll
  for (int p = 0; p < kFoldDensePar; ++p) {
    dIf[p] = dcs[p] = Dn[p] = 0.0;
    gmask |= (pg.g[p] & 3) << (2 * p);
  }
  const bool solo = o.solo_absorption != 0;
  for (int v0 = 0; v0 < n_visits; v0 += kSensChunk) {
    const int nv = min(kSensChunk, n_visits - v0);
    __syncthreads(); // (the previous chunk has been consumed)
    for (int i = threadIdx.x; i < nv * kSensRecD; i += 256) lrec[i] = src[(size_t)v0 * kSensRecD + i];
    if (threadIdx.x < kFoldDensePar) lrec[kSensChunk * kSensRecD + threadIdx.x] = 0.0;
    __syncthreads();
    double an[NG], en[NG];
    int has_n;
    auto fetch = [&](int v) { // header of shell v of the chunk; its coefficients requested
      has_n = 0;
      if (v < nv) {
        const int2 hd = *reinterpret_cast<const int2 *>(lrec + v * kSensRecD);
        has_n = __builtin_amdgcn_readfirstlane(hd.y);
        const int layer = __builtin_amdgcn_readfirstlane(hd.x);
        if (has_n) {
          const size_t ofs = (size_t)layer * n_pts + j;
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            an[g] = abs_c[g * gstride + ofs];
            en[g] = emi_c[g * gstride + ofs];
          }
        }
      }
    };
    fetch(0);
    for (int v = 0; v < nv; ++v) {
      const int has = has_n;
      double a[NG], e[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        a[g] = an[g];
        e[g] = en[g];
      }
      fetch(v + 1);
      if (!has) continue;
      const double *lr = lrec + v * kSensRecD;
      // one pass for a shell whose two segments have the same columns (has & 4: every shell of a 1-D limb path) or one
      // segment; otherwise the far segment, then the near one
      const int n_pass = (has & 7) == 3 ? 2 : 1;
      for (int k = 0; k < n_pass; ++k) {
        const bool far = (has & 1) && k == 0, near = (has & 2) && (k == 1 || n_pass == 1);
        const double *u = lr + (far ? 2 : 6);
        const double *lcf = lr + 10, *lcn = near ? lr + 18 : lrec + kSensChunk * kSensRecD; // (no near segment: zeros)
        // (LDS broadcasts: the first half of the parameters in flight under the attenuation, the second half under the
        // first half's steps -- all sixteen doubles at the top were 140 VGPRs, three waves per SIMD)
        constexpr int kH = kFoldDensePar / 2;
        double cf[kFoldDensePar], cn[kFoldDensePar];
#pragma unroll
        for (int p = 0; p < kH; ++p) {
          cf[p] = lcf[p];
          cn[p] = lcn[p];
        }
        double tau = a[0] * u[0], E = e[0] * u[0];
#pragma unroll
        for (int g = 1; g < NG; ++g) {
          tau = tau + a[g] * u[g];
          E = E + e[g] * u[g];
        }
        if (solo) E = 0.0;
        const Atten A = attenuation_sc(tau);
        const double fp = A.thin ? -0.5 : (tau * A.t - A.em1) * (A.rtau * A.rtau);
        const double Ef = E * A.f, we = solo ? 0.0 : A.f;
        double t_f = 1.0, nEfTn = 0.0, bf[NG], wn[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) bf[g] = wn[g] = 0.0;
        if (far) {
          const double wt = fma(E, fp, -If * A.t);
#pragma unroll
          for (int g = 0; g < NG; ++g) bf[g] = fma(wt, a[g], we * e[g]);
          If = If * A.t + Ef;
          t_f = A.t;
        }
        if (near) {
          const double wtn = (E * Tn) * fp, wen = we * Tn;
#pragma unroll
          for (int g = 0; g < NG; ++g) wn[g] = fma(wtn, a[g], wen * e[g]);
          cs = fma(E * Tn, A.f, cs); // (sr_limb_fold_fwd_kernel's operations: the two kernels' radiances are the same doubles)
          nEfTn = -(Ef * Tn);
          Tn *= A.t;
        }
        // The parameters' gases as a bit field, made opaque per pass (hoisted out of the loop as 16 compare results they
        // cost 32 SGPRs and spill), and gas by gas, so that every (gas, parameter) step is its own block behind a scalar
        // branch (one chain over the gases per parameter was tail-merged into moves of the chosen weights)
        int gm = gmask, np = n_par;
        asm volatile("" : "+s"(gm), "+s"(np) : : "memory");
#pragma unroll
        for (int p = kH; p < kFoldDensePar; ++p) {
          cf[p] = lcf[p];
          cn[p] = lcn[p];
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int q = 0; q < NG; ++q) {
#pragma unroll
            for (int p = h * kH; p < (h + 1) * kH; ++p) {
              // (Round 6: a two-instruction step for the parameters whose column derivatives are both zero in a shell -- a
              // bit mask in the record header, a scalar branch per parameter -- was built: the second code path per
              // parameter cost 44 VGPR spills at this kernel's 128 registers, 0.26 -> 0.81 ms.  Not kept.)
              if (p < np && (NG == 1 || ((gm >> (2 * p)) & 3) == q)) {
                asm volatile("");
                dIf[p] = fma3(dIf[p], t_f, bf[q] * cf[p]);
                dcs[p] = fma(wn[q], cn[p], dcs[p]);
                dcs[p] = fma(nEfTn, Dn[p], dcs[p]);
                Dn[p] = fma(a[q], cn[p], Dn[p]);
              }
            }
          }
        }
      }
    }
  }
  if constexpr (BANDS) {
    // Wave by wave, no block barrier: a wave's values go through ITS rows of the tile, its 9 x 16 sums to its own slot of
    // `part`.  configs[4] (variant builds that leave parts out): main loop 259 us, + 23 here -- the sixteen MFMAs 9 (their
    // flops at the fp64 rate: rows and bands padded to 16), the stores 5, LDS 3; the weight loads nothing.
    constexpr int kV = kBandVals;
    const int c_lo = pb * 256, n_slots = 4 * ((n_pts + 255) / 256);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lo = lane & 15, kq = lane >> 4;
    const int n_tiles = (bd.n_bands + 15) >> 4;
    double *vt = lrec + kRecDoubles + wave * (kV * kBandRow);
    vt[lane] = live ? fma(If, Tn, cs) : 0.0;
#pragma unroll
    for (int p = 0; p < kFoldDensePar; ++p)
      vt[(1 + p) * kBandRow + lane] = live && p < n_par ? fma(Tn, fma(-If, Dn[p], dIf[p]), dcs[p]) : 0.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // A[row = lane & 15][k = lane >> 4], B[k][col = lane & 15], D[row = (lane >> 4) + 4 r][col]
    const double *va = vt + min(lo, kV - 1) * kBandRow + kq;
    double A[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) A[ks] = lo < kV ? va[4 * ks] : 0.0;
    for (int tile = 0; tile < n_tiles; ++tile) {
      const double *wt = bd.Wt + (size_t)tile * n_pts * 16 + lo;
      double Bv[16]; // a tile's sixteen B operands requested together
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) Bv[ks] = wt[(size_t)min(c_lo + 64 * wave + 4 * ks + kq, n_pts - 1) * 16]; // (beyond the grid A is zero)
      v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[ks], Bv[ks], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = kq + 4 * r;
        if (q <= n_par) {
          const size_t row = q == 0 ? (size_t)ray : (size_t)n_rays + (size_t)ray * n_par + (q - 1);
          bd.part[((row * n_slots + 4 * pb + wave) * n_tiles + tile) * 16 + lo] = acc[r];
        }
      }
    }
    return;
  }
  if (!live) return;
#pragma unroll
  for (int p = 0; p < kFoldDensePar; ++p)
    if (p < n_par) jac_par[((size_t)ray * n_par + p) * n_pts + j] = fma(Tn, fma(-If, Dn[p], dIf[p]), dcs[p]);
  rad[(size_t)ray * n_pts + j] = fma(If, Tn, cs);
}

// The radiances alone, folded (ray batches; BASELINE configs[2]: 64 rays): one sweep over the shells, a shell's coefficients
// loaded once for the ray's two segments and -- the path being symmetric -- ONE attenuation() for both:
// I_obs = I_f(tangent) Tn(all) + sum E f Tn over the near side (sr_limb_adjoint_fold_kernel's sweep 1).
template <int NG>
__global__ __launch_bounds__(256) void sr_limb_fold_fwd_kernel(const double *__restrict__ abs_c, const double *__restrict__ emi_c,
                                                               int n_pts, int n_layers, const FoldDense *__restrict__ rec,
                                                               LimbOpts o, int n_visits, int n_rays, double *__restrict__ rad) {
  int pb, ray;
  if (!limb_block((n_pts + 255) / 256, n_rays, pb, ray)) return;
  const int j = pb * 256 + threadIdx.x;
  if (j >= n_pts) return;
  const FoldDense *rc = rec + (size_t)ray * n_visits;
  const size_t gstride = (size_t)n_layers * n_pts;
  double If = limb_initial(o, rad, (size_t)ray * n_pts + j, j), Tn = 1.0, cs = 0.0;
  constexpr int kB = NG == 1 ? 4 : 2; // visits whose loads are issued together
  for (int vb = 0; vb < n_visits; vb += kB) {
    double a[kB][NG], e[kB][NG];
#pragma unroll
    for (int t = 0; t < kB; ++t) {
      const FoldDense &Rl = rc[min(vb + t, n_visits - 1)];
      const bool on = Rl.has != 0; // wave-uniform: a shell this ray does not cross costs no loads
      const size_t ofs = (size_t)Rl.layer * n_pts + j;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        a[t][g] = on ? abs_c[g * gstride + ofs] : 0.0;
        e[t][g] = on ? emi_c[g * gstride + ofs] : 0.0;
      }
    }
#pragma unroll
    for (int t = 0; t < kB; ++t) {
      if (vb + t >= n_visits) break;
      const FoldDense &R = rc[vb + t];
      if (!R.has) continue; // wave-uniform
      Atten A;
      if (R.has & 1) {
        double tau = a[t][0] * R.u_f[0], E = e[t][0] * R.u_f[0];
#pragma unroll
        for (int g = 1; g < NG; ++g) {
          tau = tau + a[t][g] * R.u_f[g];
          E = E + e[t][g] * R.u_f[g];
        }
        A = attenuation(tau);
        If = If * A.t + (o.solo_absorption ? 0.0 : E * A.f);
        if (R.has & 4) { // the near-side segment: same columns, same tau and E
          cs = fma(o.solo_absorption ? 0.0 : E * Tn, A.f, cs);
          Tn *= A.t;
        }
      }
      if ((R.has & 6) == 2) {
        double tau = a[t][0] * R.u_n[0], E = e[t][0] * R.u_n[0];
#pragma unroll
        for (int g = 1; g < NG; ++g) {
          tau = tau + a[t][g] * R.u_n[g];
          E = E + e[t][g] * R.u_n[g];
        }
        A = attenuation(tau);
        cs = fma(o.solo_absorption ? 0.0 : E * Tn, A.f, cs);
        Tn *= A.t;
      }
    }
  }
  rad[(size_t)ray * n_pts + j] = fma(If, Tn, cs);
}

int launch_fold_fwd(const int *plan, const double *col, int n_seg, int n_rec, FoldDense *rec, const double *abs_c,
                    const double *emi_c, int n_pts, int n_layers, int n_rays, int n_visits, const LimbOpts &o, double *rad,
                    hipStream_t st, bool pack) {
  if (n_rec <= 0 || n_rays <= 0) return 0;
  if (pack)
    hipLaunchKernelGGL(sr_fold_dense_pack_kernel, dim3((n_rec + 63) / 64), dim3(64), 0, st, plan, col, o.n_gas, 0, n_seg, n_rec, rec);
  if (n_pts <= 0) return (int)hipGetLastError(); // (packing only)
  const dim3 grid(limb_grid((n_pts + 255) / 256, n_rays));
#define SR_FF(NG) hipLaunchKernelGGL(sr_limb_fold_fwd_kernel<NG>, grid, dim3(256), 0, st, abs_c, emi_c, n_pts, n_layers, rec, o, n_visits, \
                                     n_rays, rad)
  switch (o.n_gas) { case 1: SR_FF(1); break; case 2: SR_FF(2); break; case 3: SR_FF(3); break; default: SR_FF(4); break; }
#undef SR_FF
  return (int)hipGetLastError();
}

size_t fold_dense_bytes(int n_rec) { return sizeof(FoldDense) * (size_t)n_rec; }

int launch_fold_dense(const int *plan, const double *col, const int *par_gas_host, int n_par, int n_seg, int n_rec, FoldDense *rec,
                      const double *abs_c, const double *emi_c, int n_pts, int n_layers, int n_rays, int n_visits,
                      const LimbOpts &o, double *rad, double *jac_par, hipStream_t st, const void *lowres_scratch, int n_bands) {
  if (n_rec <= 0 || n_pts <= 0 || n_rays <= 0 || n_par <= 0 || n_par > kFoldDensePar) return 0;
  hipLaunchKernelGGL(sr_fold_dense_pack_kernel, dim3((n_rec + 63) / 64), dim3(64), 0, st, plan, col, o.n_gas, n_par, n_seg, n_rec, rec);
  ParGas pg;
  for (int p = 0; p < kFoldDensePar; ++p) pg.g[p] = p < n_par ? par_gas_host[p] : 0;
  const dim3 grid(limb_grid((n_pts + 255) / 256, n_rays));
  FoldBands bd{nullptr, nullptr, nullptr, 0};
  if (lowres_scratch) {
    const LowresScratch L = lowres_layout(const_cast<void *>(lowres_scratch), n_pts, n_bands);
    bd = FoldBands{L.Wt, L.range, L.part, n_bands};
  }
#define SR_FD(NG)                                                                                                                  \
  do {                                                                                                                             \
    if (lowres_scratch)                                                                                                            \
      hipLaunchKernelGGL((sr_limb_fold_sens_lds_kernel<NG, true>), grid, dim3(256), 0, st, abs_c, emi_c, n_pts, n_layers, rec, n_par, \
                         pg, o, n_visits, n_rays, rad, jac_par, bd);                                                               \
    else                                                                                                                           \
      hipLaunchKernelGGL((sr_limb_fold_sens_lds_kernel<NG, false>), grid, dim3(256), 0, st, abs_c, emi_c, n_pts, n_layers, rec, n_par, \
                         pg, o, n_visits, n_rays, rad, jac_par, bd);                                                               \
  } while (0)
  switch (o.n_gas) { case 1: SR_FD(1); break; case 2: SR_FD(2); break; case 3: SR_FD(3); break; default: SR_FD(4); break; }
#undef SR_FD
  return (int)hipGetLastError();
}

size_t fold_rec_bytes(int n_rec) { return sizeof(FoldRec) * (size_t)n_rec; }

int launch_fold_pack(const int *plan, const double *col, int n_gas, int n_seg, int n_rec, FoldRec *out, hipStream_t st) {
  static_assert(kFoldPlanInts == 4 + 2 * kAdjEnt + 2, "fold plan layout");
  if (n_rec <= 0) return 0;
  hipLaunchKernelGGL(sr_fold_pack_kernel, dim3((n_rec + 63) / 64), dim3(64), 0, st, plan, col, n_gas, n_seg, n_rec, out);
  return (int)hipGetLastError();
}

int launch_limb_adjoint_fold(const double *abs_c, const double *emi_c, const double *dabs, const double *demi, int n_pts,
                             int n_layers, int n_jrows, int two_rows, int n_rays, const FoldRec *rec, const int *zero_off,
                             const int *zero_row, int n_par, const LimbOpts &o, int n_visits, double *rad, double *jac_layer,
                             double *jac_par, hipStream_t st) {
  if (n_pts <= 0 || n_rays <= 0 || n_visits <= 0) return 0;
  if (two_rows) { // one ray per thread, a coefficient row per segment
#if SR_FOLD_XCD
    const dim3 grid2(limb_grid((n_pts + 255) / 256, n_rays));
#else
    const dim3 grid2((n_pts + 255) / 256, n_rays);
#endif
#define SR_AF2(NG, L, P) hipLaunchKernelGGL((sr_limb_adjoint_fold_kernel<NG, L, P, 1, true>), grid2, dim3(256), 0, st, abs_c, emi_c, \
                                            dabs, demi, n_pts, n_layers, n_jrows, rec, zero_off, zero_row, n_par, o, n_visits, n_rays, \
                                            rad, jac_layer, jac_par)
#define SR_AF23(NG)                                                    \
  do {                                                                 \
    if (jac_layer && jac_par) SR_AF2(NG, true, true);                  \
    else if (jac_layer) SR_AF2(NG, true, false);                       \
    else SR_AF2(NG, false, true);                                      \
  } while (0)
    switch (o.n_gas) { case 1: SR_AF23(1); break; case 2: SR_AF23(2); break; case 3: SR_AF23(3); break; default: SR_AF23(4); break; }
#undef SR_AF23
#undef SR_AF2
    return (int)hipGetLastError();
  }
#if SR_FOLD_XCD
  const dim3 grid(limb_grid((n_pts + 255) / 256, (n_rays + kAdjFoldRays - 1) / kAdjFoldRays));
#else
  const dim3 grid((n_pts + 255) / 256, (n_rays + kAdjFoldRays - 1) / kAdjFoldRays);
#endif
#define SR_AF(NG, L, P) hipLaunchKernelGGL((sr_limb_adjoint_fold_kernel<NG, L, P, kAdjFoldRays, false>), grid, dim3(256), 0, st, abs_c, emi_c, \
                                           dabs, demi, n_pts, n_layers, n_jrows, rec, zero_off, zero_row, n_par, o, n_visits, n_rays, rad, \
                                           jac_layer, jac_par)
#define SR_AF3(NG)                                                     \
  do {                                                                 \
    if (jac_layer && jac_par) SR_AF(NG, true, true);                   \
    else if (jac_layer) SR_AF(NG, true, false);                        \
    else SR_AF(NG, false, true);                                       \
  } while (0)
  switch (o.n_gas) { case 1: SR_AF3(1); break; case 2: SR_AF3(2); break; case 3: SR_AF3(3); break; default: SR_AF3(4); break; }
#undef SR_AF3
#undef SR_AF
  return (int)hipGetLastError();
}

size_t adj_prog_bytes(int n_seg) { return sizeof(SegProg) * (size_t)n_seg; }
static_assert(kAdjPlanInts == 4 + 2 * kAdjEnt, "host plan layout");

int launch_adj_pack(const int *plan, const double *col, int n_gas, int n_seg, SegProg *out, hipStream_t st) {
  if (n_seg <= 0) return 0;
  hipLaunchKernelGGL(sr_adj_pack_kernel, dim3((n_seg + 63) / 64), dim3(64), 0, st, plan, col, n_gas, n_seg, out);
  return (int)hipGetLastError();
}

int launch_limb_adjoint(const double *abs_c, const double *emi_c, const double *dabs, const double *demi, int n_pts,
                        int n_layers, int n_jrows, int n_rays, const int *seg_off, const SegProg *prog, const int *zero_off,
                        const int *zero_row, int n_par, const LimbOpts &o, double *rad, double *jac_layer,
                        double *jac_par, hipStream_t st) {
  if (n_pts <= 0 || n_rays <= 0) return 0;
  const dim3 grid((n_pts + 255) / 256, n_rays);
#define SR_A(NG, L, P) hipLaunchKernelGGL((sr_limb_adjoint_kernel<NG, L, P>), grid, dim3(256), 0, st, abs_c, emi_c, dabs, demi, \
                                          n_pts, n_layers, n_jrows, seg_off, prog, zero_off, zero_row, n_par, o, rad, jac_layer, jac_par)
#define SR_A3(NG)                                                      \
  do {                                                                 \
    if (jac_layer && jac_par) SR_A(NG, true, true);                    \
    else if (jac_layer) SR_A(NG, true, false);                         \
    else SR_A(NG, false, true);                                        \
  } while (0)
  switch (o.n_gas) { case 1: SR_A3(1); break; case 2: SR_A3(2); break; case 3: SR_A3(3); break; default: SR_A3(4); break; }
#undef SR_A3
#undef SR_A
  return (int)hipGetLastError();
}

#define SR_BY_NGAS(NGV, CALL1, CALL2, CALL3, CALL4) \
  switch (NGV) { case 1: CALL1; break; case 2: CALL2; break; case 3: CALL3; break; default: CALL4; break; }

int launch_limb(const double *abs_c, const double *emi_c, int n_pts, int n_layers, int n_rays, const int *seg_off,
                const int *seg_layer, const double *col, const LimbOpts &o, double *rad, hipStream_t st) {
  if (n_pts <= 0 || n_rays <= 0) return 0;
  // fewer than two waves per SIMD: the latency-bound variant (a function of the launch shape only)
  if (limb_launch_is_small(n_pts, n_rays)) {
    const dim3 gs(limb_grid((n_pts + 63) / 64, n_rays));
#define SR_LS(NG) hipLaunchKernelGGL(sr_limb_split_kernel<NG>, gs, dim3(64 * kLimbParts), 0, st, abs_c, emi_c, n_pts, n_layers, \
                                     seg_off, seg_layer, col, o, n_rays, rad)
    SR_BY_NGAS(o.n_gas, SR_LS(1), SR_LS(2), SR_LS(3), SR_LS(4))
#undef SR_LS
    return (int)hipGetLastError();
  }
  const dim3 grid(limb_grid((n_pts + 255) / 256, n_rays));
#define SR_L(NG) hipLaunchKernelGGL(sr_limb_kernel<NG>, grid, dim3(256), 0, st, abs_c, emi_c, n_pts, n_layers, seg_off, \
                                    seg_layer, col, o, n_rays, rad)
  SR_BY_NGAS(o.n_gas, SR_L(1), SR_L(2), SR_L(3), SR_L(4))
#undef SR_L
  return (int)hipGetLastError();
}

int launch_limb_jac(const double *abs_c, const double *emi_c, int n_pts, int n_layers, int n_rays, const int *seg_off,
                    const int *seg_layer, const double *col, const double *dcol, const int *par_gas, int n_par,
                    const LimbOpts &o, double *rad, double *jac, hipStream_t st) {
  if (n_pts <= 0 || n_rays <= 0 || n_par <= 0) return 0;
  // Every block of NP parameters repeats the recursion (exp, expm1, the coefficient loads): with many parameters
  // (configs[3]: one per layer) 16 per thread instead of 4 cut the kernel's instructions 3.4x (160 segments x
  // (60 + NP) per block of NP).
#define SR_L(NG, NP) hipLaunchKernelGGL((sr_limb_jac_kernel<NG, NP>), dim3((n_pts + 255) / 256, n_rays, (n_par + NP - 1) / NP), \
                                        dim3(256), 0, st, abs_c, emi_c, n_pts, n_layers, seg_off, seg_layer, col, dcol,      \
                                        par_gas, n_par, o, rad, jac)
  if (n_par > 8) {
    SR_BY_NGAS(o.n_gas, SR_L(1, 16), SR_L(2, 16), SR_L(3, 16), SR_L(4, 16))
  } else {
    SR_BY_NGAS(o.n_gas, SR_L(1, 4), SR_L(2, 4), SR_L(3, 4), SR_L(4, 4))
  }
#undef SR_L
  return (int)hipGetLastError();
}

int launch_limb_jac_layer(int forward, const double *abs_c, const double *emi_c, const double *dabs, const double *demi, int n_pts,
                          int n_layers, int n_rays, const int *seg_off, const int *seg_layer, const double *col,
                          const LimbOpts &o, double *jac, hipStream_t st) {
  if (n_pts <= 0 || n_rays <= 0 || n_layers <= 0) return 0;
#define SR_L(NG, NP) hipLaunchKernelGGL((sr_limb_jac_layer_kernel<NG, NP>), dim3((n_pts + 255) / 256, n_rays, (n_layers + NP - 1) / NP), \
                                        dim3(256), 0, st, abs_c, emi_c, dabs, demi, n_pts, n_layers, seg_off, seg_layer, col, o, jac)
  (void)forward; // the one-pass formulation is sr_limb_adjoint_kernel (launch_limb_adjoint)
  if (n_layers > 8) {
    SR_BY_NGAS(o.n_gas, SR_L(1, 16), SR_L(2, 16), SR_L(3, 16), SR_L(4, 16))
  } else {
    SR_BY_NGAS(o.n_gas, SR_L(1, 4), SR_L(2, 4), SR_L(3, 4), SR_L(4, 4))
  }
#undef SR_L
  return (int)hipGetLastError();
}
#undef SR_BY_NGAS

int launch_radiance_jac_layer(const double *abs_c, const double *emi_c, const double *dabs, const double *demi,
                              int n_pts, int n_layers, int n_rays, const int *seg_off, const int *seg_layer,
                              const double *seg_col, double *jac, hipStream_t st) {
  if (n_pts <= 0 || n_rays <= 0 || n_layers <= 0) return 0;
  constexpr int NP = 4;
  dim3 grid((n_pts + 255) / 256, n_rays, (n_layers + NP - 1) / NP);
  hipLaunchKernelGGL(sr_radiance_jac_layer_kernel<NP>, grid, dim3(256), 0, st, abs_c, emi_c, dabs, demi, n_pts,
                     n_layers, seg_off, seg_layer, seg_col, jac);
  return (int)hipGetLastError();
}

int launch_radiance_jac(const double *abs_c, const double *emi_c, int n_pts, int n_rays, const int *seg_off,
                        const int *seg_layer, const double *seg_col, const double *dcol, int n_par, double *rad,
                        double *jac, hipStream_t st) {
  if (n_pts <= 0 || n_rays <= 0 || n_par <= 0) return 0;
  constexpr int NP = 4;
  dim3 grid((n_pts + 255) / 256, n_rays, (n_par + NP - 1) / NP);
  hipLaunchKernelGGL(sr_radiance_jac_kernel<NP>, grid, dim3(256), 0, st, abs_c, emi_c, n_pts, n_rays, seg_off,
                     seg_layer, seg_col, dcol, n_par, rad, jac);
  return (int)hipGetLastError();
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
  r1_set(r, B.ry);
  r.wabs = r.wemi = 1.0; r.j1 = 0;
  r.ilir = (uint32_t)B.il | ((uint32_t)B.ir << 16);
  const ColdFull c = expand_cold(make_cold(B, dwp, x0, xf));
  for (int k = blockIdx.x * blockDim.x + threadIdx.x + 1; k <= n; k += gridDim.x * blockDim.x)
    y[i1 - 1 + k - 1] = humliv_point(k, r, c, xf);
}

// lineshape.f:272-442: x0 at or beyond an end of x(i1..i2).  These two branches advance rx and the
// region starts by running sums from the near end, so they are inherently sequential; no caller on
// the hot path reaches them, one thread walks the Fortran's loops as written.
__global__ void sr_humliv_outer_kernel(const double *__restrict__ x, int i1, int i2, double x0, double lw,
                                       double dw, double *__restrict__ y) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
#define SR_X(k) x[(k)-1]
#define SR_Y(k) y[(k)-1]
  const double ry = lw / dw, xstep = (SR_X(i1 + 1) - SR_X(i1)) / dw, ryf = (double)(float)ry;
  double q2[8], a1, b1, c1, d1;
  region2_coef(ry, q2);
  region1_coef(ry, a1, b1, c1, d1);
  auto r1 = [&](double xr) { const double x2 = xr * xr; return (a1 + x2 * b1) / (c1 + x2 * (d1 + 4. * x2)); };
  auto r2 = [&](double xr) {
    const double x2 = xr * xr;
    return (q2[0] + x2 * (q2[1] + x2 * (q2[2] + q2[3] * x2))) / (q2[4] + x2 * (q2[5] + x2 * (q2[6] + x2 * (q2[7] + x2))));
  };
  if (x0 <= SR_X(i1)) { // :272-357
    int j = i1;
    double rx = (SR_X(j) - x0) / dw;
    while ((rx + ry < 5.5) && (j <= i2)) {
      SR_Y(j) = core_point(rx, ry, ryf);
      j = j + 1;
      rx = rx + xstep;
    }
    if (j <= i2) {
      int l = max((int)round((15.0 - ry - rx) / xstep), 0) + j;
      l = min(l, i2);
      if (l > j) {
        double xrun = (SR_X(j) - x0) / dw;
        for (int k = j; k <= l; ++k) { SR_Y(k) = r2(xrun); xrun = xrun + xstep; }
        l = l + 1;
      }
      if (l < j) l = j;
      if (l < i2) {
        double xrun = (SR_X(l) - x0) / dw;
        for (int k = l; k <= i2; ++k) { SR_Y(k) = r1(xrun); xrun = xrun + xstep; }
      }
    }
  } else { // x0 >= x(i2), :358-442
    int j = i2;
    double rx = (x0 - SR_X(j)) / dw;
    while ((rx + ry < 5.5) && (j >= i1)) {
      SR_Y(j) = core_point(rx, ry, ryf);
      j = j - 1;
      rx = rx + xstep;
    }
    if (j >= i1) {
      int l = j - max((int)round((15.0 - ry - rx) / dw / xstep), 0); // sic, :404
      l = max(l, i1);
      if (l == i2) l = i2 + 1;
      if (l < j) {
        double xrun = (x0 - SR_X(l)) / dw;
        for (int k = l; k <= j; ++k) { SR_Y(k) = r2(xrun); xrun = xrun - xstep; }
      }
      if (l >= i1) {
        double xrun = (x0 - SR_X(i1)) / dw;
        for (int k = i1; k <= l - 1; ++k) { SR_Y(k) = r1(xrun); xrun = xrun - xstep; }
      }
    }
  }
#undef SR_X
#undef SR_Y
}

int launch_humliv(const double *x, int i1, int i2, double x0, double lw, double dwp, double *y, int outer,
                  hipStream_t st) {
  const int n = i2 - i1 + 1;
  if (outer)
    hipLaunchKernelGGL(sr_humliv_outer_kernel, dim3(1), dim3(64), 0, st, x, i1, i2, x0, lw, dwp, y);
  else
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

// ------------------------------------------------------------------------
// A9: look-up-table route.  LutSet.calculate (spect_main_module.py:997-1066) interpolates the three G
// spectra of one level between tabulated (P, T) couples -- linear in P between the two nearest pressures
// at each of the two nearest temperatures, then linear in T (SpectralGcoeff.interpolate,
// spect_classes.py:1349-1375); below the lowest tabulated pressure in T only -- and
// make_abscoeff_isomolec / make_abscoeff_LUTS_fast combine them with the level population
// (spect_main_module.py:2073-2080, 2241-2249).  Thread = (grid point, LOS step); tab: [3][n_pt][n_pts].
// idx[s][4]: table rows (P1,T1), (P1,T2), (P2,T1), (P2,T2), or (P1,TA), (P1,TB), -1, -1 for the T-only case.
// COMBINE: abs += pop (Gabs - Gind), emi += pop Gsp; else the interpolated set itself, g_out[3][n_steps][n_pts].
// ------------------------------------------------------------------------
template <bool COMBINE>
__global__ __launch_bounds__(256) void sr_lut_kernel(const double *__restrict__ tab, int n_pt, int n_pts,
                                                     const int *__restrict__ idx, const double *__restrict__ wgt,
                                                     const double *__restrict__ pop, double *__restrict__ out_a,
                                                     double *__restrict__ out_e) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, s = blockIdx.y, n_steps = gridDim.y;
  if (j >= n_pts) return;
  const int i1 = idx[4 * s + 0], i2 = idx[4 * s + 1], i3 = idx[4 * s + 2], i4 = idx[4 * s + 3];
  const double wp1 = wgt[4 * s + 0], wp2 = wgt[4 * s + 1], wt1 = wgt[4 * s + 2], wt2 = wgt[4 * s + 3];
  double v[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const double *t = tab + (size_t)c * n_pt * n_pts + j;
    if (i3 >= 0) {
      const double c13 = wp1 * t[(size_t)i1 * n_pts] + wp2 * t[(size_t)i3 * n_pts];
      const double c24 = wp1 * t[(size_t)i2 * n_pts] + wp2 * t[(size_t)i4 * n_pts];
      v[c] = wt1 * c13 + wt2 * c24;
    } else {
      v[c] = wt1 * t[(size_t)i1 * n_pts] + wt2 * t[(size_t)i2 * n_pts];
    }
  }
  const size_t o = (size_t)s * n_pts + j;
  if (COMBINE) {
    const double p = pop[s];
    double a = out_a[o];
    a = a + v[2] * p; // spect_main_module.py:2078-2080, in this order
    a = a - v[1] * p;
    out_a[o] = a;
    out_e[o] = out_e[o] + v[0] * p;
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) out_a[(size_t)c * n_steps * n_pts + o] = v[c];
  }
}

int launch_lut(int combine, const double *tab, int n_pt, int n_pts, int n_steps, const int *idx, const double *wgt,
               const double *pop, double *out_a, double *out_e, hipStream_t st) {
  if (n_pts <= 0 || n_steps <= 0) return 0;
  const dim3 grid((n_pts + 255) / 256, n_steps);
  if (combine)
    hipLaunchKernelGGL(sr_lut_kernel<true>, grid, dim3(256), 0, st, tab, n_pt, n_pts, idx, wgt, pop, out_a, out_e);
  else
    hipLaunchKernelGGL(sr_lut_kernel<false>, grid, dim3(256), 0, st, tab, n_pt, n_pts, idx, wgt, pop, out_a, out_e);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------
// Level-factored combine (spect_main_module.py:2036-2106 / 2200-2276 for all steps of a row at once): a block takes
// 256 points of ONE (P, T) row, holds the row's 2 NL pair spectra (and their T-differences) in registers and walks the
// row's steps; the populations of a step are wave-uniform (scalar loads).  HBM: the tables once + the outputs.
// ------------------------------------------------------------------------
template <int NL, bool DT>
__global__ __launch_bounds__(256) void sr_glevel_combine_kernel(
    const double *__restrict__ tab, const double *__restrict__ tab_dT, int n_levels, int n_rows, int n_pts,
    const int *__restrict__ rows_used, const int *__restrict__ row_off, const int *__restrict__ step_of,
    const double *__restrict__ pop, const double *__restrict__ dpop, double inv_dT, double *__restrict__ abs_out,
    double *__restrict__ emi_out, double *__restrict__ dabs_out, double *__restrict__ demi_out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, u = blockIdx.y;
  if (j >= n_pts) return;
  const int row = rows_used[u];
  const size_t plane = (size_t)n_rows * n_pts; // one (level, A | E) plane of the tables
  double A[NL], E[NL], DA[DT ? NL : 1], DE[DT ? NL : 1];
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    const bool on = l < n_levels;
    const size_t o = (size_t)(2 * (on ? l : 0)) * plane + (size_t)row * n_pts + j;
    A[l] = on ? tab[o] : 0.0;
    E[l] = on ? tab[o + plane] : 0.0;
    if (DT) {
      DA[l] = on ? (tab_dT[o] - A[l]) * inv_dT : 0.0;
      DE[l] = on ? (tab_dT[o + plane] - E[l]) * inv_dT : 0.0;
    }
  }
  for (int q = row_off[u]; q < row_off[u + 1]; ++q) {
    const int s = step_of[q];
    const double *p = pop + (size_t)s * n_levels, *dp = DT ? dpop + (size_t)s * n_levels : nullptr;
    double a = 0., e = 0., da = 0., de = 0.;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      if (l < n_levels) { // wave-uniform
        const double pl = p[l];
        a = fma(pl, A[l], a);
        e = fma(pl, E[l], e);
        if (DT) {
          const double dl = dp[l];
          da = fma(pl, DA[l], fma(dl, A[l], da));
          de = fma(pl, DE[l], fma(dl, E[l], de));
        }
      }
    }
    const size_t o = (size_t)s * n_pts + j;
    abs_out[o] = a;
    emi_out[o] = e;
    if (DT) {
      dabs_out[o] = da;
      demi_out[o] = de;
    }
  }
}

// More than 16 levels (SR_MAX_LEVELS is 64): the pair spectra do not fit the registers; every step re-reads them
// (from L2: the steps of a row follow each other).
template <bool DT>
__global__ __launch_bounds__(256) void sr_glevel_combine_stream_kernel(
    const double *__restrict__ tab, const double *__restrict__ tab_dT, int n_levels, int n_rows, int n_pts,
    const int *__restrict__ rows_used, const int *__restrict__ row_off, const int *__restrict__ step_of,
    const double *__restrict__ pop, const double *__restrict__ dpop, double inv_dT, double *__restrict__ abs_out,
    double *__restrict__ emi_out, double *__restrict__ dabs_out, double *__restrict__ demi_out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, u = blockIdx.y;
  if (j >= n_pts) return;
  const int row = rows_used[u];
  const size_t plane = (size_t)n_rows * n_pts;
  for (int q = row_off[u]; q < row_off[u + 1]; ++q) {
    const int s = step_of[q];
    double a = 0., e = 0., da = 0., de = 0.;
    for (int l = 0; l < n_levels; ++l) {
      const size_t o = (size_t)(2 * l) * plane + (size_t)row * n_pts + j;
      const double pl = pop[(size_t)s * n_levels + l], Al = tab[o], El = tab[o + plane];
      a = fma(pl, Al, a);
      e = fma(pl, El, e);
      if (DT) {
        const double dl = dpop[(size_t)s * n_levels + l];
        da = fma(pl, (tab_dT[o] - Al) * inv_dT, fma(dl, Al, da));
        de = fma(pl, (tab_dT[o + plane] - El) * inv_dT, fma(dl, El, de));
      }
    }
    const size_t o = (size_t)s * n_pts + j;
    abs_out[o] = a;
    emi_out[o] = e;
    if (DT) {
      dabs_out[o] = da;
      demi_out[o] = de;
    }
  }
}

int launch_glevel_combine(const double *tab, const double *tab_dT, int n_levels, int n_rows, int n_pts, int n_used,
                          const int *rows_used, const int *row_off, const int *step_of, const double *pop,
                          const double *dpop, double inv_dT, double *abs_out, double *emi_out, double *dabs_out,
                          double *demi_out, hipStream_t st) {
  if (n_pts <= 0 || n_used <= 0) return 0;
  const dim3 grid((n_pts + 255) / 256, n_used);
#define SR_GLC(NL, DT)                                                                                               \
  hipLaunchKernelGGL((sr_glevel_combine_kernel<NL, DT>), grid, dim3(256), 0, st, tab, tab_dT, n_levels, n_rows, n_pts, \
                     rows_used, row_off, step_of, pop, dpop, inv_dT, abs_out, emi_out, dabs_out, demi_out)
#define SR_GLC2(NL) do { if (tab_dT) SR_GLC(NL, true); else SR_GLC(NL, false); } while (0)
  if (n_levels <= 1) SR_GLC2(1);
  else if (n_levels <= 4) SR_GLC2(4);
  else if (n_levels <= 8) SR_GLC2(8);
  else if (n_levels <= 12) SR_GLC2(12);
  else if (n_levels <= 16) SR_GLC2(16);
  else if (tab_dT)
    hipLaunchKernelGGL(sr_glevel_combine_stream_kernel<true>, grid, dim3(256), 0, st, tab, tab_dT, n_levels, n_rows, n_pts,
                       rows_used, row_off, step_of, pop, dpop, inv_dT, abs_out, emi_out, dabs_out, demi_out);
  else
    hipLaunchKernelGGL(sr_glevel_combine_stream_kernel<false>, grid, dim3(256), 0, st, tab, tab_dT, n_levels, n_rows, n_pts,
                       rows_used, row_off, step_of, pop, dpop, inv_dT, abs_out, emi_out, dabs_out, demi_out);
#undef SR_GLC2
#undef SR_GLC
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

// ------------------------------------------------------------------------
// N2: SpectralIntensity.hires_to_lowres (spect_classes.py:1180-1191): cm-1 -> nm
// (grid and spectrum, spcl:404-407, 779-783), Gaussian ILS over +-n_sigma sigma
// by the trapezoid rule on the irregular nm grid (convolve_to_grid_from_irregular,
// spcl:883-918; gaussian, spcl:1926-1934; conv_single = np.trapz, spcl:1162-1164).
// One block per (band, ray); nm index i <-> cm-1 index n-1-i.
// ------------------------------------------------------------------------
// g_lo: grid index of rad's first point (a spectral shard's PARTIAL band integrals: the trapezoids between the
// shard's own points, grid values exactly those of the whole grid; the partial sums of the shards add up).
//
// Round 5: two kernels.  The trapezoid sum of a band, sum_i (x_(i+1) - x_i) (y_(i+1) + y_i) / 2 with y_i = s_i u_i, is
// sum_i s_i W_i with W_i = u_i c_i, c_i = (x_(i+1) - x_(i-1)) / 2 inside the window and the half interval at its two
// ends -- and W depends on the band and the grid alone, not on the spectrum.  A retrieval iteration degrades 144 spectra
// (18 LOS x (radiance + 7 derivatives)) with the same 14 bands: one block per (band, spectrum) re-evaluated the
// Gaussian, four IEEE divisions and two exp per trapezoid 144 times over (0.19 ms of a 1.26 ms configs[4] iteration).
// sr_lowres_weights_kernel makes W [n_bands][n_pts] (cm-1 index order) and the bands' point ranges once per call,
// sr_lowres_apply_kernel is the banded product: a block per (spectrum, chunk of 4096 points, group of 16 bands), the
// chunks' partial sums added in chunk order by sr_lowres_sum_kernel.
__global__ __launch_bounds__(256) void sr_lowres_weights_kernel(int n_pts, int g_lo, double w0, double gstep,
                                                                const double *__restrict__ cen, const double *__restrict__ wid,
                                                                double n_sigma, int n_bands, double *__restrict__ W, // [n_bands][n_pts]
                                                                int *__restrict__ range,  // [n_bands][2]: j_lo, j_hi (exclusive)
                                                                double *__restrict__ Wt) { // [gridDim.y / 16][n_pts][16]
  const int b = blockIdx.y;
  if (b >= n_bands) { // the zero columns that fill the last tile of Wt
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_pts) Wt[((size_t)(b >> 4) * n_pts + j) * 16 + (b & 15)] = 0.0;
    return;
  }
  const double f = cen[b], w = wid[b];
  const double lo = f - n_sigma * w, hi = f + n_sigma * w;
  const double fac = 1 / (w * sqrt(2. * kPi));
  auto gcm = [&](int i) { return w0 + (double)(g_lo + n_pts - 1 - i) * gstep; }; // nm index i <-> cm-1 index n-1-i
  auto xnm = [&](int i) { return 1.e7 / gcm(i); };
  // first i with x >= lo, first i with x > hi (x ascending in i): every thread the same two searches (17 steps)
  int i0, i1;
  {
    int a = 0, c = n_pts;
    while (a < c) { int m = (a + c) >> 1; if (xnm(m) < lo) a = m + 1; else c = m; }
    i0 = a;
    a = 0; c = n_pts;
    while (a < c) { int m = (a + c) >> 1; if (xnm(m) <= hi) a = m + 1; else c = m; }
    i1 = a; // selected: i0 .. i1-1; fewer than two points: no trapezoid
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const bool any = i1 - i0 >= 2;
    range[2 * b] = any ? n_pts - i1 : 0;       // cm-1 indices j = n_pts - 1 - i, i in [i0, i1)
    range[2 * b + 1] = any ? n_pts - i0 : 0;
  }
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_pts) return;
  const int i = n_pts - 1 - j;
  double wgt = 0.0;
  if (i >= i0 && i < i1 && i1 - i0 >= 2) {
    const double g = gcm(i), x = 1.e7 / g, t = (x - f) / w;
    const double u = ((g * g) * 1.e-7) * (fac * exp(-0.5 * (t * t))); // y = s u (spcl:779-783 the cm-1 -> nm factor, :1926-1934 the Gaussian)
    const double xl = i > i0 ? xnm(i - 1) : x, xr = i + 1 < i1 ? xnm(i + 1) : x;
    wgt = u * ((xr - xl) / 2.0);
  }
  W[(size_t)b * n_pts + j] = wgt;
  Wt[((size_t)(b >> 4) * n_pts + j) * 16 + (b & 15)] = wgt;
}

constexpr int kLowresBands = 16;   // bands per block of the apply kernel
#ifndef SR_LOWRES_CHUNK
#define SR_LOWRES_CHUNK 4096
#endif
constexpr int kLowresChunk = SR_LOWRES_CHUNK; // (2048 / 1024 measured 60 / 80 us against 56 for the 144 spectra of a configs[4] iteration) points per block: 144 spectra x 15 chunks fill the chip (a block per spectrum walked
                                   // its 60 000 points alone: 117 dependent rounds of 15 loads, 0.21 ms)
__global__ __launch_bounds__(256) void sr_lowres_apply_kernel(const double *__restrict__ rad, int n_pts,
                                                              const double *__restrict__ W, const int *__restrict__ range,
                                                              int n_bands, int n_chunks,
                                                              double *__restrict__ part) { // [n_rays][n_chunks][n_bands]
  const int ray = blockIdx.x, chunk = blockIdx.y, b0 = blockIdx.z * kLowresBands, nb = min(kLowresBands, n_bands - b0);
  const double *sp = rad + (size_t)ray * n_pts;
  const int c_lo = chunk * kLowresChunk, c_hi = min(c_lo + kLowresChunk, n_pts);
  double acc[kLowresBands];
#pragma unroll
  for (int q = 0; q < kLowresBands; ++q) acc[q] = 0.0;
  // W is zero outside a band's window: no per-point range test, only whole bands that miss the chunk are skipped
  // (block-uniform); the loads of a point's bands are independent of each other
  bool use[kLowresBands];
  bool any_use = false;
#pragma unroll
  for (int q = 0; q < kLowresBands; ++q) {
    use[q] = q < nb && range[2 * (b0 + q)] < c_hi && range[2 * (b0 + q) + 1] > c_lo;
    any_use = any_use || use[q];
  }
  if (!any_use) { // none of this block's bands reaches the chunk (block-uniform): zeros, and the spectrum is not read
    if ((int)threadIdx.x < nb) part[((size_t)ray * n_chunks + chunk) * n_bands + b0 + threadIdx.x] = 0.0;
    return;
  }
  for (int j = c_lo + (int)threadIdx.x; j < c_hi; j += (int)blockDim.x) {
    const double sv = sp[j];
#pragma unroll
    for (int q = 0; q < kLowresBands; ++q)
      if (use[q]) acc[q] = fma(sv, W[(size_t)(b0 + q) * n_pts + j], acc[q]);
  }
  __shared__ double red[kLowresBands][4];
#pragma unroll
  for (int q = 0; q < kLowresBands; ++q) {
    double v = acc[q];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    if ((threadIdx.x & 63) == 0) red[q][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if ((int)threadIdx.x < nb) {
    const int q = threadIdx.x;
    part[((size_t)ray * n_chunks + chunk) * n_bands + b0 + q] = (red[q][0] + red[q][1]) + (red[q][2] + red[q][3]);
  }
}
__global__ void sr_lowres_sum_kernel(const double *__restrict__ part, int n_rays, int n_chunks, int n_bands, int out_units,
                                     double *__restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_rays * n_bands) return;
  const int ray = t / n_bands, b = t - ray * n_bands;
  double v = 0.0;
  for (int c = 0; c < n_chunks; ++c) v += part[((size_t)ray * n_chunks + c) * n_bands + b]; // in chunk order: deterministic
  v = v * 1.e-3;                    // 'ergscm2' -> 'Wm2', spcl:1215-1218
  if (out_units == 1) v = v * 1.e3; // -> 'ergscm2', spcl:1224-1228
  if (out_units == 2) v = v * 1.e5; // -> 'nWcm2',   spcl:1230-1234
  out[t] = v;
}

// ... of the partial sums sr_limb_fold_sens_lds_kernel<., true> leaves per 64-point slot (a wave's points): a block per
// (spectrum, band tile), thread = (band of the tile, one of 16 interleaved slices of the slots); a band's slots inside its
// window only; slice sums added in slice order: deterministic.
__global__ __launch_bounds__(256) void sr_lowres_sum_blocks_kernel(const double *__restrict__ part, const int *__restrict__ range,
                                                                   int n_blocks, int n_bands, int out_units, double *__restrict__ out) {
  __shared__ double red[16][16];
  const int row = blockIdx.x, tile = blockIdx.y, n_tiles = gridDim.y;
  const int col = threadIdx.x & 15, slice = threadIdx.x >> 4, b = 16 * tile + col;
  double v = 0.0;
  if (b < n_bands) {
    const int r0 = range[2 * b], r1 = range[2 * b + 1];
    if (r1 > r0)
      for (int c = (r0 >> 6) + slice; c <= (r1 - 1) >> 6; c += 16) v += part[(((size_t)row * n_blocks + c) * n_tiles + tile) * 16 + col];
  }
  red[slice][col] = v;
  __syncthreads();
  if (slice != 0 || b >= n_bands) return;
  v = 0.0;
#pragma unroll
  for (int sl = 0; sl < 16; ++sl) v += red[sl][col];
  v = v * 1.e-3;                    // as sr_lowres_sum_kernel
  if (out_units == 1) v = v * 1.e3;
  if (out_units == 2) v = v * 1.e5;
  out[(size_t)row * n_bands + b] = v;
}

static int lowres_chunks(int n_pts) { return (n_pts + kLowresChunk - 1) / kLowresChunk; }
// fused: the partial sums are per wave of the recursion (launch_fold_dense with the scratch)
size_t lowres_scratch_bytes(int n_pts, int n_bands, int n_rays, bool fused) {
  const size_t tiles16 = (size_t)((n_bands + 15) / 16) * 16;
  const size_t n_part = fused ? 4 * (size_t)((n_pts + 255) / 256) * tiles16 : (size_t)lowres_chunks(n_pts) * n_bands;
  return sizeof(double) * ((size_t)n_bands + tiles16) * n_pts + sizeof(double) * (size_t)n_rays * n_part +
         sizeof(int) * (2 * (size_t)n_bands + 2);
}

int launch_lowres_weights(int n_pts, int g_lo, double w0, double gstep, const double *cen, const double *wid, int n_bands,
                          double n_sigma, void *scratch, hipStream_t st) {
  if (n_bands <= 0) return 0;
  const LowresScratch L = lowres_layout(scratch, n_pts, n_bands);
  hipLaunchKernelGGL(sr_lowres_weights_kernel, dim3((n_pts + 255) / 256, (n_bands + 15) / 16 * 16), dim3(256), 0, st, n_pts, g_lo, w0,
                     gstep, cen, wid, n_sigma, n_bands, L.W, L.range, L.Wt);
  return (int)hipGetLastError();
}

int launch_lowres_sum_blocks(int n_pts, int n_rows, int n_bands, int out_units, double *out, void *scratch, hipStream_t st) {
  if (n_bands <= 0 || n_rows <= 0) return 0;
  const LowresScratch L = lowres_layout(scratch, n_pts, n_bands);
  hipLaunchKernelGGL(sr_lowres_sum_blocks_kernel, dim3(n_rows, (n_bands + 15) / 16), dim3(256), 0, st, L.part, L.range,
                     4 * ((n_pts + 255) / 256), n_bands, out_units, out);
  return (int)hipGetLastError();
}

int launch_lowres(const double *rad, int n_pts, int g_lo, int n_rays, double w0, double gstep, const double *cen,
                  const double *wid, int n_bands, double n_sigma, int out_units, double *out, void *scratch, hipStream_t st,
                  bool weights) {
  if (n_bands <= 0 || n_rays <= 0) return 0;
  const int n_chunks = lowres_chunks(n_pts);
  const LowresScratch L = lowres_layout(scratch, n_pts, n_bands);
  if (weights) launch_lowres_weights(n_pts, g_lo, w0, gstep, cen, wid, n_bands, n_sigma, scratch, st);
  hipLaunchKernelGGL(sr_lowres_apply_kernel, dim3(n_rays, n_chunks, (n_bands + kLowresBands - 1) / kLowresBands), dim3(256), 0, st,
                     rad, n_pts, L.W, L.range, n_bands, n_chunks, L.part);
  hipLaunchKernelGGL(sr_lowres_sum_kernel, dim3((n_rays * n_bands + 255) / 256), dim3(256), 0, st, L.part, n_rays, n_chunks, n_bands,
                     out_units, out);
  return (int)hipGetLastError();
}

int launch_curgod(int which, const double *nd, const double *vmr, const double *f, const double *x,
                  const int *off, int n_seg, double *res, hipStream_t st) {
  if (n_seg <= 0) return 0;
  hipLaunchKernelGGL(sr_curgod_kernel, dim3((n_seg + 63) / 64), dim3(64), 0, st, which, nd, vmr, f, x, off,
                     n_seg, res);
  return (int)hipGetLastError();
}

} // namespace sr
