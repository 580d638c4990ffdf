"""Host-side engine objects over the C ABI: device-resident line sets, per-layer
coefficient spectra, limb radiances.  torch is used for device memory, streams
and (in distributed.py) the RCCL all-gather only.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import lib, check, dp, ip


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(ip)


_GRID_CACHE = []   # [(grid array, (w0, step, n))], the last few grids checked: a retrieval loop passes the same array in every iteration


def grid_params(grid):
    """(w0, step, n) of an equally spaced numpy-arange grid (spect_main_module.py:1267);
    raises if the grid is not exactly w0 + j*step.  The check walks the whole grid (1e5..2e6 points): the result is
    remembered for the array OBJECT it was made for (grids are not modified in place anywhere in this package)."""
    for g, res in _GRID_CACHE:
        # (identity alone says nothing about an array edited in place, ADVICE round 5: the remembered answer must still
        # describe the array's ends and length; an edit of interior points of a grid no caller makes goes unseen)
        if g is grid and g.size == res[2] and g[0] == res[0] and g[1] - g[0] == res[1] and g[-1] == res[0] + (res[2] - 1) * res[1]:
            return res
    arr = np.asarray(grid, dtype=np.float64)
    if arr.ndim != 1 or arr.size < 2:
        raise ValueError("spectral grid must be 1-D with at least 2 points")
    w0, step = float(arr[0]), float(arr[1] - arr[0])
    if not np.array_equal(arr, w0 + np.arange(arr.size) * step):
        raise ValueError("spectral grid is not an np.arange grid (w0 + j*step); the reference assumes "
                         "equal spacing (SpectralGrid.step, spect_classes.py:366)")
    res = (w0, step, arr.size)
    if isinstance(grid, np.ndarray):
        _GRID_CACHE[:] = [e for e in _GRID_CACHE if e[0] is not grid]
        _GRID_CACHE.append((grid, res))
        del _GRID_CACHE[:-8]
    return res


def set_device(index):
    check(lib.sr_set_device(int(index)), "sr_set_device")
    torch.cuda.set_device(int(index))


def device_info():
    name = C.create_string_buffer(256)
    cu = C.c_int(0)
    mem = C.c_double(0)
    check(lib.sr_device_info(name, 256, C.byref(cu), C.byref(mem)), "sr_device_info")
    rec, conf = C.c_int(0), C.c_int(0)
    check(lib.sr_recommended_hw_queues(C.byref(rec), C.byref(conf)), "sr_recommended_hw_queues")
    return dict(name=name.value.decode(), cu_count=cu.value, hbm_gib=mem.value, hw_queues=conf.value,
                hw_queues_recommended=rec.value)


def _stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class LineSet(object):
    """Device-resident line list of one iso-molecule bound to a spectral grid.

    lines: dict of arrays freq, a_coeff, e_lower, g_up, g_lo, air_broad, t_dep_broad
           [, lev_up, lev_lo] (SpectLine fields, spect_classes.py:50-51)
    level_energies: [] for the reference's 'all' set, else E_vib per level (cm^-1)
    """

    def __init__(self, lines, grid, mol, iso, mm, level_energies=()):
        self.w0, self.step, self.n_grid = grid_params(grid)
        self.mol, self.iso, self.mm = int(mol), int(iso), float(mm)
        self.level_energies = np.ascontiguousarray(level_energies, dtype=np.float64)
        keep = {}
        ld = _lib.LinesDesc()
        ld.n_lines = len(lines["freq"])
        for n in ("freq", "a_coeff", "e_lower", "g_up", "g_lo", "air_broad", "t_dep_broad"):
            keep[n], p = _d(lines[n])
            if keep[n].size != ld.n_lines:
                raise ValueError("line array %s has the wrong length" % n)
            setattr(ld, n, p)
        if self.level_energies.size:
            for n in ("lev_up", "lev_lo"):
                keep[n], p = _i(lines[n])
                setattr(ld, n, p)
        iso_d = _lib.IsoMolecDesc(self.mol, self.iso, self.mm, self.level_energies.size,
                                  self.level_energies.ctypes.data_as(dp))
        gd = _lib.GridDesc(self.w0, self.step, self.n_grid)
        h = C.c_void_p()
        kept = C.c_int64(0)
        check(lib.sr_lineset_create(C.byref(ld), C.byref(iso_d), C.byref(gd), C.byref(h), C.byref(kept)),
              "sr_lineset_create")
        self._h = h
        self.n_kept = kept.value
        self._step_cache = None

    def close(self):
        if getattr(self, "_h", None) and lib is not None:
            lib.sr_lineset_destroy(self._h)
            self._h = None

    __del__ = close

    def _layers(self, temps, press, tvib, q_part):
        temps, tp = _d(temps)
        press, pp = _d(press)
        n = temps.size
        if press.size != n:
            raise ValueError("temps and press differ in length")
        keep = [temps, press]
        tvp = qp = None
        if tvib is not None:
            tvib, tvp = _d(tvib)
            if tvib.shape != (self.level_energies.size, n):
                raise ValueError("tvib must be [n_levels, n_layers]")
            keep.append(tvib)
        if q_part is not None:
            q_part, qp = _d(q_part)
            if q_part.size != n:
                raise ValueError("q_part must be [n_layers]")
            keep.append(q_part)
        return _lib.LayersDesc(n, tp, pp, tvp, qp), keep, n

    def abscoeff_layers(self, temps, press, tvib=None, q_part=None, g_lo=0, g_hi=None, out=None):
        """Per-layer abs/emi coefficient spectra on the GPU over grid shard [g_lo, g_hi).
        Returns two torch float64 CUDA tensors [n_layers, g_hi-g_lo] (HBM resident)."""
        g_hi = self.n_grid if g_hi is None else int(g_hi)
        desc, keep, n = self._layers(temps, press, tvib, q_part)
        npts = g_hi - int(g_lo)
        if npts <= 0:
            raise ValueError("empty shard")
        if out is None:
            ab = torch.empty((n, npts), dtype=torch.float64, device="cuda")
            em = torch.empty((n, npts), dtype=torch.float64, device="cuda")
        else:
            ab, em = out
            assert ab.shape == (n, npts) and em.shape == (n, npts) and ab.is_contiguous() and em.is_contiguous()
        check(lib.sr_abscoeff_layers_dev(self._h, C.byref(desc), int(g_lo), g_hi, C.c_void_p(ab.data_ptr()),
                                         C.c_void_p(em.data_ptr()), _stream_ptr()), "sr_abscoeff_layers_dev")
        return ab, em

    def limb_step(self, temps, press, los, tvib=None, q_part=None, g_lo=0, g_hi=None, out=None, rad=None, grid=None):
        """One forward-model step in ONE library call (sr_limb_step_dev): the coefficient spectra of this gas over
        [g_lo, g_hi) into `out` (as abscoeff_layers) and the radiances of the resident LOS batch `los` (LimbLOS, one
        gas) through them -- make_abscoeff_isomolec + radtran_fast per spectrum (spect_main_module.py:1979-1990, 2838).
        The layer descriptor is cached on the arrays' identity: a loop over the same atmosphere objects pays for
        neither conversions nor staging.  Returns (abs, emi, rad)."""
        g_hi = self.n_grid if g_hi is None else int(g_hi)
        key = (id(temps), id(press), id(tvib), id(q_part))
        c = self._step_cache
        if c is None or c[0] != key:
            desc, keep, n = self._layers(temps, press, tvib, q_part)
            c = (key, desc, keep, n)
            # cached only when the descriptor points into the caller's own arrays (contiguous float64: no copies were
            # made), so that values changed in place are seen; the arrays are referenced, their ids stay theirs
            given = [a for a in (temps, press, tvib, q_part) if a is not None]
            self._step_cache = c if all(any(k is a for k in keep) for a in given) else None
        _, desc, _, n = c
        npts = g_hi - int(g_lo)
        if npts <= 0:
            raise ValueError("empty shard")
        if out is None:
            out = (torch.empty((n, npts), dtype=torch.float64, device="cuda"), torch.empty((n, npts), dtype=torch.float64, device="cuda"))
        ab, em = out
        if rad is None:
            rad = torch.empty((los.n_rays, npts), dtype=torch.float64, device="cuda")
        assert ab.shape == (n, npts) and em.shape == (n, npts) and rad.shape == (los.n_rays, npts)
        assert ab.is_contiguous() and em.is_contiguous() and rad.is_contiguous()
        h = los.handle(n, grid)
        check(lib.sr_limb_step_dev(self._h, C.byref(desc), int(g_lo), g_hi, ab.data_ptr(), em.data_ptr(), h, rad.data_ptr(),
                                   _stream_ptr()), "sr_limb_step_dev")
        return ab, em, rad

    def set_bounds_temps(self, temps=None, linear_weights=False):
        """Place the Humlicek region boundaries of the next coefficient calls as at `temps` [n_layers] (None: at each
        call's own temperatures again): sr_lineset_set_bounds_temps.  For finite differences in T: c(T + dT) and c(T)
        then share their index-computed seams and the difference quotient is smooth (coefficients_dT).
        linear_weights: the line weights (G coefficients, normalisation) taken at `temps` and continued to the call's
        temperature to first order (sr_lineset_set_linear_weights): the quotient then has no curvature term from the
        Boltzmann factors and its step can be 0.05 K instead of 0.002 K (LevelFactored)."""
        if temps is None:
            check(lib.sr_lineset_set_bounds_temps(self._h, None, 0), "sr_lineset_set_bounds_temps")
            check(lib.sr_lineset_set_linear_weights(self._h, 0), "sr_lineset_set_linear_weights")
        else:
            t, tp = _d(temps)
            check(lib.sr_lineset_set_bounds_temps(self._h, tp, int(t.size)), "sr_lineset_set_bounds_temps")
            check(lib.sr_lineset_set_linear_weights(self._h, int(bool(linear_weights))), "sr_lineset_set_linear_weights")

    def gcoeff_layers(self, temps, press, level=0, g_lo=0, g_hi=None):
        """Per-ctype G-coefficient spectra of one level at every (P, T): CUDA float64
        [3, n_layers, g_hi-g_lo], ctype 0 sp_emission, 1 ind_emission, 2 absorption
        (sr_gcoeff_layers_dev = LutSet.add_PT / SpectralGcoeff.BuildCoeff)."""
        g_hi = self.n_grid if g_hi is None else int(g_hi)
        desc, keep, n = self._layers(temps, press, None, None)
        npts = g_hi - int(g_lo)
        if npts <= 0:
            raise ValueError("empty shard")
        g = torch.empty((3, n, npts), dtype=torch.float64, device="cuda")
        check(lib.sr_gcoeff_layers_dev(self._h, C.byref(desc), int(level), int(g_lo), g_hi,
                                       C.c_void_p(g.data_ptr()), _stream_ptr()), "sr_gcoeff_layers_dev")
        return g

    def gcoeff_levels(self, temps, press, g_lo=0, g_hi=None, out=None):
        """G-coefficient spectra of ALL levels at every (P, T): CUDA float64 [n_levels or 1, 3, n_rows, g_hi-g_lo], ctype
        0 sp_emission, 1 ind_emission, 2 absorption -- LookUpTable.make's inner loop over the levels
        (spect_main_module.py:759-772) in one call (sr_gcoeff_levels_dev: the multi-channel pass, every line once)."""
        g_hi = self.n_grid if g_hi is None else int(g_hi)
        desc, keep, n = self._layers(temps, press, None, None)
        npts = g_hi - int(g_lo)
        if npts <= 0:
            raise ValueError("empty shard")
        nl = max(int(self.level_energies.size), 1)
        if out is None:
            out = torch.empty((nl, 3, n, npts), dtype=torch.float64, device="cuda")
        assert out.shape == (nl, 3, n, npts) and out.is_contiguous() and out.dtype == torch.float64
        check(lib.sr_gcoeff_levels_dev(self._h, C.byref(desc), int(g_lo), g_hi, C.c_void_p(out.data_ptr()), _stream_ptr()),
              "sr_gcoeff_levels_dev")
        return out

    def glevel_pairs(self, temps, press, g_lo=0, g_hi=None, out=None):
        """Level-pair tables of the level-factored route (sr_glevel_pairs_dev): CUDA float64
        [n_levels or 1, 2, n_rows, g_hi-g_lo] with [L, 0] = Gabs_L - Gind_L and [L, 1] = Gsp_L at every (P, T) row --
        what pop_L multiplies in the reference's combine loop (spect_main_module.py:2073-2080).  No populations enter:
        the tables depend on (P, T) alone and serve every LOS step / SZA set that shares a row (glevel_combine)."""
        g_hi = self.n_grid if g_hi is None else int(g_hi)
        desc, keep, n = self._layers(temps, press, None, None)
        npts = g_hi - int(g_lo)
        if npts <= 0:
            raise ValueError("empty shard")
        nl = max(int(self.level_energies.size), 1)
        if out is None:
            out = torch.empty((nl, 2, n, npts), dtype=torch.float64, device="cuda")
        assert out.shape == (nl, 2, n, npts) and out.is_contiguous() and out.dtype == torch.float64
        check(lib.sr_glevel_pairs_dev(self._h, C.byref(desc), int(g_lo), g_hi, C.c_void_p(out.data_ptr()), _stream_ptr()),
              "sr_glevel_pairs_dev")
        return out

    def level_populations(self, temps, tvib=None, q_part=None, derivative=False):
        """pop [n_steps, n_levels or 1] = exp(-c2 E_L / Tvib_L) / Q(T) (spect_main_module.py:2049-2073; 1 / Q for the
        'all' set), tvib [n_levels, n_steps] or None (LTE: Tvib = T).  derivative=True: also d pop / d T with the
        vibrational temperatures held fixed when given (the definition of coefficients_dT), following T in LTE; Q'
        from the derivative of CalcPartitionSum's own interpolant unless q_part pins Q (then Q' = 0)."""
        from . import spect_classes as spcl
        T = np.ascontiguousarray(temps, dtype=np.float64)
        if q_part is not None:
            Q = np.asarray(q_part, float)
        else:   # on the distinct temperatures only (thousands of LOS steps share a few hundred)
            Tu, inv = np.unique(T, return_inverse=True)
            Q = np.atleast_1d(spcl.CalcPartitionSum(self.mol, self.iso, Tu))[inv]
        E = self.level_energies
        c2 = spcl.c2
        if E.size == 0:
            pop = (1.0 / Q)[:, None]
            boltz_dT = np.zeros_like(pop)
        else:
            tv = np.broadcast_to(T, (E.size, T.size)) if tvib is None else np.asarray(tvib, float)
            if tv.shape != (E.size, T.size):
                raise ValueError("tvib must be [n_levels, n_steps]")
            pop = (np.exp(-c2 * E[:, None] / tv) / Q[None, :]).T
            boltz_dT = (c2 * E[:, None] / tv ** 2).T if tvib is None else np.zeros((T.size, E.size))
        if not derivative:
            return np.ascontiguousarray(pop)
        dq = np.zeros_like(Q) if q_part is not None else np.atleast_1d(spcl.CalcPartitionSum_dT(self.mol, self.iso, T))
        dpop = pop * (boltz_dT - (dq / Q)[:, None])
        return np.ascontiguousarray(pop), np.ascontiguousarray(dpop)

    def abscoeff_level(self, temps, press, level, tvib=None, q_part=None, g_lo=0, g_hi=None):
        """One level's share of abs / emi (track_levels, spect_main_module.py:2083-2087)."""
        g_hi = self.n_grid if g_hi is None else int(g_hi)
        desc, keep, n = self._layers(temps, press, tvib, q_part)
        npts = g_hi - int(g_lo)
        ab = torch.empty((n, npts), dtype=torch.float64, device="cuda")
        em = torch.empty((n, npts), dtype=torch.float64, device="cuda")
        check(lib.sr_abscoeff_level_dev(self._h, C.byref(desc), int(level), int(g_lo), g_hi,
                                        C.c_void_p(ab.data_ptr()), C.c_void_p(em.data_ptr()), _stream_ptr()),
              "sr_abscoeff_level_dev")
        return ab, em

    def abscoeff_layers_host(self, temps, press, tvib=None, q_part=None, g_lo=0, g_hi=None):
        """Same through the host-buffer entry point (numpy in, numpy out)."""
        g_hi = self.n_grid if g_hi is None else int(g_hi)
        desc, keep, n = self._layers(temps, press, tvib, q_part)
        ab = np.empty((n, g_hi - int(g_lo)))
        em = np.empty_like(ab)
        check(lib.sr_abscoeff_layers(self._h, C.byref(desc), int(g_lo), g_hi, ab.ctypes.data_as(dp),
                                     em.ctypes.data_as(dp)), "sr_abscoeff_layers")
        return ab, em

    def last_kernel_ms(self):
        """Kernel times (ms) of the last abscoeff_layers call: (prep, farfield|wings,
        near wings|cores, near zones|0, reserved); with overlap (default, far-field mode): (prep, whole
        coefficient op, 0, 0, 0).  See sr_last_kernel_ms."""
        ms = (C.c_float * 5)()
        check(lib.sr_last_kernel_ms(self._h, ms), "sr_last_kernel_ms")
        return tuple(ms)


    def last_level_tables_ms(self):
        """HIP-event times [ms] of the last multi-channel table build (glevel_pairs / gcoeff_levels): (tables, zones kernel,
        rest of the far passes, wings kernel); under set_overlap(0) the two kernels' stand-alone durations."""
        ms = (C.c_float * 4)()
        check(lib.sr_last_level_tables_ms(self._h, ms), "sr_last_level_tables_ms")
        return [float(v) for v in ms]

    def last_eval_counts(self):
        """Executed-work counters of the last abscoeff_layers call made under set_counting(1)
        (sr_last_eval_counts): dict name -> count."""
        c = (C.c_uint64 * 10)()
        check(lib.sr_last_eval_counts(self._h, c), "sr_last_eval_counts")
        names = ("farfield_expansions", "region1_evals", "window_end_expansions", "poly_point_levels",
                 "region2_evals", "region3_evals", "region4_evals", "multipole_line_sides", "box_pair_translations")
        return dict(zip(names, (int(v) for v in c)))


def lut_interp(table, idx4, wgt4, pops=None, out=None):
    """LutSet.calculate for n_steps LOS steps on a G table resident in HBM (sr_lut_interp_dev).
    table: CUDA [3, n_PT, n_grid]; idx4 [n_steps, 4] rows, wgt4 [n_steps, 4] = (wP1, wP2, wT1, wT2).
    pops None: returns the interpolated set [3, n_steps, n_grid]; else accumulates the population-weighted
    combine into out = (abs, emi), CUDA [n_steps, n_grid]."""
    assert table.is_cuda and table.dtype == torch.float64 and table.is_contiguous() and table.dim() == 3
    idx4, ipp = _i(np.asarray(idx4).reshape(-1, 4))
    wgt4, wp = _d(np.asarray(wgt4).reshape(-1, 4))
    n_steps, n_pt, n_pts = idx4.shape[0], table.shape[1], table.shape[2]
    if pops is None:
        g = torch.empty((3, n_steps, n_pts), dtype=torch.float64, device="cuda")
        check(lib.sr_lut_interp_dev(C.c_void_p(table.data_ptr()), n_pt, n_pts, n_steps, ipp, wp, None, 0,
                                    C.c_void_p(g.data_ptr()), None, _stream_ptr()), "sr_lut_interp_dev")
        return g
    pops, pp = _d(pops)
    ab, em = out
    for t in (ab, em):
        assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.shape == (n_steps, n_pts)
    check(lib.sr_lut_interp_dev(C.c_void_p(table.data_ptr()), n_pt, n_pts, n_steps, ipp, wp, pp, 1,
                                C.c_void_p(ab.data_ptr()), C.c_void_p(em.data_ptr()), _stream_ptr()),
          "sr_lut_interp_dev")
    return ab, em


def glevel_combine(tab, step_row, pop, tab_dT=None, dpop=None, dT=None, out=None):
    """The combine loop of the level-factored route for all LOS steps at once (sr_glevel_combine_dev):
    abs[s] = sum_L pop[s, L] tab[L, 0, row[s]], emi[s] = sum_L pop[s, L] tab[L, 1, row[s]] -- one pass over the pair
    tables (LineSet.glevel_pairs), the steps of a (P, T) row taken together.  With tab_dT (the tables at T + dT, region
    boundaries frozen at T), dpop [n_steps, n_levels] and dT also d abs / d T and d emi / d T of every step: population
    part analytic, d G / d T by difference.  Returns (abs, emi) or ((abs, emi), (dabs, demi)), CUDA [n_steps, n_pts]."""
    assert tab.is_cuda and tab.dtype == torch.float64 and tab.is_contiguous() and tab.dim() == 4 and tab.shape[1] == 2
    nl, _, n_rows, n_pts = tab.shape
    step_row, sp = _i(step_row)
    pop, pp = _d(pop)
    n_steps = step_row.size
    if pop.shape != (n_steps, nl):
        raise ValueError("pop must be [n_steps, n_levels]")
    want_dT = tab_dT is not None
    if want_dT:
        assert tab_dT.shape == tab.shape and tab_dT.is_contiguous() and tab_dT.dtype == torch.float64
        dpop, dpp = _d(dpop)
        if dpop.shape != pop.shape or not dT:
            raise ValueError("the temperature derivative needs dpop [n_steps, n_levels] and dT")
    else:
        dpp = None
    n_out = 4 if want_dT else 2
    if out is None:
        out = [torch.empty((n_steps, n_pts), dtype=torch.float64, device="cuda") for _ in range(n_out)]
    for t in out:
        assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.shape == (n_steps, n_pts)
    ptr = lambda t: C.c_void_p(t.data_ptr())
    check(lib.sr_glevel_combine_dev(ptr(tab), ptr(tab_dT) if want_dT else None, nl, n_rows, n_pts, n_steps, sp, pp, dpp,
                                    (1.0 / dT) if want_dT else 0.0, ptr(out[0]), ptr(out[1]),
                                    ptr(out[2]) if want_dT else None, ptr(out[3]) if want_dT else None, _stream_ptr()),
          "sr_glevel_combine_dev")
    return ((out[0], out[1]), (out[2], out[3])) if want_dT else (out[0], out[1])


class LevelFactored(object):
    """The level-factored route as one object: the pair tables of a LineSet on a set of (P, T) rows (LineSet.glevel_pairs;
    with dT also at T + dT, region boundaries frozen at T) and the combine for any number of LOS steps on those rows
    (glevel_combine).  The reference's own structure (make_abscoeff_isomolec: G spectra per (P, T) through
    calc_shapes_lines + add_PT, spect_main_module.py:1963-1990, then the population loop :2036-2106): line shapes are
    evaluated once per (P, T) row however many steps / SZA sets / vibrational-temperature states share it -- a 3-D path
    whose kinetic temperature lives on (latitude box, altitude) and whose vibrational temperatures follow the local
    SZA (radtran_3Dvs2D_sza30-80_test.py:66-116) has ~10x more steps than rows.

    Worth it when rows are shared: the tables cost ~3.7 folded ops per 80 rows (24 output spectra instead of 2: twelve
    passes over sub-linesets, tools/glevel_probe.py), a folded op per step costs 1 per 80 steps."""

    def __init__(self, ls, temps_rows, press_rows, dT=None, g_lo=0, g_hi=None, linear_weights=True):
        """dT: also build the tables at T + dT for the temperature derivative of `steps` -- region boundaries frozen at
        T and (linear_weights, default) the line weights linearised about T, so that (A(T + dT) - A(T)) / dT is
        sum_i w_i' y_i(T + dT) + sum_i w_i (y_i(T + dT) - y_i(T)) / dT: the weights' part exact, only the shapes
        differenced -- their dependence on T is weak (d ln y / d T ~ 1 / 2T) and smooth, so dT = 0.05 K carries
        1.5e-4 of truncation where the exact-weight quotient needed 0.002 K and sat on the reference's
        single-precision staircase (1e-7 |c| / dT = 1e-3 of the derivative)."""
        self.ls = ls
        self.dT = dT
        self._shard, self._linear = (g_lo, g_hi), linear_weights
        self.tab = self.tab_dT = None
        self.rebuild(temps_rows, press_rows)

    def rebuild(self, temps_rows, press_rows):
        """New (P, T) rows (a retrieval iteration that moved the temperatures): the tables are rebuilt IN PLACE when the
        number of rows is the same -- 6 GB per build at configs[3] size stay out of the allocator."""
        ls, (g_lo, g_hi) = self.ls, self._shard
        self.temps = np.ascontiguousarray(temps_rows, dtype=np.float64)
        self.press = np.ascontiguousarray(press_rows, dtype=np.float64)
        keep = self.tab is not None and self.tab.shape[2] == self.temps.size
        self.tab = ls.glevel_pairs(self.temps, self.press, g_lo=g_lo, g_hi=g_hi, out=self.tab if keep else None)
        if self.dT:
            ls.set_bounds_temps(self.temps, linear_weights=self._linear)
            try:
                self.tab_dT = ls.glevel_pairs(self.temps + self.dT, self.press, g_lo=g_lo, g_hi=g_hi,
                                              out=self.tab_dT if keep and self.tab_dT is not None else None)
            finally:
                ls.set_bounds_temps(None)
        return self

    @staticmethod
    def unique_rows(temps, press):
        """(T_rows, P_rows, step_row): the distinct (P, T) couples of a list of steps and every step's row."""
        pt = np.stack([np.asarray(press, float), np.asarray(temps, float)], axis=1)
        rows, inv = np.unique(pt, axis=0, return_inverse=True)
        return rows[:, 1].copy(), rows[:, 0].copy(), np.asarray(inv, np.int32).reshape(-1)

    def steps(self, step_row, tvib=None, q_part=None, derivative=False, out=None):
        """(abs, emi) [n_steps, n_pts] of the steps (each on table row step_row[s], vibrational temperatures tvib
        [n_levels, n_steps], None = LTE); derivative=True: ((abs, emi), (d abs / d T, d emi / d T)), needs dT."""
        step_row = np.ascontiguousarray(step_row, dtype=np.int32)
        T = self.temps[step_row]
        if not derivative:
            return glevel_combine(self.tab, step_row, self.ls.level_populations(T, tvib=tvib, q_part=q_part), out=out)
        if self.tab_dT is None:
            raise ValueError("LevelFactored was built without dT: no temperature derivative")
        pop, dpop = self.ls.level_populations(T, tvib=tvib, q_part=q_part, derivative=True)
        return glevel_combine(self.tab, step_row, pop, tab_dT=self.tab_dT, dpop=dpop, dT=self.dT, out=out)


def set_timing(on):
    """0: no timing events in the coefficient op (host-bound loops: small shards); last_kernel_ms is then unavailable.
    2: also around the recursion of the retrieval's forward model (LimbLOS.last_forward_kernel_ms)."""
    check(lib.sr_set_timing(int(on)), "sr_set_timing")


def set_counting(on):
    """1: the next coefficient ops run the counting instantiations (executed-work accounting, untimed)."""
    check(lib.sr_set_counting(int(bool(on))), "sr_set_counting")


def radiance_rays(abs_c, emi_c, seg_off, seg_layer, seg_col, rad0=None):
    """Limb radiance recursion for a batch of rays (include/spectrobot_hip.h).
    abs_c/emi_c: CUDA float64 [n_layers, n_pts]; returns CUDA float64 [n_rays, n_pts]."""
    assert abs_c.is_cuda and abs_c.dtype == torch.float64 and abs_c.is_contiguous()
    assert emi_c.shape == abs_c.shape and emi_c.is_contiguous()
    n_layers, n_pts = abs_c.shape
    seg_off, op = _i(seg_off)
    seg_layer, lp = _i(seg_layer)
    seg_col, cp = _d(seg_col)
    n_rays = seg_off.size - 1
    if rad0 is None:
        rad = torch.empty((n_rays, n_pts), dtype=torch.float64, device="cuda")
        init = 0
    else:
        rad = rad0
        assert rad.shape == (n_rays, n_pts) and rad.is_contiguous()
        init = 1
    check(lib.sr_radiance_rays_dev(C.c_void_p(abs_c.data_ptr()), C.c_void_p(emi_c.data_ptr()), n_layers, n_pts,
                                   n_rays, op, lp, cp, init, C.c_void_p(rad.data_ptr()), _stream_ptr()),
          "sr_radiance_rays_dev")
    return rad


class _ParHandle(object):
    def __init__(self, h, keep=None):
        self.h, self.keep = h, keep     # keep: the arrays an identity key refers to (their ids stay theirs while kept)


def _array_key(a):
    """Key of an array for the resident-handle cache: a READ-ONLY ndarray cannot change under its identity (and the cache
    entry keeps it alive, so the id is not re-used): ("id", id(a)), free; anything else by content (_content_key).
    retrieval.LimbScene.profile_weights hands out read-only arrays: a retrieval iteration then pays no hashing of its
    60 000-entry weight table (60-100 us of a 480 us iteration)."""
    if isinstance(a, np.ndarray) and not a.flags.writeable:
        return ("id", id(a))
    return _content_key(a)


def _content_key(a):
    """A key for an array's CONTENT (dtype, shape, bytes hashed): identity says nothing about an array edited in place or
    about a new array that got a dead one's id."""
    a = np.ascontiguousarray(a)
    if a.nbytes <= 4096 or a.nbytes % 8:
        return (a.dtype.str, a.shape, a.tobytes())
    # large tables (a retrieval's parameter weights, every iteration): two vectorised reductions over the 64-bit words
    # -- xor and a position-weighted wrapped sum -- instead of a cryptographic hash (30 us against 0.5 ms per call)
    w = a.reshape(-1).view(np.uint64)
    pos = np.arange(1, w.size + 1, dtype=np.uint64)
    return (a.dtype.str, a.shape, int(np.bitwise_xor.reduce(w)), int((w * (pos | np.uint64(1))).sum(dtype=np.uint64)))


class LimbLOS(object):
    """A batch of lines of sight for the device LOS pipeline (sr_los_desc): per ray the segments it crosses,
    per segment the layer whose coefficients apply and the LOS sample points (path coordinate x [cm], number
    density nd [cm^-3], VMR of every gas) over which the Curtis-Godson column is integrated on the device.

    seg_off [n_rays+1], seg_layer [n_seg], pt_off [n_seg+1], x / nd [n_pt], vmr [n_gas, n_pt].
    Options as at the reference's call sites: LOS_order ('photon' | 'observer'), solo_absorption,
    initial_intensity: None, a temperature (Planck source, Calc_BB) or 'rad0' (given per call)."""

    def __init__(self, seg_off, seg_layer, pt_off, x, nd, vmr, col_scale=None, LOS_order='photon',
                 solo_absorption=False, initial_temperature=None):
        # (copies: the object is the geometry as given -- its device-resident form is built once, see handle())
        self.seg_off, self._so = _i(np.array(seg_off))
        self.seg_layer, self._sl = _i(np.array(seg_layer))
        self.pt_off, self._po = _i(np.array(pt_off))
        self.x, self._x = _d(np.array(x))
        self.nd, self._nd = _d(np.array(nd))
        self.vmr, self._v = _d(np.atleast_2d(np.array(vmr)))
        self.n_rays, self.n_gas = self.seg_off.size - 1, self.vmr.shape[0]
        self.n_seg, self.n_pt = self.seg_layer.size, self.x.size
        if self.vmr.shape[1] != self.n_pt or self.nd.size != self.n_pt or self.pt_off.size != self.n_seg + 1:
            raise ValueError("inconsistent LOS arrays")
        self.col_scale, self._cs = (None, None) if col_scale is None else _d(col_scale)
        if LOS_order not in ('photon', 'observer'):
            raise ValueError("LOS_order must be 'photon' or 'observer'")
        self.LOS_order, self.solo_absorption, self.initial_temperature = LOS_order, bool(solo_absorption), initial_temperature
        self._handles = {}      # insertion-ordered: least recently used first (see _keep)

    MAX_HANDLES = 8   # resident forms kept per batch: each holds pinned and device memory until it is destroyed

    def _keep(self, key, ent):
        """Insert / refresh an entry of the bounded handle cache; the least recently used forms are destroyed."""
        self._handles.pop(key, None)
        self._handles[key] = ent
        while len(self._handles) > self.MAX_HANDLES:
            old = self._handles.pop(next(iter(self._handles)))
            lib.sr_los_destroy(getattr(old, "h", old))
        return ent

    def handle(self, n_layers, grid=None, rad0=False):
        """The batch resident on the device (sr_los_create): staged, its columns integrated and the folded sweep's
        records packed once per (n_layers, grid, rad0); limb_rays and LineSet.limb_step then only launch.  The
        object IS the geometry it was built from: its arrays were copied at construction.  New VMRs go through
        set_vmr() (which updates the resident forms); other paths need a new LimbLOS."""
        gp = None if (grid is None or self.initial_temperature is None or rad0) else grid_params(grid)[:2]
        key = (int(n_layers), gp, bool(rad0))
        h = self._handles.get(key)
        if h is None:
            d = self.desc(grid, 0, rad0=rad0)
            h = C.c_void_p()
            check(lib.sr_los_create(C.byref(d), int(n_layers), C.byref(h)), "sr_los_create")
        return self._keep(key, h)

    def handle_par(self, n_layers, par_gas, par_w, grid=None, rad0=False):
        """handle() with the column parameters (par_gas [n_par], par_w [n_par, n_pt]) staged alongside
        (sr_los_create_par): the batch of a retrieval.  Keyed on the two arrays' CONTENT (a hash of their bytes: a
        caller that builds par_w afresh every iteration finds its batch again, an in-place edit makes a new one) -- or,
        for read-only arrays, on their identity (_array_key); at most MAX_HANDLES resident forms are kept, the least
        recently used is destroyed."""
        gp = None if (grid is None or self.initial_temperature is None or rad0) else grid_params(grid)[:2]
        pg_a, pg = _i(par_gas)
        pw_a, pw = _d(par_w)
        if pw_a.shape != (pg_a.size, self.n_pt):
            raise ValueError("par_w must be [n_par, n_pt]")
        key = (int(n_layers), gp, bool(rad0), _array_key(par_gas), _array_key(par_w))
        ent = self._handles.get(key)
        if ent is None:
            d = self.desc(grid, 0, rad0=rad0)
            h = C.c_void_p()
            check(lib.sr_los_create_par(C.byref(d), int(n_layers), pg_a.size, pg, pw, C.byref(h)), "sr_los_create_par")
            ent = _ParHandle(h, keep=(par_gas, par_w))
        return self._keep(key, ent).h

    def last_forward_kernel_ms(self, n_layers, par_gas, par_w, grid=None):
        """HIP-event time [ms] of the recursion (record packing + the one-sweep kernel with the bands in its epilogue) of
        the most recent retrieval_forward / _step / _loop on this batch with these parameters, made under set_timing(2)
        (sr_los_last_kernel_ms)."""
        ms = C.c_float(0.0)
        check(lib.sr_los_last_kernel_ms(self.handle_par(n_layers, par_gas, par_w, grid), C.byref(ms)), "sr_los_last_kernel_ms")
        return float(ms.value)

    def refresh_columns(self):
        """Integrate the Curtis-Godson columns of every resident form of the batch again, on the device, from the staged
        sample points (sr_los_refresh_columns: launches only)."""
        for h in self._handles.values():
            check(lib.sr_los_refresh_columns(getattr(h, "h", h), _stream_ptr()), "sr_los_refresh_columns")

    def set_vmr(self, vmr):
        """New VMRs [n_gas, n_pt] at the sample points (the next iteration of a retrieval): the host copy and every
        resident form of the batch (sr_los_set_vmr: one small copy + the column kernel; nothing else is re-staged)."""
        v = np.ascontiguousarray(np.atleast_2d(vmr), dtype=np.float64)
        if v.shape != self.vmr.shape:
            raise ValueError("vmr must be [n_gas, n_pt]")
        if self.LOS_order != 'photon' and self._handles:
            # sr_los_set_vmr refuses observer-order batches (their sample points are re-listed): checked BEFORE the
            # host copy changes, so that it and the resident forms never disagree -- drop the forms, they are rebuilt
            self.close()
        self.vmr[...] = v
        for h in self._handles.values():
            check(lib.sr_los_set_vmr(getattr(h, "h", h), self._v, _stream_ptr()), "sr_los_set_vmr")

    def close(self):
        for h in getattr(self, "_handles", {}).values():
            if lib is not None:
                lib.sr_los_destroy(getattr(h, "h", h))
        self._handles = {}

    __del__ = close

    def desc(self, grid=None, g_lo=0, rad0=False):
        d = _lib.LosDesc()
        d.n_rays, d.n_gas = self.n_rays, self.n_gas
        d.seg_off, d.seg_layer, d.pt_off, d.x, d.nd, d.vmr = self._so, self._sl, self._po, self._x, self._nd, self._v
        d.col_scale = self._cs
        d.los_order = 0 if self.LOS_order == 'photon' else 1
        d.solo_absorption = int(self.solo_absorption)
        d.init_mode, d.t_init, d.w0, d.step, d.g_lo = 0, 0.0, 0.0, 0.0, int(g_lo)
        if rad0:
            d.init_mode = 1
        elif self.initial_temperature is not None:
            if grid is None:
                raise ValueError("a Planck initial intensity needs the spectral grid")
            w0, step, _ = grid_params(grid)
            d.init_mode, d.t_init, d.w0, d.step = 2, float(self.initial_temperature), w0, step
        return d

    def columns(self):
        """[n_gas, n_seg] Curtis-Godson columns (curgod_fort_2 per segment, on the device)."""
        out = np.zeros((self.n_gas, self.n_seg))
        d = self.desc()
        check(lib.sr_los_columns(C.byref(d), out.ctypes.data_as(dp)), "sr_los_columns")
        return out


def gas_stack(coeffs):
    """The gases' coefficient tables stacked once, [n_gas, n_layers, n_pts] x 2, for callers that pass the same tables to
    many limb_rays* calls (a retrieval loop whose coefficients do not change: the stacking copies every table)."""
    return _gas_stack(coeffs)


def _gas_stack(coeffs):
    """[(abs, emi)] per gas, one (abs, emi) pair or an already stacked pair (gas_stack) -> contiguous CUDA
    [n_gas, n_layers, n_pts] x 2."""
    if isinstance(coeffs[0], torch.Tensor):
        if coeffs[0].dim() == 3:
            a, e = coeffs
            assert a.is_cuda and a.dtype == torch.float64 and e.shape == a.shape and a.is_contiguous() and e.is_contiguous()
            return a, e
        coeffs = [coeffs]
    a = torch.stack([c[0] for c in coeffs]).contiguous() if len(coeffs) > 1 else coeffs[0][0].contiguous()[None]
    e = torch.stack([c[1] for c in coeffs]).contiguous() if len(coeffs) > 1 else coeffs[0][1].contiguous()[None]
    assert a.is_cuda and a.dtype == torch.float64 and e.shape == a.shape
    return a, e


def limb_rays(coeffs, los, grid=None, g_lo=0, rad0=None, resident=True):
    """Radiances [n_rays, n_pts] of a LimbLOS batch through the gases' coefficients: columns per segment on the device,
    then the recursion.  coeffs: (abs, emi) or [(abs, emi)] per gas, CUDA [n_layers, n_pts] each.  rad0: CUDA
    [n_rays, n_pts] initial intensity (overwritten).  resident (default): through the batch's device-resident form
    (LimbLOS.handle -> sr_limb_rays_los_dev); False: staged per call (sr_limb_rays_dev)."""
    a, e = _gas_stack(coeffs)
    n_gas, n_layers, n_pts = a.shape
    if n_gas != los.n_gas:
        raise ValueError("%d coefficient sets for %d gases" % (n_gas, los.n_gas))
    rad = rad0 if rad0 is not None else torch.empty((los.n_rays, n_pts), dtype=torch.float64, device="cuda")
    assert rad.shape == (los.n_rays, n_pts) and rad.is_contiguous()
    if resident:   # the LOS staged, its columns and plan made once (sr_los_create); this call only launches
        h = los.handle(n_layers, grid, rad0=rad0 is not None)
        check(lib.sr_limb_rays_los_dev(a.data_ptr(), e.data_ptr(), n_layers, n_pts, h, int(g_lo), rad.data_ptr(), _stream_ptr()),
              "sr_limb_rays_los_dev")
        return rad
    d = los.desc(grid, g_lo, rad0=rad0 is not None)
    check(lib.sr_limb_rays_dev(C.c_void_p(a.data_ptr()), C.c_void_p(e.data_ptr()), n_layers, n_pts, C.byref(d),
                               C.c_void_p(rad.data_ptr()), _stream_ptr()), "sr_limb_rays_dev")
    return rad


def limb_rays_jacobian(coeffs, los, par_gas, par_w, grid=None, g_lo=0, rad0=None, joint=False, resident=False):
    """Radiances and d rad / d x_p [n_rays, n_par, n_pts] for VMR-profile parameters: the VMR of gas
    par_gas[p] at LOS sample point i is sum_p par_w[p, i] x_p (sr_limb_rays_jac_dev).
    joint=True: both live in ONE buffer [n_rays (1 + n_par), n_pts] -- the radiances' rows first -- which is returned
    as a third value (one hires_to_lowres call, one copy to the host, for a retrieval iteration).
    resident=True: through the batch's device-resident form with these parameter arrays (LimbLOS.handle_par, keyed on
    the arrays' identity) -- a retrieval loop that keeps its LimbLOS and calls los.set_vmr between iterations."""
    a, e = _gas_stack(coeffs)
    n_gas, n_layers, n_pts = a.shape
    par_gas_in, par_w_in = par_gas, par_w
    par_gas, pg = _i(par_gas)
    par_w, pw = _d(par_w)
    n_par = par_gas.size
    if par_w.shape != (n_par, los.n_pt):
        raise ValueError("par_w must be [n_par, n_pt]")
    buf = None
    if joint:
        if rad0 is not None:
            raise ValueError("joint=True allocates the radiances itself (no rad0)")
        buf = torch.empty((los.n_rays * (1 + n_par), n_pts), dtype=torch.float64, device="cuda")
        rad, jac = buf[:los.n_rays], buf[los.n_rays:].view(los.n_rays, n_par, n_pts)
    else:
        rad = rad0 if rad0 is not None else torch.empty((los.n_rays, n_pts), dtype=torch.float64, device="cuda")
        jac = torch.empty((los.n_rays, n_par, n_pts), dtype=torch.float64, device="cuda")
    if resident:   # the batch with its parameters staged once (sr_los_create_par); VMR updates through los.set_vmr
        h = los.handle_par(n_layers, par_gas_in, par_w_in, grid, rad0=rad0 is not None)
        check(lib.sr_limb_rays_jac_los_dev(a.data_ptr(), e.data_ptr(), n_layers, n_pts, h, int(g_lo), rad.data_ptr(), jac.data_ptr(),
                                           _stream_ptr()), "sr_limb_rays_jac_los_dev")
        return (rad, jac, buf) if joint else (rad, jac)
    d = los.desc(grid, g_lo, rad0=rad0 is not None)
    check(lib.sr_limb_rays_jac_dev(C.c_void_p(a.data_ptr()), C.c_void_p(e.data_ptr()), n_layers, n_pts, C.byref(d),
                                   n_par, pg, pw, C.c_void_p(rad.data_ptr()), C.c_void_p(jac.data_ptr()),
                                   _stream_ptr()), "sr_limb_rays_jac_dev")
    return (rad, jac, buf) if joint else (rad, jac)


def fov_factors(pixel_rot):
    """[n_pix, 7] = delta, delta^3, 2 dmax^2, edge, m2, esse, has_edge of the rotated square pixels (degrees): the geometry
    factors of spect_main_module.fov_closed_form (FOV_integr_1D, spect_main_module.py:3342-3374), in its operations."""
    out = np.zeros((len(pixel_rot), 7))
    for i, rot_deg in enumerate(pixel_rot):
        rot = abs(np.deg2rad(rot_deg))
        dmax = np.sqrt(2.0) / 2.0 * np.cos(np.pi / 4 - rot)
        delta = dmax - np.sin(rot)
        esse = 1.0 / np.cos(rot)
        edge = dmax - delta
        m2 = dmax * (dmax ** 3 - delta ** 3) / 3.0 - (dmax ** 4 - delta ** 4) / 4.0
        out[i] = (delta, delta ** 3, 2.0 * dmax ** 2, edge, m2, esse, 1.0 if edge > 1e-14 * dmax else 0.0)
    return out


def retrieval_forward(coeffs, los, par_gas, par_w, x, grid, centers_nm, widths_nm, out_units="Wm2", n_sigma=5.0, g_lo=0,
                      fov=None, buf=None):
    """The forward model of one retrieval iteration in ONE library call (sr_retrieval_forward_dev) on the batch's
    device-resident form with these parameter arrays (LimbLOS.handle_par): parameter vector x -> VMRs of the retrieved
    gases at the sample points -> columns -> radiances and parameter Jacobians -> instrument bands -> (fov: fov_factors
    of the pixels, three rays each) the closed-form field-of-view integral.  Returns numpy [n_pix or n_rays, 1 + n_par,
    n_bands]: row 0 the radiance, row 1 + p its derivative to x_p; a spectral shard (g_lo, the coefficient tables'
    width) gives its partial band integrals.  The batch's host copy of the VMRs (los.vmr) is NOT updated: call
    los.set_vmr when the loop ends.  buf: scratch from an earlier call (returned as second value)."""
    a, e = _gas_stack(coeffs)
    n_gas, n_layers, n_pts = a.shape
    w0, step, n = grid_params(grid)
    x_a, xp = _d(x)
    n_par = x_a.size
    centers_nm, cp = _d(centers_nm)
    widths_nm, wp = _d(widths_nm)
    if widths_nm.size != centers_nm.size:
        raise ValueError("{} spectral widths for {} grid points".format(widths_nm.size, centers_nm.size))
    if not (0 <= g_lo and g_lo + n_pts <= n):
        raise ValueError("the coefficient tables cover grid points outside the grid")
    h = los.handle_par(n_layers, par_gas, par_w, grid)
    if np.asarray(par_gas).size != n_par:
        raise ValueError("x must hold one value per parameter")
    n_out = los.n_rays
    fp = None
    if fov is not None:
        fov, fp = _d(fov)
        if los.n_rays % 3 or fov.shape != (los.n_rays // 3, 7):
            raise ValueError("fov must be [n_rays / 3, 7] (three rays per pixel)")
        n_out = los.n_rays // 3
    if buf is None or buf.shape != (los.n_rays * (1 + n_par), n_pts):
        buf = torch.empty((los.n_rays * (1 + n_par), n_pts), dtype=torch.float64, device="cuda")
    out = np.empty((n_out, 1 + n_par, centers_nm.size))
    check(lib.sr_retrieval_forward_dev(a.data_ptr(), e.data_ptr(), n_layers, n_pts, h, int(g_lo), xp, w0, step, cp, wp,
                                       centers_nm.size, float(n_sigma), _UNITS[out_units], fp, buf.data_ptr(),
                                       out.ctypes.data_as(dp), _stream_ptr()), "sr_retrieval_forward_dev")
    return out, buf


class OeProblem(object):
    """What sr_retrieval_step_dev needs of a retrieval besides the forward model, marshalled once per loop: observations,
    noise and mask of the pixels (pixel-major, band-minor, as genvec concatenates them), the inverse a-priori
    covariance, the a-priori vector, the Levenberg-Marquardt factor."""

    def __init__(self, obs, noise, mask, sa_inv, x_apriori, lambda_lm):
        self.obs, self._o = _d(np.asarray(obs, float).reshape(-1))
        self.noise, self._n = _d(np.asarray(noise, float).reshape(-1))
        self.mask = None if mask is None else np.ascontiguousarray(np.asarray(mask).reshape(-1), dtype=np.uint8)
        self.sa_inv, self._s = _d(sa_inv)
        self.x_ap, self._x = _d(x_apriori)
        n_par = self.x_ap.size
        if self.sa_inv.shape != (n_par, n_par) or self.noise.size != self.obs.size or (self.mask is not None and self.mask.size != self.obs.size):
            raise ValueError("inconsistent optimal-estimation problem")
        self.desc = _lib.OeDesc(self.obs.size, self._o, self._n,
                                None if self.mask is None else self.mask.ctypes.data_as(C.POINTER(C.c_uint8)), self._s, self._x,
                                float(lambda_lm))
        self.n_par = n_par


def retrieval_step(coeffs, los, par_gas, par_w, x, grid, centers_nm, widths_nm, oe, out_units="Wm2", n_sigma=5.0, fov=None,
                   buf=None):
    """retrieval_forward + the rest of the iteration in the same library call (sr_retrieval_step_dev): chi square of the
    pixels against the observations and the Levenberg-Marquardt step of the optimal-estimation algebra
    (spect_main_module.inversion_algebra) on the band spectra and Jacobians the call brings to the host.  oe: OeProblem.
    Returns (out [n_pix, 1 + n_par, n_bands], chi_sum, n_used, dx [n_par], S_x, AVK [n_par, n_par], buf).  Single process
    only (a shard's band integrals are partial)."""
    a, e = _gas_stack(coeffs)
    n_gas, n_layers, n_pts = a.shape
    w0, step, n = grid_params(grid)
    if n_pts != n:
        raise ValueError("retrieval_step needs the whole grid (a spectral shard's band integrals are partial)")
    x_a, xp = _d(x)
    n_par = x_a.size
    centers_nm, cp = _d(centers_nm)
    widths_nm, wp = _d(widths_nm)
    h = los.handle_par(n_layers, par_gas, par_w, grid)
    if np.asarray(par_gas).size != n_par or oe.n_par != n_par or los.n_rays % 3:
        raise ValueError("x, the parameters and the problem must agree; three rays per pixel")
    n_pix = los.n_rays // 3
    fp = None
    if fov is not None:
        fov, fp = _d(fov)
        if fov.shape != (n_pix, 7):
            raise ValueError("fov must be [n_rays / 3, 7]")
    if buf is None or buf.shape != (los.n_rays * (1 + n_par), n_pts):
        buf = torch.empty((los.n_rays * (1 + n_par), n_pts), dtype=torch.float64, device="cuda")
    out = np.empty((n_pix, 1 + n_par, centers_nm.size))
    chi, n_used = C.c_double(0.0), C.c_int32(0)
    dx, s_x, avk = np.empty(n_par), np.empty((n_par, n_par)), np.empty((n_par, n_par))
    check(lib.sr_retrieval_step_dev(a.data_ptr(), e.data_ptr(), n_layers, n_pts, h, 0, xp, w0, step, cp, wp, centers_nm.size,
                                    float(n_sigma), _UNITS[out_units], fp, buf.data_ptr(), out.ctypes.data_as(dp), C.byref(oe.desc),
                                    C.byref(chi), C.byref(n_used), dx.ctypes.data_as(dp), s_x.ctypes.data_as(dp),
                                    avk.ctypes.data_as(dp), _stream_ptr()), "sr_retrieval_step_dev")
    return out, chi.value, n_used.value, dx, s_x, avk, buf


def retrieval_loop(coeffs, los, par_gas, par_w, x0, grid, centers_nm, widths_nm, oe, positive, n_dof_par, chi_threshold=0.01,
                   max_it=10, out_units="Wm2", n_sigma=5.0, fov=None, buf=None):
    """The whole loop of a retrieval whose coefficient spectra stay fixed, in one library call (sr_retrieval_loop_dev):
    retrieval_step per iteration, reduced chi square, the stopping rule and the update with the positivity rule of
    spect_main_module.inversion_fast_limb.  positive [n_par] bool: constrain_positive of every parameter; n_dof_par: the
    parameters in use.  Returns (out of the last iteration [n_pix, 1 + n_par, n_bands], chi history [n_it], x history
    [n_updates + 1, n_par], stop '' | 'converged' | 'raised', S_x, AVK (None when no update was applied), buf)."""
    a, e = _gas_stack(coeffs)
    n_gas, n_layers, n_pts = a.shape
    w0, step, n = grid_params(grid)
    if n_pts != n:
        raise ValueError("retrieval_loop needs the whole grid (a spectral shard's band integrals are partial)")
    x_a, xp = _d(x0)
    n_par = x_a.size
    centers_nm, cp = _d(centers_nm)
    widths_nm, wp = _d(widths_nm)
    h = los.handle_par(n_layers, par_gas, par_w, grid)
    if np.asarray(par_gas).size != n_par or oe.n_par != n_par or los.n_rays % 3:
        raise ValueError("x, the parameters and the problem must agree; three rays per pixel")
    n_pix = los.n_rays // 3
    fp = None
    if fov is not None:
        fov, fp = _d(fov)
        if fov.shape != (n_pix, 7):
            raise ValueError("fov must be [n_rays / 3, 7]")
    if buf is None or buf.shape != (los.n_rays * (1 + n_par), n_pts):
        buf = torch.empty((los.n_rays * (1 + n_par), n_pts), dtype=torch.float64, device="cuda")
    pos = np.ascontiguousarray(np.asarray(positive, dtype=bool).reshape(-1), dtype=np.uint8)
    if pos.size != n_par:
        raise ValueError("positive must be [n_par]")
    max_it = int(max_it)
    lp = _lib.LoopDesc(max_it, float(chi_threshold), pos.ctypes.data_as(C.POINTER(C.c_uint8)), int(n_dof_par))
    out = np.zeros((n_pix, 1 + n_par, centers_nm.size))
    chi_hist, x_hist = np.zeros(max(max_it, 1)), np.zeros((max_it + 1, n_par))
    n_it, stop = C.c_int32(0), C.c_int32(0)
    s_x, avk = np.zeros((n_par, n_par)), np.zeros((n_par, n_par))
    check(lib.sr_retrieval_loop_dev(a.data_ptr(), e.data_ptr(), n_layers, n_pts, h, 0, xp, w0, step, cp, wp, centers_nm.size,
                                    float(n_sigma), _UNITS[out_units], fp, buf.data_ptr(), out.ctypes.data_as(dp), C.byref(oe.desc),
                                    C.byref(lp), chi_hist.ctypes.data_as(dp), x_hist.ctypes.data_as(dp), C.byref(n_it),
                                    C.byref(stop), s_x.ctypes.data_as(dp), avk.ctypes.data_as(dp), _stream_ptr()),
          "sr_retrieval_loop_dev")
    n_upd = n_it.value if stop.value == 0 else n_it.value - 1
    return (out, chi_hist[:n_it.value].copy(), x_hist[:n_upd + 1].copy(), ["", "converged", "raised"][stop.value],
            s_x if n_upd > 0 else None, avk if n_upd > 0 else None, buf)


def limb_rays_layer_jacobian(coeffs, dcoeffs, los, grid=None, g_lo=0):
    """d rad / d (one scalar per layer) [n_rays, n_layers, n_pts]; dcoeffs like coeffs: d(abs, emi of layer k) /
    d(parameter of layer k) per gas (sr_limb_rays_jac_layer_dev)."""
    a, e = _gas_stack(coeffs)
    da, de = _gas_stack(dcoeffs)
    n_gas, n_layers, n_pts = a.shape
    assert da.shape == a.shape
    jac = torch.empty((los.n_rays, n_layers, n_pts), dtype=torch.float64, device="cuda")
    d = los.desc(grid, g_lo)
    check(lib.sr_limb_rays_jac_layer_dev(C.c_void_p(a.data_ptr()), C.c_void_p(e.data_ptr()), C.c_void_p(da.data_ptr()),
                                         C.c_void_p(de.data_ptr()), n_layers, n_pts, C.byref(d),
                                         C.c_void_p(jac.data_ptr()), _stream_ptr()), "sr_limb_rays_jac_layer_dev")
    return jac


def limb_rays_jacobians(coeffs, los, dcoeffs=None, par_gas=None, par_w=None, grid=None, g_lo=0, want_rad=True,
                        seg_jac_row=None, n_jac_rows=None):
    """Radiances, per-layer Jacobian (dcoeffs given: d(abs, emi of layer k)/d(scalar of layer k) per gas) and
    column-parameter Jacobian (par_gas / par_w given, as limb_rays_jacobian) in ONE pass over each ray
    (sr_limb_rays_jacobians_dev).  Returns (rad | None, jac_layer | None, jac_par | None).
    seg_jac_row [n_seg] (with n_jac_rows): the per-layer Jacobian row of every segment when it is not the segment's
    coefficient row (3-D paths: a coefficient row per LOS step, the Jacobian per altitude layer)."""
    a, e = _gas_stack(coeffs)
    n_gas, n_layers, n_pts = a.shape
    if n_gas != los.n_gas:
        raise ValueError("%d coefficient sets for %d gases" % (n_gas, los.n_gas))
    da = de = jl = jp = rad = None
    if dcoeffs is not None:
        da, de = _gas_stack(dcoeffs)
        assert da.shape == a.shape
        jl = torch.empty((los.n_rays, n_layers if seg_jac_row is None else int(n_jac_rows), n_pts), dtype=torch.float64,
                         device="cuda")
    n_par, pg, pw = 0, None, None
    sjr, sjp = (None, None) if seg_jac_row is None else _i(seg_jac_row)
    if sjr is not None and sjr.size != los.n_seg:
        raise ValueError("seg_jac_row must be [n_seg]")
    if par_gas is not None:
        par_gas, pg = _i(par_gas)
        par_w, pw = _d(par_w)
        n_par = par_gas.size
        if par_w.shape != (n_par, los.n_pt):
            raise ValueError("par_w must be [n_par, n_pt]")
        jp = torch.empty((los.n_rays, n_par, n_pts), dtype=torch.float64, device="cuda")
    if want_rad:
        rad = torch.empty((los.n_rays, n_pts), dtype=torch.float64, device="cuda")
    d = los.desc(grid, g_lo)
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    check(lib.sr_limb_rays_jacobians_dev(ptr(a), ptr(e), ptr(da), ptr(de), n_layers, n_pts, C.byref(d), sjp,
                                         0 if sjr is None else int(n_jac_rows), n_par, pg, pw, ptr(rad), ptr(jl), ptr(jp),
                                         _stream_ptr()), "sr_limb_rays_jacobians_dev")
    return rad, jl, jp


def radiance_jacobian(abs_c, emi_c, seg_off, seg_layer, seg_col, dcol_dpar):
    """Radiances [n_rays, n_pts] and d rad / d x_p [n_rays, n_par, n_pts] for parameters on which the
    segment columns depend linearly: dcol_dpar[s, p] = d col_s / d x_p (sr_radiance_jac_dev)."""
    assert abs_c.is_cuda and abs_c.dtype == torch.float64 and abs_c.is_contiguous()
    assert emi_c.shape == abs_c.shape and emi_c.is_contiguous()
    n_layers, n_pts = abs_c.shape
    seg_off, op = _i(seg_off)
    seg_layer, lp = _i(seg_layer)
    seg_col, cp = _d(seg_col)
    dcol, dpp = _d(dcol_dpar)
    if dcol.ndim != 2 or dcol.shape[0] != seg_col.size:
        raise ValueError("dcol_dpar must be [n_seg, n_par]")
    n_rays, n_par = seg_off.size - 1, dcol.shape[1]
    rad = torch.empty((n_rays, n_pts), dtype=torch.float64, device="cuda")
    jac = torch.empty((n_rays, n_par, n_pts), dtype=torch.float64, device="cuda")
    check(lib.sr_radiance_jac_dev(C.c_void_p(abs_c.data_ptr()), C.c_void_p(emi_c.data_ptr()), n_layers, n_pts,
                                  n_rays, op, lp, cp, dpp, n_par, C.c_void_p(rad.data_ptr()),
                                  C.c_void_p(jac.data_ptr()), _stream_ptr()), "sr_radiance_jac_dev")
    return rad, jac


def radiance_layer_jacobian(abs_c, emi_c, dabs, demi, seg_off, seg_layer, seg_col):
    """d rad / d (one scalar per layer) [n_rays, n_layers, n_pts], the scalar acting through the layer's own
    coefficients: dabs, demi [n_layers, n_pts] = d(abs, emi of layer k)/d(parameter of layer k)
    (sr_radiance_jac_layer_dev)."""
    for t in (abs_c, emi_c, dabs, demi):
        assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.shape == abs_c.shape
    n_layers, n_pts = abs_c.shape
    seg_off, op = _i(seg_off)
    seg_layer, lp = _i(seg_layer)
    seg_col, cp = _d(seg_col)
    n_rays = seg_off.size - 1
    jac = torch.empty((n_rays, n_layers, n_pts), dtype=torch.float64, device="cuda")
    check(lib.sr_radiance_jac_layer_dev(C.c_void_p(abs_c.data_ptr()), C.c_void_p(emi_c.data_ptr()),
                                        C.c_void_p(dabs.data_ptr()), C.c_void_p(demi.data_ptr()), n_layers, n_pts,
                                        n_rays, op, lp, cp, C.c_void_p(jac.data_ptr()), _stream_ptr()),
          "sr_radiance_jac_layer_dev")
    return jac


def calc_radtran_steps(gases, z, temps, press, z_tans, radtran_opt=None, R=2575.0, n_sub=3, **los_opts):
    """Adaptive LOS stepping + the coefficient rows of the steps, with the reference's knobs: radtran_opt =
    dict(max_T_variation=[K], max_Plog_variation=[ln P], max_opt_depth=...) as its drivers pass them to
    LineOfSight.calc_radtran_steps (spect_main_module.py:2746-2767; radtran_test_CO.py:184-186;
    spect_radtran_test.py:175).  gases: [dict(lineset=LineSet, vmr=[n_levels], iso_ratio=1.0, tvib=[n_lev, n_levels] | None)].
    Geometry by geometry.calc_radtran_steps (temperature / log-pressure bounds, vectorised); the optical-depth bound is
    checked on the device: the steps' coefficient rows at their own (P, T) -- abscoeff_layers on the step list --, per
    step the largest absorption coefficient over the grid times the step's Curtis-Godson column (sr_los_columns),
    summed over the gases; steps above the bound are halved and only then re-evaluated.
    Returns dict(L=geometry dict, los=LimbLOS, coeffs=[(abs, emi)] per gas on the step rows [n_steps, n_grid])."""
    from . import geometry as geo, synthetic as syn
    opt = dict(radtran_opt or {})
    z, temps, press = (np.asarray(v, float) for v in (z, temps, press))
    nd = syn.number_density(press, temps)
    vm = [np.asarray(g["vmr"], float) for g in gases]
    state = {}

    def rows(L):
        key = (len(L["step_temp"]), float(L["step_temp"].sum()), float(L["x"].sum()))
        if state.get("key") != key:
            co = []
            for g in gases:
                tv = None
                if g.get("tvib") is not None:     # vibrational temperatures linear in altitude between the levels, like T
                    tvl = np.asarray(g["tvib"], float)
                    zz = np.append(z, z[-1] + (z[-1] - z[-2]))
                    tv = np.array([np.interp(L["step_alt"], zz, np.append(t, t[-1])) for t in tvl])
                co.append(g["lineset"].abscoeff_layers(L["step_temp"], L["step_pres"], tvib=tv))
            los = LimbLOS(L["seg_off"], L["seg_layer"], L["pt_off"], L["x"], L["nd"], L["vmr"],
                          col_scale=[g.get("iso_ratio", 1.0) for g in gases], **los_opts)
            state.update(key=key, coeffs=co, los=los)
        return state["coeffs"], state["los"]

    def opt_depth_of(L):
        co, los = rows(L)
        col = los.columns()                                   # [n_gas, n_steps]
        amax = np.array([c[0].abs().amax(dim=1).cpu().numpy() for c in co])
        return (amax * col).sum(axis=0)

    L = geo.calc_radtran_steps(z, temps, press, nd, vm, z_tans, R=R, n_sub=n_sub,
                               max_T_variation=opt.get("max_T_variation"), max_Plog_variation=opt.get("max_Plog_variation"),
                               max_opt_depth=opt.get("max_opt_depth"), opt_depth_of=opt_depth_of if opt.get("max_opt_depth") else None)
    co, los = rows(L)
    return dict(L=L, los=los, coeffs=co)


def mix_gases(coeffs, ratios):
    """Coefficients of a gas mixture on the column scale of a reference absorber: the column of gas g in
    a segment of layer k is ratios[g][k] times the reference column (VMR_g / VMR_ref of the layer), so
    tau = (sum_g ratios[g][k] abs_g[k]) * u_ref.  coeffs: [(abs_g, emi_g)] CUDA [n_layers, n_pts] on one
    grid; ratios: [n_gas][n_layers].  Returns (abs_mix, emi_mix)."""
    a_mix = e_mix = None
    for (a, e), r in zip(coeffs, ratios):
        w = torch.as_tensor(np.ascontiguousarray(r, dtype=np.float64), device=a.device)[:, None]
        a_mix = a * w if a_mix is None else a_mix + a * w
        e_mix = e * w if e_mix is None else e_mix + e * w
    return a_mix.contiguous(), e_mix.contiguous()


def gas_layer_jacobian(abs_mix, emi_mix, abs_g, emi_g, dratio_dx, seg_off, seg_layer, seg_col):
    """d rad / d x_k [n_rays, n_layers, n_pts] for one scalar per layer that scales gas g's amount in
    that layer inside a mixture (its VMR): the mixture's coefficients depend on it through
    d abs_mix[k]/d x_k = dratio_dx[k] * abs_g[k] (same for emi), which is the layer Jacobian's input.
    Profile parameters follow by the chain rule: J_p = sum_k mask_p[k] * J_k."""
    w = torch.as_tensor(np.ascontiguousarray(dratio_dx, dtype=np.float64), device=abs_g.device)[:, None]
    return radiance_layer_jacobian(abs_mix, emi_mix, (abs_g * w).contiguous(), (emi_g * w).contiguous(),
                                   seg_off, seg_layer, seg_col)


def coefficients_dT(ls, temps, press, tvib=None, q_part=None, g_lo=0, g_hi=None, scheme="central", dT=None,
                    coeffs=None, frozen=True):
    """(abs, emi) at T and their derivatives with respect to each layer's kinetic temperature (pressure and, in
    non-LTE, the vibrational temperatures held fixed; with q_part=None the partition sum follows T).  The reference
    has no temperature Jacobian (spect_main_module.py:300-306 is commented out): build's definition, by finite
    differences of the coefficient op.

    frozen (default): the perturbed ops place their Humlicek region boundaries as at T (LineSet.set_bounds_temps).
    The regions disagree by 1e-5..1e-4 at their index-computed seams; with boundaries that move with T a difference
    quotient carries a spike of (1e-5 y) / dT wherever a seam crosses a point.  What remains non-smooth in T is the
    reference's single-precision cmplx(ry, -rx) (lineshape.f:529): a staircase of ~1e-7 of a core value, i.e. noise of
    ~1e-7 |c| / dT in any quotient -- which is what bounds dT from below.
      scheme "central" (default, THREE coefficient ops): (c(T + dT) - c(T - dT)) / 2 dT, dT = 0.05 K: ~3e-5 of a
        layer's largest derivative (2e-4 with moving boundaries, frozen=False: rounds 1-3's definition).
      scheme "forward" (TWO ops): (c(T + dT) - c(T)) / dT, dT = 0.002 K: ~5e-4 (truncation dT/2 |c''|, with c'' / c'
        up to 0.3 / K from level populations and Doppler cores, against the staircase noise): good for a
        Gauss-Newton step, a third cheaper.
    tests/test_gpu_configs.py::test_temperature_derivative_schemes measures the three against each other."""
    temps = np.ascontiguousarray(temps, dtype=np.float64)
    kw = dict(tvib=tvib, q_part=q_part, g_lo=g_lo, g_hi=g_hi)
    if scheme not in ("forward", "central"):
        raise ValueError("scheme must be 'forward' or 'central'")
    if coeffs is None:
        coeffs = ls.abscoeff_layers(temps, press, **kw)
    if frozen:
        ls.set_bounds_temps(temps)
    try:
        if scheme == "forward":
            dT = 0.002 if dT is None else dT
            a_p, e_p = ls.abscoeff_layers(temps + dT, press, **kw)
            return coeffs, ((a_p - coeffs[0]) / dT, (e_p - coeffs[1]) / dT)
        dT = 0.05 if dT is None else dT
        a_p, e_p = ls.abscoeff_layers(temps + dT, press, **kw)
        a_m, e_m = ls.abscoeff_layers(temps - dT, press, **kw)
        return coeffs, ((a_p - a_m) / (2.0 * dT), (e_p - e_m) / (2.0 * dT))
    finally:
        if frozen:
            ls.set_bounds_temps(None)


def temperature_jacobian(ls, temps, press, seg_off, seg_layer, seg_col, tvib=None, q_part=None, dT=0.05,
                         g_lo=0, g_hi=None, coeffs=None):
    """d rad / d T_k [n_rays, n_layers, n_pts] for the kinetic temperature of every layer (pressure,
    columns and, in non-LTE, the vibrational temperatures held fixed): the coefficient op at T + dT
    and T - dT (coefficients_dT: central differences of the layer's own abs / emi with the region boundaries
    frozen at T; with q_part=None the partition sum follows T), then the forward sensitivity of the radiance
    recursion.  `coeffs` = (abs, emi) at T
    if already computed.  The reference has no temperature Jacobian: build's definition."""
    coeffs, (dabs, demi) = coefficients_dT(ls, temps, press, tvib=tvib, q_part=q_part, g_lo=g_lo, g_hi=g_hi,
                                           scheme="central", dT=dT, coeffs=coeffs)
    return radiance_layer_jacobian(coeffs[0], coeffs[1], dabs, demi, seg_off, seg_layer, seg_col)


def set_points_per_lane(p):
    check(lib.sr_set_points_per_lane(int(p)), "sr_set_points_per_lane")


def set_level_route(multi_channel):
    """1 (default): level tables (glevel_pairs, gcoeff_levels) by the multi-channel pass -- every line once --; 0: one
    coefficient op per level (sr_set_level_route)."""
    check(lib.sr_set_level_route(int(multi_channel)), "sr_set_level_route")


def set_overlap(on):
    """1 (default): the zones kernel runs beside the far-field kernel on an internal stream;
    0: kernels one after the other (per-kernel times in last_kernel_ms)."""
    check(lib.sr_set_overlap(int(on)), "sr_set_overlap")


def far_field_truncation_bound():
    """18 theta^-(degree + 1): what the far-field mode may differ by from the exact mode, relative to a line's own
    contribution (sr_far_field_truncation_bound; 1.6e-11 as built by default)."""
    return float(lib.sr_far_field_truncation_bound())


def far_field_degree(theta=4):
    """The expansion degree the library was built with (from its truncation bound 18 theta^-(degree + 1))."""
    import math
    return int(round(math.log(18.0 / far_field_truncation_bound()) / math.log(theta))) - 1


def set_band_fusion(on):
    """1 (default): a retrieval iteration's recursion kernel integrates the instrument bands itself; 0: spectra, then the
    instrument step's own kernels (sr_set_band_fusion; the check and the A/B partner)."""
    check(lib.sr_set_band_fusion(int(bool(on))), "sr_set_band_fusion")


def set_jac_layer_mode(forward):
    """0 (default): Jacobians of the device LOS pipeline in one pass -- folded (a ray's two segments of a shell together,
    every value stored once) where the rays share their coefficient rows (1-D atmospheres), else in path order; 1: the
    forward-sensitivity kernels; 2: path order, one ray per thread, always; 3: path order, two rays per thread sharing
    a shell's loads (sr_set_jac_layer_mode)."""
    check(lib.sr_set_jac_layer_mode(int(forward)), "sr_set_jac_layer_mode")


def last_limb_route():
    """Which recursion kernel the last limb_rays call of this thread launched: 1 path order, 2 the folded sweep
    (sr_last_limb_route; diagnostic)."""
    return int(lib.sr_last_limb_route())


FAR_FIELD_DEFAULT = 3  # the library's default far-field mode (sr_set_far_field)


def set_far_field(on):
    """1: far wings by per-line local expansions; 2: by box pairs (multipole moments of the lines of a source
    box, translated to the local expansions of the well-separated target boxes); 3 (default): box pairs, sparse line
    sets (< 0.35 lines per grid point: per-level sub-linesets) by per-line expansions; 0: every evaluation exact."""
    check(lib.sr_set_far_field(int(on)), "sr_set_far_field")


_UNITS = {"Wm2": 0, "ergscm2": 1, "nWcm2": 2}


def hires_to_lowres(rad, grid, centers_nm, widths_nm, out_units="Wm2", n_sigma=5.0, g_lo=None):
    """Gaussian-ILS degradation of hi-res spectra (CUDA float64 [n_rays, n_grid], 'ergscm2' on the
    cm-1 grid) onto low-resolution bands given in nm: SpectralIntensity.hires_to_lowres
    (spect_classes.py:1180-1191).  Returns numpy [n_rays, n_bands] in out_units.
    g_lo given: rad holds the grid points g_lo .. g_lo + rad.shape[-1] - 1 only (a spectral shard) and the PARTIAL band
    integrals over them are returned (sr_hires_to_lowres_shard_dev); `grid` is always the whole grid.  Without g_lo
    the spectrum must cover the whole grid (a mis-sized spectrum is an error, not a partial integral)."""
    w0, step, n = grid_params(grid)
    n_sh = rad.shape[-1]
    if g_lo is None:
        if n_sh != n:
            raise ValueError("spectrum of %d points on a grid of %d (pass g_lo for the partial integrals of a shard)" % (n_sh, n))
        g_lo = 0
    assert rad.is_cuda and rad.dtype == torch.float64 and rad.is_contiguous() and 0 <= g_lo and g_lo + n_sh <= n
    rad2 = rad.reshape(-1, n_sh)
    centers_nm, cp = _d(centers_nm)
    widths_nm, wp = _d(widths_nm)
    if widths_nm.size != centers_nm.size:
        raise ValueError("{} spectral widths for {} grid points".format(widths_nm.size, centers_nm.size))
    out = np.zeros((rad2.shape[0], centers_nm.size))
    check(lib.sr_hires_to_lowres_shard_dev(C.c_void_p(rad2.data_ptr()), rad2.shape[0], n_sh, int(g_lo), w0, step, cp, wp,
                                           centers_nm.size, float(n_sigma), _UNITS[out_units], out.ctypes.data_as(dp),
                                           _stream_ptr()), "sr_hires_to_lowres_shard_dev")
    return out


def set_table_budget(n_bytes):
    """Bytes of per-(line, layer) record tables per launch; longer layer stacks run in batches."""
    check(lib.sr_set_table_budget(int(n_bytes)), "sr_set_table_budget")
