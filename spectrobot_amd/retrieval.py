"""Forward model + retrieval loop around the hot path: the structure of the reference's
`inversion_fast_limb` / `radtrans` (spect_main_module.py:2598-2987, 2990-3287) on the GPU engine.

The reference's loop is, per iteration: LOS stepping and coefficients per LOS (`calc_radtran_steps`),
radiances + derivatives per spectral split (`radtran_fast`), `hires_to_lowres`, `FOV_integr_1D` over the
three LOS of every pixel, `chicalc`, the stopping rule (|dchi2|/chi2 < chi_threshold, or chi2 raised),
`inversion_algebra`, `add_clim(profile)`.  Planet / atmosphere / LineOfSight / VIMSPixel are classes of the
absent spect_base_module; `LimbScene` is the minimal stand-in the loop needs (a 1-D atmosphere on altitude
levels, gases with VMR profiles, limb pixels with three lines of sight each).  What runs where:

  per scene      engine.LineSet per gas (lines -> HBM once)
  per iteration  coefficients of every gas whose temperature / vibrational temperatures changed (once if
                 only VMRs are retrieved, as in the reference's drivers: radtran_3D_ch4.py:127-170),
                 engine.limb_rays_jacobian: columns on the device, radiances and d rad / d x_p for ALL LOS
                 of ALL pixels in one launch, engine.hires_to_lowres for radiances and derivatives
  host           FOV integration, chi square, the n_par x n_par algebra (smm mirror), bookkeeping
"""
import numpy as np

from . import engine
from . import spect_main_module as smm
from . import synthetic as syn


STEP_IN_ONE_CALL = True   # inversion_fast_limb: forward model + chi square + the optimal-estimation algebra of an iteration
                          # in one library call (engine.retrieval_step); False: the algebra in numpy (the check)
LOOP_IN_ONE_CALL = True   # ... and the loop around it too (engine.retrieval_loop) when the coefficient spectra stay fixed; False: the
                          # loop below, an engine.retrieval_step per iteration (the check)
ONE_CALL = True   # simulate(arrays=True): the iteration's forward model in one library call (engine.retrieval_forward);
                  # False: columns + Jacobians, instrument step and FOV as separate calls (the A/B partner and the check)


class Spectrum(object):
    """Low-resolution spectrum holder with the attributes the smm algebra reads (.spectrum, .spectral_grid)."""

    def __init__(self, spectrum, grid=None):
        self.spectrum = np.asarray(spectrum, dtype=float)
        self.spectral_grid = None if grid is None else _Grid(grid)


class _Grid(object):
    def __init__(self, g):
        self.grid = np.asarray(g, dtype=float)


class Gas(object):
    """One absorber: its line set on the scene's grid, isotopic abundance, VMR profile on the altitude levels
    and (non-LTE) vibrational temperatures [n_levels, n_layers]."""

    def __init__(self, name, lineset, vmr, iso_ratio=1.0, tvib=None):
        self.name, self.lineset, self.iso_ratio = name, lineset, float(iso_ratio)
        self.vmr = np.asarray(vmr, dtype=float)
        self.tvib = tvib
        self.coeffs = None      # (abs, emi) CUDA [n_layers, n_grid], computed on demand

    def add_clim(self, profile):
        """New VMR profile (planet.gases[gas].add_clim, spect_main_module.py:2625, 2985)."""
        self.vmr = np.asarray(profile, dtype=float)


class LimbPixel(object):
    """A limb pixel: tangent altitude of its centre LOS, half extent of the FOV in altitude (the lower /
    upper LOS are at alt -+ fov_half: pix.low_LOS() / LOS() / up_LOS(), spect_main_module.py:2704-2706),
    rotation of the square pixel, observation / noise / mask on the low-resolution bands."""

    def __init__(self, limb_tg_alt, fov_half=0.0, pixel_rot=0.0, observation=None, noise=None, mask=None):
        self.limb_tg_alt, self.fov_half, self.pixel_rot = float(limb_tg_alt), float(fov_half), float(pixel_rot)
        self.observation, self.noise, self.mask = observation, noise, mask

    def los_alts(self):
        return [self.limb_tg_alt - self.fov_half, self.limb_tg_alt, self.limb_tg_alt + self.fov_half]


class LimbScene(object):
    """1-D atmosphere on altitude levels z [km] (temps [K], press [hPa]), gases, spectral grid, instrument bands
    (centres / Gaussian sigmas in nm)."""

    def __init__(self, grid, z, temps, press, gases, bands_nm, widths_nm, R=2575.0, n_sub=3, out_units="Wm2"):
        self.grid = np.asarray(grid, dtype=float)
        self.z, self.temps, self.press = (np.asarray(v, dtype=float) for v in (z, temps, press))
        self.nd = syn.number_density(self.press, self.temps)
        self.gases = list(gases)
        self.bands_nm, self.widths_nm = np.asarray(bands_nm, float), np.asarray(widths_nm, float)
        self.R, self.n_sub, self.out_units = R, n_sub, out_units

    def gas(self, name):
        return [g for g in self.gases if g.name == name][0]

    def coefficients(self, refresh=False, g_lo=0, g_hi=None):
        """(abs, emi) of every gas at the layer stack over grid points [g_lo, g_hi); cached per shard: only VMRs change
        between the iterations of a VMR retrieval (refresh=True recomputes, e.g. when temperatures are retrieved)."""
        g_hi = len(self.grid) if g_hi is None else int(g_hi)
        for g in self.gases:
            if g.coeffs is None or refresh or getattr(g, "coeffs_shard", None) != (g_lo, g_hi):
                g.coeffs = g.lineset.abscoeff_layers(self.temps, self.press, tvib=g.tvib, g_lo=g_lo, g_hi=g_hi)
                g.coeffs_shard = (g_lo, g_hi)
        return [g.coeffs for g in self.gases]

    def coefficient_stack(self, refresh=False, g_lo=0, g_hi=None):
        """coefficients() as the stacked pair the limb_rays* calls work on (engine.gas_stack), re-stacked only when a
        gas's tables were recomputed: a VMR retrieval passes the same tables in every iteration."""
        co = self.coefficients(refresh=refresh, g_lo=g_lo, g_hi=g_hi)
        key = tuple(id(c[0]) for c in co) + tuple(id(c[1]) for c in co)
        if getattr(self, "_stack_key", None) != key:
            self._stack, self._stack_key = engine.gas_stack(co), key
            self._stack_of = co          # keeps the ids alive
        return self._stack

    def los(self, tangent_alts, update=True, **opts):
        """engine.LimbLOS of rays with the given tangent altitudes (photon order) + the sample altitudes.  The geometry
        (paths, sample points, densities) depends on the altitudes alone and is kept; between the iterations of a
        retrieval only the VMRs at the sample points change."""
        # (the key holds everything the geometry is made from: altitudes of the rays, the level grid, the densities, the
        # sub-stepping and the radius -- by content)
        key = (tuple(float(a) for a in tangent_alts), np.asarray(self.z, float).tobytes(), np.asarray(self.nd, float).tobytes(),
               int(self.n_sub), float(self.R))
        geo = getattr(self, "_los_geo", None)
        if geo is None or geo[0] != key:
            L = syn.limb_los(self.z, self.nd, [g.vmr for g in self.gases], tangent_alts, R=self.R, n_sub=self.n_sub)
            top = self.z[-1] + (self.z[-1] - self.z[-2])
            geo = self._los_geo = (key, L, np.append(self.z, top))
        _, L, zz = geo
        # the batch itself is kept too (photon order, no options): its device-resident form (engine.LimbLOS.handle_par)
        # then needs only the new VMRs (LimbLOS.set_vmr: one small copy + the column kernel) -- or nothing at all
        # (update=False) when the caller sets them on the device from the parameter vector (engine.retrieval_forward)
        kept = getattr(self, "_los_obj", None)
        if not update and not opts and kept is not None and kept[0] == key:
            return kept[1], L["alt"]
        # (geometry._profiles: every VMR linear in altitude between the levels, constant above the last one)
        vmr = np.array([np.interp(L["alt"], zz, np.append(g.vmr, g.vmr[-1])) for g in self.gases])
        if not opts and kept is not None and kept[0] == key:
            kept[1].set_vmr(vmr)
            return kept[1], L["alt"]
        los = engine.LimbLOS(L["seg_off"], L["seg_layer"], L["pt_off"], L["x"], L["nd"], vmr,
                             col_scale=[g.iso_ratio for g in self.gases], **opts)
        if not opts:
            self._los_obj = (key, los)
        return los, L["alt"]

    def profile_weights(self, bayes_set, alt):
        """par_gas [n_par], par_w [n_par, n_pt]: the masks of every retrieved parameter at the LOS sample
        altitudes (masks are piecewise linear on the altitude levels, like the VMR between them)."""
        # masks and sample altitudes do not change between the iterations of a retrieval
        params = list(bayes_set.params())
        key = (tuple(np.asarray(par.maskgrid.mask, dtype=float).tobytes() for par in params), tuple(par.nameset for par in params), alt.tobytes())
        cached = getattr(self, "_weights_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1], cached[2]
        names = [g.name for g in self.gases]
        top = self.z[-1] + (self.z[-1] - self.z[-2])
        zz = np.append(self.z, top)
        par_gas, par_w = [], []
        for par in params:
            par_gas.append(names.index(par.nameset))
            m = np.asarray(par.maskgrid.mask, dtype=float)
            par_w.append(np.interp(alt, zz, np.append(m, m[-1])))
        out = (np.array(par_gas, np.int32), np.array(par_w))
        for a in out:
            a.setflags(write=False)     # shared with every caller until the masks change: engine keys resident batches on them
        self._weights_cache = (key, out[0], out[1])
        return out


def _one_call_eligible(pixels, bayes_set, fov_closed_form):
    """The one-call routes (engine.retrieval_forward / retrieval_step) take: every pixel with the closed-form field of view
    or none with one, every retrieved set a linear altitude profile (its VMR at a sample point is sum_p x_p w_p)."""
    with_fov = sum(pix.fov_half > 0 for pix in pixels)
    return ONE_CALL and bayes_set is not None and (with_fov == 0 or (with_fov == len(pixels) and fov_closed_form)) and \
        all(type(st) in (smm.LinearProfile_1D_new, smm.LinearProfile_1D) for st in bayes_set.sets.values())


def _one_call_batch(scene, pixels, bayes_set, alts, with_fov):
    """The resident LOS batch, the parameter weights and the pixels' FOV factors (scene._fov_fac) of a one-call forward
    model.  The gases WITHOUT parameters keep the VMRs the batch was last given: pushed again when one of them got a new
    profile (by CONTENT: an id() can be re-used by a new array and says nothing about an in-place edit)."""
    fixed = tuple(np.asarray(g.vmr, float).tobytes() for g in scene.gases if g.name not in bayes_set.sets)
    stale = getattr(scene, "_los_obj", None) is None or getattr(scene, "_fixed_vmr_key", None) != fixed
    scene._fixed_vmr_key = fixed
    los, alt = scene.los(alts, update=stale)
    par_gas, par_w = scene.profile_weights(bayes_set, alt)
    rots = tuple(pix.pixel_rot for pix in pixels)
    if with_fov and getattr(scene, "_fov_key", None) != rots:
        scene._fov_key, scene._fov_fac = rots, engine.fov_factors(rots)
    return los, par_gas, par_w


def shard_with_halo(n_grid, g_lo, g_hi):
    """Grid range a rank evaluates for its spectral shard [g_lo, g_hi) of an instrument-band integral: one point
    beyond its upper end, so that the trapezoid between the last own point and the next rank's first is counted
    exactly once (every interval (j, j + 1) belongs to the shard that owns j)."""
    return int(g_lo), int(min(g_hi + 1, n_grid))


def simulate(scene, pixels, bayes_set=None, fov_closed_form=True, shard=None, refresh=False, arrays=False, group=None):
    """One forward-model pass for all pixels (the body of the reference's iteration,
    spect_main_module.py:2736-2940): returns (sims, derivs) with sims[i] the FOV-integrated low-resolution
    spectrum of pixel i (Spectrum) and derivs[i][p] its derivative w.r.t. parameter p of bayes_set.

    shard = (g_lo, g_hi): this rank's spectral window of a multi-GPU run -- the reference splits the forward model
    of a retrieval the same way (spect_main_module.py:2814-2818: n_split contiguous chunks of the grid, results
    put together before the instrument step).  The rank computes radiances and Jacobians on its shard only and its
    PARTIAL instrument-band integrals; one all-reduce (sum) of [n_los x (1 + n_par) x n_bands] doubles over the
    ranks completes them (distributed.all_reduce_sum; a no-op without a process group), then every rank holds the
    same low-resolution spectra and runs the same n_par x n_par algebra.

    arrays=True (every pixel with a field of view and the closed form -- or none with a field of view): returns
    (low [n_pix, n_bands], dlow [n_pix, n_par, n_bands]) instead of the spectrum objects (the retrieval loop: the
    objects of an iteration were ~0.1 ms of its 0.65).

    group = (alt_step_sims, alt_first_los): the reference's group_observations route (spect_main_module.py:2668-2670,
    2908-2930, 3056-3058, 3263-3273) -- the forward model runs on a regular ladder of tangent altitudes
    (smm.make_group_observations) instead of three LOS per pixel, and the pixels' LOS spectra (and derivatives) are read
    off quadratic splines in tangent altitude (smm.make_radtran_spline) before the FOV integration."""
    from . import distributed as sd
    alts = [a for pix in pixels for a in pix.los_alts()]
    alts_pix = None
    if group is not None:
        alts_sim, _ = smm.make_group_observations(list(pixels), alt_step=group[0], alt_first_los=group[1])
        alts_pix, alts = alts, [float(a) for a in alts_sim]
    n_grid = len(scene.grid)
    g_lo, g_hi = (0, n_grid) if shard is None else shard_with_halo(n_grid, *shard)
    coeffs = scene.coefficient_stack(refresh=refresh, g_lo=g_lo, g_hi=g_hi)
    n_los = len(alts)
    lowres = lambda r: engine.hires_to_lowres(r, scene.grid, scene.bands_nm, scene.widths_nm, out_units=scene.out_units, g_lo=g_lo)
    with_fov = sum(pix.fov_half > 0 for pix in pixels)
    if arrays and group is None and _one_call_eligible(pixels, bayes_set, fov_closed_form):
        # The iteration in ONE library call (engine.retrieval_forward): the VMRs of the retrieved gases are set on the
        # device from the parameter vector (their profile IS sum_p mask_p x_p: LinearProfile_1D.profile), columns,
        # radiances + Jacobians, instrument bands and the pixels' closed-form FOV integral follow; one copy comes back.
        los, par_gas, par_w = _one_call_batch(scene, pixels, bayes_set, alts, with_fov)
        out, scene._fwd_buf = engine.retrieval_forward(coeffs, los, par_gas, par_w, bayes_set.param_vector(), scene.grid,
                                                       scene.bands_nm, scene.widths_nm, out_units=scene.out_units, g_lo=g_lo,
                                                       fov=scene._fov_fac if with_fov else None,
                                                       buf=getattr(scene, "_fwd_buf", None))
        if not with_fov:
            out = out[1::3]
        if shard is not None:
            import torch
            dev = "cuda" if (torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl") else "cpu"
            t = torch.from_numpy(np.ascontiguousarray(out)).to(dev)
            sd.all_reduce_sum(t)
            out = t.cpu().numpy()
        return out[:, 0, :], out[:, 1:, :]
    los, alt = scene.los(alts)
    if bayes_set is None:
        rad = engine.limb_rays(coeffs, los)
        low = lowres(rad)
        both = low[:, None, :]
    else:
        par_gas, par_w = scene.profile_weights(bayes_set, alt)
        # radiances and derivatives in one buffer: one instrument-step launch and one copy to the host per iteration
        _, _, buf = engine.limb_rays_jacobian(coeffs, los, par_gas, par_w, joint=True, resident=True)
        n_par = len(par_gas)
        lo_all = lowres(buf)
        both = np.concatenate([lo_all[:n_los, None, :], lo_all[n_los:].reshape(n_los, n_par, -1)], axis=1)
    if shard is not None:
        # the one exchange of a sharded iteration
        import torch
        dev = "cuda" if (torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl") else "cpu"
        t = torch.from_numpy(np.ascontiguousarray(both)).to(dev)
        sd.all_reduce_sum(t)
        both = t.cpu().numpy()
    if alts_pix is not None:
        # group_observations: every quantity (radiance, each derivative) from its spline over the simulated ladder
        at = np.empty((len(alts_pix),) + both.shape[1:])
        for q in range(both.shape[1]):
            f = smm.make_radtran_spline(alts, np.ascontiguousarray(both[:, q, :]))
            for i, a in enumerate(alts_pix):
                at[i, q] = f(a)
        both = at
    low = both[:, 0, :]
    dlow = None if bayes_set is None else both[:, 1:, :]
    sims, derivs = [], []
    grid_lo = _Grid(scene.bands_nm)              # shared by the iteration's spectra (never modified)

    def spectrum_of(v):
        sp = Spectrum.__new__(Spectrum)
        sp.spectrum, sp.spectral_grid = v, grid_lo
        return sp

    # pixels with a field of view: the closed form for all of them in one pass -- the same elementwise operations on
    # [n_pix, 1 + n_par, n_bands] stacks, a rotation per pixel (it was 45 us of a 640 us configs[4] iteration)
    fov_all = None
    if fov_closed_form and pixels and all(pix.fov_half > 0 for pix in pixels):
        fov_all = smm.fov_closed_form(both[0::3], both[1::3], both[2::3], [pix.pixel_rot for pix in pixels])
    if arrays:
        if fov_all is None:
            if any(pix.fov_half > 0 for pix in pixels):
                raise ValueError("simulate(arrays=True) needs the closed-form field of view for every pixel, or for none")
            fov_all = both[1::3]
        return fov_all[:, 0, :], fov_all[:, 1:, :]
    for i, pix in enumerate(pixels):
        if pix.fov_half > 0 and fov_closed_form:
            # the closed form is linear in the three spectra: the pixel's radiances and all its derivatives at once
            # (FOV_integr_1D per spectrum, 8 per pixel, was a third of a configs[4] iteration's host time)
            fov = fov_all[i] if fov_all is not None else \
                smm.fov_closed_form(both[3 * i], both[3 * i + 1], both[3 * i + 2], pix.pixel_rot)   # [1 + n_par, n_bands]
            sims.append(spectrum_of(np.array(fov[0])))
            if dlow is not None:
                derivs.append([spectrum_of(np.array(fov[1 + p])) for p in range(dlow.shape[1])])
            continue
        three = [Spectrum(low[3 * i + q], scene.bands_nm) for q in range(3)]
        if pix.fov_half > 0:
            sims.append(smm.FOV_integr_1D(three, pix.pixel_rot, closed_form=fov_closed_form))
        else:
            sims.append(three[1])
        if dlow is not None:
            row = []
            for p in range(dlow.shape[1]):
                d3 = [Spectrum(dlow[3 * i + q, p], scene.bands_nm) for q in range(3)]
                row.append(smm.FOV_integr_1D(d3, pix.pixel_rot, closed_form=fov_closed_form) if pix.fov_half > 0 else d3[1])
            derivs.append(row)
    return sims, derivs


def inversion_fast_limb(scene, bayes_set, pixels, chi_threshold=0.01, max_it=10, lambda_LM=0.1, L1_reg=False,
                        solo_simulation=False, check_log=None, fov_closed_form=True, shard=None, refresh=False,
                        group_observations=False, alt_step_sims=50., alt_first_los=None):
    """The retrieval loop of spect_main_module.inversion_fast_limb (:2725-2987): Levenberg-Marquardt
    optimal estimation of the VMR-profile parameters in bayes_set from the pixels' observations.
    Returns (chi, obs, sims, bayes_set) like the reference, plus .history on bayes_set (chi per iteration)
    and .stop ('converged' | 'raised' | 'max_it').  shard / refresh: see simulate (every rank of a multi-GPU run
    calls this with its own spectral shard; all ranks hold the same chi square history and parameters)."""
    pixels = sorted(pixels, key=lambda x: x.limb_tg_alt)                       # :2607
    group = (alt_step_sims, alt_first_los) if group_observations else None     # :2668-2670
    for name in bayes_set.sets.keys():                                         # :2624-2625
        scene.gas(name).add_clim(bayes_set.sets[name].profile())
    obs = [pix.observation for pix in pixels]
    masks = None if all(pix.mask is None for pix in pixels) else [pix.mask for pix in pixels]
    noise = [pix.noise for pix in pixels]
    bayes_set.history, bayes_set.stop = [], 'max_it'
    chi_old, chi, sims = None, None, []
    # The loop on arrays: when every pixel has the closed-form field of view (or none has one) an
    # iteration's spectra stay [n_pix, (n_par,) n_bands] arrays -- the same numbers in the same operations as the
    # object path (genvec / build_jacobian concatenate and mask exactly these rows) -- and become spectrum objects
    # once, when the loop ends; otherwise the objects of the reference's loop, iteration by iteration.
    with_fov = sum(pix.fov_half > 0 for pix in pixels)
    fast = not solo_simulation and ((with_fov == len(pixels) and fov_closed_form) or with_fov == 0)
    if fast:
        obs_vec, _, noi_vec = smm.genvec(obs, obs, noise, masks=masks)
        masktot = None if masks is None else np.concatenate([np.asarray(m, dtype=bool) for m in masks])
        Sa_inv = np.linalg.inv(np.asarray(bayes_set.VCM_apriori(), dtype=float))
        grid_lo = _Grid(scene.bands_nm)

        def wrap(v):
            sp = Spectrum.__new__(Spectrum)
            sp.spectrum, sp.spectral_grid = np.array(v), grid_lo
            return sp

        low = dlow = None

        def finish(low, dlow):
            if low is None:                                            # (max_it = 0: nothing was simulated)
                return []
            scene.los([a for pix in pixels for a in pix.los_alts()])   # the batch's VMRs = the final profiles (host copy too)
            jac = np.transpose(dlow, (1, 0, 2)).reshape(dlow.shape[1], -1)     # build_jacobian's rows, of the last iteration
            bayes_set.jacobian = (jac if masktot is None else jac[:, masktot]).T
            for num in range(len(pixels)):
                for ip, par in enumerate(bayes_set.params()):
                    par.store_deriv(wrap(dlow[num, ip]), num=num)              # :2929, 2940 (of the last iteration)
            return [wrap(v) for v in low]
    # Round 6: the whole iteration in ONE library call where it can be (engine.retrieval_step: forward model, chi square
    # and the Levenberg-Marquardt algebra -- the n_par x n_par systems in the library's host code -- on one process, the
    # whole grid, no observation grouping); what stays here is the reference's bookkeeping: the positivity rule of the
    # update, the stopping rule, the history.
    one_step = STEP_IN_ONE_CALL and fast and shard is None and group is None and _one_call_eligible(pixels, bayes_set, fov_closed_form)
    if one_step:
        n_pb = len(pixels) * len(scene.bands_nm)
        obs_all = np.concatenate([np.asarray(o.spectrum, float) for o in obs])
        noi_all = np.concatenate([np.asarray(nz.spectrum, float) for nz in noise])
        oe = engine.OeProblem(obs_all, noi_all, masktot, Sa_inv, bayes_set.apriori_vector(), lambda_LM)
        assert obs_all.size == n_pb
        alts_all = [a for pix in pixels for a in pix.los_alts()]
    if one_step and LOOP_IN_ONE_CALL and not refresh and max_it > 0:
        # The loop in ONE library call (sr_retrieval_loop_dev): iterations, stopping rule and the updates with their
        # positivity rule run in the library; the objects' bookkeeping is replayed here from the vectors it returns.
        coeffs = scene.coefficient_stack(refresh=False)
        los_b, par_gas, par_w = _one_call_batch(scene, pixels, bayes_set, alts_all, with_fov)
        params = list(bayes_set.params())
        for par in params:
            par.set_used()
        both, hist, xh, why, S_x, AVK, scene._fwd_buf = engine.retrieval_loop(
            coeffs, los_b, par_gas, par_w, bayes_set.param_vector(), scene.grid, scene.bands_nm, scene.widths_nm, oe,
            [bool(par.constrain_positive) for par in params], bayes_set.n_used_par(), chi_threshold=chi_threshold, max_it=max_it,
            out_units=scene.out_units, fov=scene._fov_fac if with_fov else None, buf=getattr(scene, "_fwd_buf", None))
        for k in range(1, len(xh)):                                            # update_params / update_par, :616-624
            bayes_set.old_params.append(bayes_set.values())
            for par, v in zip(params, xh[k]):
                par.old_values.append(par.value)
                par.value = v
        if AVK is not None:
            bayes_set.store_avk(AVK)
            bayes_set.store_VCM(S_x)
        bayes_set.history = [float(c) for c in hist]
        if check_log is not None:
            for num_it, c in enumerate(bayes_set.history):
                check_log.write('Iteration {:2d}: chi is {:8.3f}\n'.format(num_it, c))
        for name in bayes_set.sets.keys():                                     # :2984-2985
            scene.gas(name).add_clim(bayes_set.sets[name].profile())
        if why:
            bayes_set.stop = why
        return bayes_set.history[-1], obs, finish(both[:, 0, :], both[:, 1:, :]), bayes_set
    step_out = None
    for num_it in range(max_it):
        if one_step:
            coeffs = scene.coefficient_stack(refresh=refresh)
            los_b, par_gas, par_w = _one_call_batch(scene, pixels, bayes_set, alts_all, with_fov)
            both, chi_sum, n_used, dx, S_x, AVK, scene._fwd_buf = engine.retrieval_step(
                coeffs, los_b, par_gas, par_w, bayes_set.param_vector(), scene.grid, scene.bands_nm, scene.widths_nm, oe,
                out_units=scene.out_units, fov=scene._fov_fac if with_fov else None, buf=getattr(scene, "_fwd_buf", None))
            low, dlow = both[:, 0, :], both[:, 1:, :]
            for par in bayes_set.params():
                par.set_used()
            chi = chi_sum / (n_used - bayes_set.n_used_par())                                              # chicalc, :2949
            step_out = (dx, S_x, AVK)
        elif fast:
            low, dlow = simulate(scene, pixels, bayes_set, fov_closed_form=fov_closed_form, shard=shard, refresh=refresh, arrays=True, group=group)
            for par in bayes_set.params():
                par.set_used()
            sim_vec = low.reshape(-1) if masktot is None else low.reshape(-1)[masktot]
            chi = np.sum(((obs_vec - sim_vec) / noi_vec) ** 2) / (len(obs_vec) - bayes_set.n_used_par())   # chicalc, :2949
        else:
            sims, derivs = simulate(scene, pixels, bayes_set, fov_closed_form=fov_closed_form, shard=shard, refresh=refresh, group=group)
            if solo_simulation:
                return None
            for num, row in enumerate(derivs):
                for par, der in zip(bayes_set.params(), row):
                    par.store_deriv(der, num=num)                              # :2929, 2940
                    par.set_used()
            chi = smm.chicalc(obs, sims, noise, masks, bayes_set.n_used_par())  # :2949
        bayes_set.history.append(chi)
        if check_log is not None:
            check_log.write('Iteration {:2d}: chi is {:8.3f}\n'.format(num_it, chi))
        why = smm.retrieval_converged(chi, chi_old, chi_threshold)             # :2963-2973
        if why:
            bayes_set.stop = why
            return chi, obs, (finish(low, dlow) if fast else sims), bayes_set
        chi_old = chi
        if one_step:
            dx, S_x, AVK = step_out                                            # inversion_algebra's results, :2977
            bayes_set.update_params(dx)
            bayes_set.store_avk(AVK)
            bayes_set.store_VCM(S_x)
        elif fast:
            n_par = dlow.shape[1]
            jac = np.transpose(dlow, (1, 0, 2)).reshape(n_par, -1)             # build_jacobian's rows
            jac = (jac if masktot is None else jac[:, masktot]).T
            bayes_set.jacobian = jac
            smm.inversion_algebra_arrays(jac, obs_vec, sim_vec, noi_vec, bayes_set, lambda_LM=lambda_LM, L1_reg=L1_reg,
                                         Sa_inv=Sa_inv)                        # :2977
        else:
            smm.inversion_algebra(obs, sims, noise, bayes_set, lambda_LM=lambda_LM, L1_reg=L1_reg, masks=masks)  # :2977
        for name in bayes_set.sets.keys():                                     # :2984-2985
            scene.gas(name).add_clim(bayes_set.sets[name].profile())
    return chi, obs, (finish(low, dlow) if fast else sims), bayes_set


def lut_coefficients(scene, temp_step=5.0, pres_step_log=1.0, refresh=False, **_unused):
    """(abs, emi) of every gas at the scene's layer stack through look-up tables, the route of the reference's
    `inversion(..., useLUTs=True)` (spect_main_module.py:2447-2467 -> check_and_build_allluts; per LOS step
    LutSet.calculate, :997-1066): per gas and level the three G spectra on a rectangular (P, T) lattice that
    covers the atmosphere (temp_step [K], pres_step_log [ln hPa]: the reference's LUTopt keys) -- one
    sr_gcoeff_levels_dev call per gas (all levels from one walk of its lines), the table stays in HBM -- then per layer the bilinear interpolation and the
    population-weighted combine on the device (LutSet.combine_steps).  Tables are built once per scene."""
    import torch
    from . import spect_classes as spcl
    press, temps = scene.press, scene.temps
    Ps = np.exp(np.arange(np.floor(np.log(press.min())), np.log(press.max()) + pres_step_log, pres_step_log))
    Ts = np.arange(temp_step * np.floor(temps.min() / temp_step) - temp_step, temps.max() + 2 * temp_step, temp_step)
    PT = [[float(P), float(T)] for P in Ps for T in Ts]
    out = []
    for g in scene.gases:
        ls = g.lineset
        n_lev = ls.level_energies.size
        if getattr(g, "luts", None) is None or refresh:
            g.luts = []
            # all levels of the lattice from one walk of the line list (sr_gcoeff_levels_dev: the multi-channel pass)
            g_all = ls.gcoeff_levels([pt[1] for pt in PT], [pt[0] for pt in PT])
            for lev in range(max(n_lev, 1)):
                lut = smm.LutSet(ls.mol, ls.iso, ls.mm, level=None, level_index=lev if n_lev else None)
                lut._append(g_all[lev], PT)
                g.luts.append(lut)
        q = np.atleast_1d(spcl.CalcPartitionSum(ls.mol, ls.iso, temps))
        ab = torch.zeros((len(temps), ls.n_grid), dtype=torch.float64, device="cuda")
        em = torch.zeros_like(ab)
        for lev, lut in enumerate(g.luts):
            if n_lev:
                tv = temps if g.tvib is None else np.asarray(g.tvib)[lev]
                pops = np.exp(-spcl.c2 * ls.level_energies[lev] / tv) / q           # smm:2073
            else:
                pops = 1.0 / q                                                       # smm:2054
            lut.combine_steps(press, temps, pops, ab, em)
        out.append((ab, em))
    return out


def inversion(scene, bayes_set, pixels, chi_threshold=0.01, max_it=10, lambda_LM=0.1, L1_reg=False, useLUTs=True,
              LUTopt=None, debugfile=None, fov_closed_form=True):
    """The reference's first retrieval driver, spect_main_module.inversion (:2422-2595; radtran_test_CO.py:189 calls
    it with useLUTs=True): pixel by pixel the three lines of sight of the FOV (low / centre / up), radiances and
    derivatives of every parameter involved, hires_to_lowres, FOV_integr_1D, then chi square over all pixels with
    n_tot parameters (:2562 -- the fast loop uses the parameters in use), the same stopping rule and
    inversion_algebra.  Coefficients through look-up tables (useLUTs, lut_coefficients) or directly.  Like the
    reference it returns None: the result is the state of bayes_set (.history, .stop are added)."""
    LUTopt = dict(LUTopt or {})
    for name in bayes_set.sets.keys():                                                    # :2445-2446
        scene.gas(name).add_clim(bayes_set.sets[name].profile())
    coeffs = lut_coefficients(scene, **LUTopt) if useLUTs else scene.coefficients()
    obs = [pix.observation for pix in pixels]
    masks = None if all(pix.mask is None for pix in pixels) else [pix.mask for pix in pixels]
    noise = [pix.noise for pix in pixels]
    bayes_set.history, bayes_set.stop = [], 'max_it'
    sims = [None] * len(pixels)
    chi_old = None
    for num_it in range(max_it):
        for num, pix in enumerate(pixels):                                                 # :2487
            los, alt = scene.los(pix.los_alts())                                           # low_LOS, LOS, up_LOS
            par_gas, par_w = scene.profile_weights(bayes_set, alt)
            rad, jac = engine.limb_rays_jacobian(coeffs, los, par_gas, par_w)
            low = engine.hires_to_lowres(rad, scene.grid, scene.bands_nm, scene.widths_nm, out_units=scene.out_units)
            dlow = engine.hires_to_lowres(jac.reshape(3 * len(par_gas), -1), scene.grid, scene.bands_nm, scene.widths_nm,
                                          out_units=scene.out_units).reshape(3, len(par_gas), -1)
            three = [Spectrum(low[q], scene.bands_nm) for q in range(3)]
            sims[num] = smm.FOV_integr_1D(three, pix.pixel_rot, closed_form=fov_closed_form) if pix.fov_half > 0 else three[1]
            for p, par in enumerate(bayes_set.params()):
                if par.not_involved:                                                        # :2507-2509: zero derivative
                    par.store_deriv(Spectrum(np.zeros_like(low[0]), scene.bands_nm), num=num)
                    continue
                d3 = [Spectrum(dlow[q, p], scene.bands_nm) for q in range(3)]
                par.store_deriv(smm.FOV_integr_1D(d3, pix.pixel_rot, closed_form=fov_closed_form) if pix.fov_half > 0 else d3[1],
                                num=num)
                par.set_used()
        chi = smm.chicalc(obs, sims, noise, masks, bayes_set.n_tot)                        # :2562
        bayes_set.history.append(chi)
        why = smm.retrieval_converged(chi, chi_old, chi_threshold)                        # :2564-2570
        if why:
            bayes_set.stop = why
            return None
        chi_old = chi
        smm.inversion_algebra(obs, sims, noise, bayes_set, lambda_LM=lambda_LM, L1_reg=L1_reg, masks=masks)   # :2573
        if debugfile is not None:
            import pickle
            pickle.dump([num_it, [o.spectrum for o in obs], [s_.spectrum for s_ in sims], bayes_set.values()], debugfile)
        for name in bayes_set.sets.keys():                                                # :2581-2582
            scene.gas(name).add_clim(bayes_set.sets[name].profile())
    return None


def radtrans(scene, pixels, fov_closed_form=True, shard=None, group_observations=False, alt_step_sims=50., alt_first_los=None):
    """Simulation only (spect_main_module.radtrans, :2990-3287): the FOV-integrated low-resolution spectra.
    group_observations / alt_step_sims / alt_first_los as in the reference's signature (:2990): simulate a ladder of
    tangent altitudes and spline to the pixels' lines of sight (simulate(group=...))."""
    group = (alt_step_sims, alt_first_los) if group_observations else None
    return simulate(scene, pixels, None, fov_closed_form=fov_closed_form, shard=shard, group=group)[0]
