"""Synthetic HITRAN-like line lists and Titan-like atmospheres.

No HITRAN database, vib-temp file or atmosphere of the reference's drivers
(radtran_3D_ch4.py:33-43, radtran_test_CO.py:25-60) is available offline, so the
benchmark and the tests use seeded synthetic inputs of the same shapes.  The
recipe (ranges, level count, 80-layer profile) is the one fixed in SURVEY.md 8-d;
`rng = default_rng(20260000 + config_id)`.
"""
import numpy as np

# 12 CH4-like vibrational level energies, cm^-1 (count from radtran_3D_ch4.py:281)
CH4_LEVEL_ENERGIES = np.array([0., 1311., 1533., 2587., 2612., 2830., 2846., 2917., 3019., 3062.,
                               3065., 4223.])
CH4_MM = 16.0313          # molparam.txt CH4 211
CH4_ISO_RATIO = 0.98827   # spect_main.py:152
CO_MM = 27.994915         # molparam.txt CO 26


def make_grid(w0, step, n_grid):
    """Grid exactly as prepare_spe_grid builds it (spect_main_module.py:1262-1272):
    np.arange(w0, w1 + step/2, step); w1 is chosen so that n_grid points result."""
    w1 = w0 + (n_grid - 1) * step
    g = np.arange(w0, w1 + step / 2, step, dtype=float)
    if len(g) != n_grid:  # arange end-point rounding
        g = g[:n_grid] if len(g) > n_grid else np.arange(w0, w1 + step, step, dtype=float)[:n_grid]
    assert len(g) == n_grid
    return g


def make_lines(n_lines, grid, config_id=2, n_levels=12, co_like=False, seed=None):
    """Structure-of-arrays line list, sorted by wavenumber."""
    rng = np.random.default_rng(20260000 + config_id if seed is None else seed)
    w0, w1 = grid[0], grid[-1]
    nu0 = np.sort(rng.uniform(w0, w1, n_lines))
    A = 10.0 ** rng.uniform(-2.0, 1.5, n_lines)
    E_low = rng.uniform(0.0, 2000.0, n_lines)
    J = rng.integers(0, 21, n_lines)
    if co_like:
        g_up = (2 * (J + 1) + 1).astype(float)
        g_lo = (2 * J + 1).astype(float)
    else:
        sym = rng.choice([5.0, 2.0, 3.0], n_lines)
        g_up = (2 * J + 1) * sym
        g_lo = (2 * J + 1) * rng.choice([5.0, 2.0, 3.0], n_lines)
    gamma_air = rng.uniform(0.04, 0.08, n_lines)
    n_air = rng.uniform(0.55, 0.85, n_lines)
    if n_levels > 0:
        lev_up = rng.integers(1, n_levels, n_lines).astype(np.int32)
        lo_alt = rng.integers(1, min(4, n_levels), n_lines)
        lev_lo = np.where(rng.random(n_lines) < 0.8, 0, lo_alt).astype(np.int32)
    else:
        lev_up = np.zeros(n_lines, np.int32)
        lev_lo = np.zeros(n_lines, np.int32)
    return dict(freq=nu0, a_coeff=A, e_lower=E_low, g_up=g_up, g_lo=g_lo, air_broad=gamma_air,
                t_dep_broad=n_air, lev_up=lev_up, lev_lo=lev_lo)


def make_atmosphere(n_layers=80, n_levels=12, level_energies=None):
    """Titan-like profile: z_k = 100+10k km (80 layers) or spread over the same
    range for other n_layers; returns dict(z, temps[K], press[hPa], tvib[n_levels,n_layers])."""
    if n_layers == 80:
        z = 100.0 + 10.0 * np.arange(80)
    else:
        z = np.linspace(100.0, 890.0, n_layers)
    T = 150.0 + 25.0 * np.tanh((z - 300.0) / 150.0) + 5.0 * np.sin(z / 40.0)
    P = 10.0 * np.exp(-(z - 100.0) / 45.0)
    tvib = None
    if n_levels > 0:
        tvib = np.empty((n_levels, n_layers))
        for L in range(n_levels):
            tvib[L] = T + (0.0 if L == 0 else 40.0 * (1.0 - np.exp(-(z - 100.0) / 300.0)))
    return dict(z=z, temps=T, press=P, tvib=tvib)


def limb_path(z, z_tan, R=2575.0):
    """Path segments of a limb ray with tangent height z_tan through spherical
    shells bounded by the levels z (km), in photon order (far side -> tangent
    point -> observer).  Returns (seg_layer[int32], seg_len_km)."""
    z = np.asarray(z, float)
    dz = np.diff(z)
    bounds = np.concatenate([z, [z[-1] + (dz[-1] if len(dz) else 10.0)]])  # shell k = [z_k, z_k+1)
    rt = R + z_tan
    lay, ln = [], []
    for k in range(len(z)):
        lo, hi = R + bounds[k], R + bounds[k + 1]
        if hi <= rt:
            continue
        s_hi = np.sqrt(hi * hi - rt * rt)
        s_lo = np.sqrt(lo * lo - rt * rt) if lo > rt else 0.0
        lay.append(k)
        ln.append(s_hi - s_lo)
    lay = np.array(lay, np.int32)
    ln = np.array(ln)
    # far side: outermost -> tangent; near side: tangent -> outermost
    return np.concatenate([lay[::-1], lay]).astype(np.int32), np.concatenate([ln[::-1], ln])


def number_density(P_hpa, T):
    """n = P/(kb*T) with the reference's kb for hPa / cm^-3 (spect_classes.py:34)."""
    return P_hpa / (1.38065e-19 * T)


_LOS_GEOMETRY = {}   # (levels, tangent heights, R, n_sub) -> segment / sample-point geometry of limb_los


def limb_los(z, nd_levels, vmr_levels, z_tans, R=2575.0, n_sub=3):
    """Lines of sight of limb rays through spherical shells for the device LOS pipeline
    (engine.LimbLOS): per ray the shell crossings in photon order (far side -> tangent point ->
    observer), per crossing n_sub + 1 sample points along the path with the number density
    interpolated exponentially and every VMR linearly in altitude between the levels z (the
    profiles curgods.f assumes).  vmr_levels: [n_gas, n_levels].  Returns the LimbLOS arguments
    dict(seg_off, seg_layer, pt_off, x [cm], nd, vmr [n_gas, n_pt]) plus `alt` [n_pt] (km)."""
    z = np.asarray(z, float)
    nd_levels = np.asarray(nd_levels, float)
    vmr_levels = np.atleast_2d(np.asarray(vmr_levels, float))
    dz = np.diff(z)
    top = z[-1] + (dz[-1] if len(dz) else 10.0)
    bounds = np.concatenate([z, [top]])
    # profiles continued to the top boundary with the last scale height / last VMR
    lognd = np.log(nd_levels)
    lognd_top = lognd[-1] + (lognd[-1] - lognd[-2]) / dz[-1] * (top - z[-1]) if len(dz) else lognd[-1]
    zz = np.concatenate([z, [top]])
    ln = np.concatenate([lognd, [lognd_top]])
    vv = np.concatenate([vmr_levels, vmr_levels[:, -1:]], axis=1)
    z_tans = np.atleast_1d(np.asarray(z_tans, float))
    # the geometry depends on the levels and the tangent heights only: a retrieval loop asks for the same rays with
    # new VMR profiles every iteration (it was 11 of the 15 ms of a configs[4] iteration)
    key = (z.tobytes(), z_tans.tobytes(), float(R), int(n_sub))
    geo = _LOS_GEOMETRY.get(key)
    if geo is None:
        seg_off, seg_layer, pt_off, xs, alts = [0], [], [0], [], []
        for zt in z_tans:
            rt = R + zt
            shells = []
            for k in range(len(z)):
                lo, hi = R + bounds[k], R + bounds[k + 1]
                if hi <= rt:
                    continue
                s_lo = np.sqrt(lo * lo - rt * rt) if lo > rt else 0.0
                shells.append((k, s_lo, np.sqrt(hi * hi - rt * rt)))
            # far side: s from -s_hi(top) up to the tangent point (s = 0), near side: 0 .. +s_hi(top)
            crossings = [(k, -s_hi, -s_lo) for k, s_lo, s_hi in shells[::-1]] + [(k, s_lo, s_hi) for k, s_lo, s_hi in shells]
            for k, a, b in crossings:
                s = np.linspace(a, b, n_sub + 1)
                seg_layer.append(k)
                xs += list(s)
                alts += list(np.sqrt(s * s + rt * rt) - R)
                pt_off.append(len(xs))
            seg_off.append(len(seg_layer))
        geo = (np.array(seg_off, np.int32), np.array(seg_layer, np.int32), np.array(pt_off, np.int32),
               np.array(xs) * 1e5, np.clip(np.array(alts), z[0], top))
        if len(_LOS_GEOMETRY) >= 8:
            _LOS_GEOMETRY.pop(next(iter(_LOS_GEOMETRY)))
        _LOS_GEOMETRY[key] = geo
    seg_off, seg_layer, pt_off, x_cm, alts = geo
    nd = np.exp(np.interp(alts, zz, ln))
    vmr = np.array([np.interp(alts, zz, v) for v in vv])
    return dict(seg_off=seg_off, seg_layer=seg_layer, pt_off=pt_off, x=x_cm, nd=nd, vmr=vmr, alt=alts)


def limb_los_3d(z, nd_levels, vmr_levels, z_tans, sza_tangent_deg, azimuth_deg, R=2575.0, n_sub=3):
    """limb_los for a 3-D atmosphere: every LOS step (shell crossing) is its own "layer" -- its own row of the
    coefficient tables -- because the state along the path depends on the local illumination, not on altitude alone.
    Stands in for the absent sbm LineOfSight.calc_atm_intersections + calc_SZA_along_los
    (spect_main_module.py:2746-2757 with use_tangent_sza = False): the solar zenith angle at path coordinate s (km
    from the tangent point, positive towards the observer) of a ray whose tangent point sees the sun at
    sza_tangent and whose direction makes the azimuth angle with the sun's horizontal direction there is

        cos SZA(s) = (r_t cos SZA_t + s sin SZA_t cos az) / sqrt(r_t^2 + s^2),     r_t = R + z_tan.

    Returns limb_los's dict with seg_layer = 0 .. n_seg-1 (one coefficient row per step) plus, per step,
    `seg_alt_layer` (the altitude shell, i.e. the row of a per-altitude Jacobian) and `seg_mu` = cos SZA at the
    middle of the step.  z_tans, azimuth_deg: one per ray; sza_tangent_deg: scalar or one per ray."""
    L = limb_los(z, nd_levels, vmr_levels, z_tans, R=R, n_sub=n_sub)
    z_tans = np.atleast_1d(np.asarray(z_tans, float))
    az = np.deg2rad(np.broadcast_to(np.asarray(azimuth_deg, float), z_tans.shape))
    szt = np.deg2rad(np.broadcast_to(np.asarray(sza_tangent_deg, float), z_tans.shape))
    n_seg = len(L["seg_layer"])
    mu = np.empty(n_seg)
    for r in range(len(z_tans)):
        rt = R + z_tans[r]
        for sg in range(L["seg_off"][r], L["seg_off"][r + 1]):
            a, b = L["pt_off"][sg], L["pt_off"][sg + 1]
            s_mid = 0.5 * (L["x"][a] + L["x"][b - 1]) * 1e-5      # km; x is the path coordinate, 0 at the tangent point
            mu[sg] = (rt * np.cos(szt[r]) + s_mid * np.sin(szt[r]) * np.cos(az[r])) / np.sqrt(rt * rt + s_mid * s_mid)
    out = dict(L)
    out["seg_alt_layer"] = L["seg_layer"].copy()
    out["seg_layer"] = np.arange(n_seg, dtype=np.int32)
    out["seg_mu"] = mu
    return out


def slant_los(z, nd_levels, vmr_levels, zenith_deg, R=2575.0, n_sub=3):
    """Upward-looking-from-below / nadir-viewing paths: rays that leave the lowest level z[0] at the given zenith
    angles (0 = nadir view / vertical path) and cross every shell once, in photon order (bottom -> top, the observer
    is above the atmosphere).  The geometry of the reference's planetary (non-limb) cases -- BASELINE configs[0]
    quotes a "40-layer 1D nadir" CO case -- whose LineOfSight code is in the absent spect_base_module: a ray of
    impact parameter b = (R + z[0]) sin(zenith) has the path length sqrt(r_hi^2 - b^2) - sqrt(r_lo^2 - b^2) in the
    shell [r_lo, r_hi].  Same dict as limb_los (LimbLOS arguments + `alt`); combine with
    LimbLOS(initial_temperature=T_surface) for the surface emission behind the path."""
    z = np.asarray(z, float)
    nd_levels = np.asarray(nd_levels, float)
    vmr_levels = np.atleast_2d(np.asarray(vmr_levels, float))
    dz = np.diff(z)
    top = z[-1] + (dz[-1] if len(dz) else 10.0)
    zz = np.concatenate([z, [top]])
    lognd = np.log(nd_levels)
    ln = np.concatenate([lognd, [lognd[-1] + (lognd[-1] - lognd[-2]) / dz[-1] * (top - z[-1]) if len(dz) else lognd[-1]]])
    vv = np.concatenate([vmr_levels, vmr_levels[:, -1:]], axis=1)
    seg_off, seg_layer, pt_off, xs, alts = [0], [], [0], [], []
    for zen in np.atleast_1d(np.asarray(zenith_deg, float)):
        if not 0.0 <= zen < 90.0:
            raise ValueError("zenith angle must be in [0, 90)")
        b = (R + z[0]) * np.sin(np.deg2rad(zen))
        for k in range(len(z)):
            lo, hi = R + zz[k], R + zz[k + 1]
            s = np.linspace(np.sqrt(lo * lo - b * b), np.sqrt(hi * hi - b * b), n_sub + 1)   # path coordinate from the closest approach
            seg_layer.append(k)
            xs += list(s)
            alts += list(np.sqrt(s * s + b * b) - R)
            pt_off.append(len(xs))
        seg_off.append(len(seg_layer))
    alts = np.clip(np.array(alts), z[0], top)
    nd = np.exp(np.interp(alts, zz, ln))
    vmr = np.array([np.interp(alts, zz, v) for v in vv])
    return dict(seg_off=np.array(seg_off, np.int32), seg_layer=np.array(seg_layer, np.int32), pt_off=np.array(pt_off, np.int32),
                x=np.array(xs) * 1e5, nd=nd, vmr=vmr, alt=alts)
