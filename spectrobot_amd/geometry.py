"""Line-of-sight geometry for the device LOS pipeline (engine.LimbLOS): limb, slant / nadir and 3-D paths through
spherical shells, and the adaptive LOS stepping of the reference's drivers.

The reference's LineOfSight class (calc_atm_intersections, calc_radtran_steps, calc_SZA_along_los, radtran*) lives in
the absent spect_base_module; what is pinned are its call sites and their knobs:

    los.calc_radtran_steps(planet, lines, max_opt_depth=..., max_T_variation=5., max_Plog_variation=1.)
        spect_main_module.py:2746-2767, 3147; radtran_test_CO.py:184-186; spect_radtran_test.py:175;
        radtran_3Dvs2D_sza30-80_test.py:285-287
    use_tangent_sza / invert_LOS_direction (LOS_order)           spect_main_module.py:2748-2757
    atmosphere on (latitude box, altitude): ['box', 'lin'] temperature, ['box', 'exp'] pressure
        radtran_3Dvs2D_sza30-80_test.py:66-90 (lat_ext = [-90, -75, -60, -30, 30, 60, 75, 90])

so these builders are the build's own definition behind those names: a step is a stretch of the path inside one
altitude shell (and one latitude box); fixed stepping = one step per shell crossing (the benches), adaptive stepping
splits a crossing until the temperature, log-pressure and optical-depth variation of every step stay inside the
bounds.  Everything is vectorised over the segments of a ray (no per-segment Python loops).

Profiles along the path are the ones curgods.f assumes: number density exponential, VMR linear in altitude between
the levels; the column of a segment is curgod_fort_2 over its n_sub + 1 sample points, evaluated on the device.
"""
import numpy as np

TITAN_RADIUS_KM = 2575.0                                   # spect_classes.py:32
LAT_EXT = np.array([-90., -75., -60., -30., 30., 60., 75., 90.])   # radtran_3Dvs2D_sza30-80_test.py:67


def _profiles(z, nd_levels, vmr_levels):
    """Level profiles continued to the top boundary (one more shell of the last thickness) with the last scale height
    / last VMR: (zz, ln nd, vmr) on the len(z) + 1 boundaries."""
    z = np.asarray(z, float)
    nd_levels = np.asarray(nd_levels, float)
    vmr_levels = np.atleast_2d(np.asarray(vmr_levels, float))
    dz = np.diff(z)
    top = z[-1] + (dz[-1] if len(dz) else 10.0)
    lognd = np.log(nd_levels)
    lognd_top = lognd[-1] + (lognd[-1] - lognd[-2]) / dz[-1] * (top - z[-1]) if len(dz) else lognd[-1]
    zz = np.concatenate([z, [top]])
    return z, zz, np.concatenate([lognd, [lognd_top]]), np.concatenate([vmr_levels, vmr_levels[:, -1:]], axis=1)


def _limb_crossings(zz, zt, R):
    """Shell crossings of one limb ray in photon order (far side -> tangent point -> observer): shell index k and the
    path coordinates (km from the tangent point, negative on the far side) of its two ends."""
    rt = R + zt
    lo, hi = R + zz[:-1], R + zz[1:]
    k = np.nonzero(hi > rt)[0]
    s_hi = np.sqrt(hi[k] * hi[k] - rt * rt)
    s_lo = np.where(lo[k] > rt, np.sqrt(np.maximum(lo[k] * lo[k] - rt * rt, 0.0)), 0.0)
    kk = np.concatenate([k[::-1], k])
    a = np.concatenate([-s_hi[::-1], s_lo])
    b = np.concatenate([-s_lo[::-1], s_hi])
    return kk.astype(np.int32), a, b


def limb_path(z, z_tan, R=TITAN_RADIUS_KM):
    """Path segments of a limb ray with tangent height z_tan through spherical shells bounded by the levels z (km), in
    photon order.  Returns (seg_layer[int32], seg_len_km)."""
    z, zz, _, _ = _profiles(z, np.ones(len(z)), np.ones((1, len(z))))
    k, a, b = _limb_crossings(zz, float(z_tan), R)
    return k, b - a


def _sample(a, b, n_sub):
    """n_sub + 1 equally spaced path coordinates per segment, [n_seg, n_sub + 1] (np.linspace per row)."""
    step = (b - a) / n_sub
    s = np.arange(n_sub + 1)[None, :] * step[:, None] + a[:, None]      # np.linspace's own arithmetic
    s[:, -1] = b                                                      # ... which ends exactly on b
    return s


_LOS_GEOMETRY = {}   # (levels, tangent heights, R, n_sub) -> segment / sample-point geometry of limb_los


def limb_los(z, nd_levels, vmr_levels, z_tans, R=TITAN_RADIUS_KM, n_sub=3):
    """Lines of sight of limb rays through spherical shells (fixed stepping: one step per shell crossing): per ray the
    crossings in photon order, per crossing n_sub + 1 sample points with the number density interpolated exponentially
    and every VMR linearly in altitude between the levels z.  vmr_levels: [n_gas, n_levels].  Returns the LimbLOS
    arguments dict(seg_off, seg_layer, pt_off, x [cm], nd, vmr [n_gas, n_pt]) plus `alt` [n_pt] (km)."""
    z, zz, ln, vv = _profiles(z, nd_levels, vmr_levels)
    z_tans = np.atleast_1d(np.asarray(z_tans, float))
    # the geometry depends on the levels and the tangent heights only: a retrieval loop asks for the same rays with
    # new VMR profiles every iteration (it was 11 of the 15 ms of a configs[4] iteration)
    key = (z.tobytes(), z_tans.tobytes(), float(R), int(n_sub))
    geo = _LOS_GEOMETRY.get(key)
    if geo is None:
        seg_off, lay, xs, alts = [0], [], [], []
        for zt in z_tans:
            k, a, b = _limb_crossings(zz, zt, R)
            s = _sample(a, b, n_sub)
            lay.append(k)
            xs.append(s.ravel())
            alts.append((np.sqrt(s * s + (R + zt) ** 2) - R).ravel())
            seg_off.append(seg_off[-1] + len(k))
        n_seg = seg_off[-1]
        geo = (np.array(seg_off, np.int32), np.concatenate(lay).astype(np.int32),
               (np.arange(n_seg + 1) * (n_sub + 1)).astype(np.int32), np.concatenate(xs) * 1e5,
               np.clip(np.concatenate(alts), z[0], zz[-1]))
        if len(_LOS_GEOMETRY) >= 8:
            _LOS_GEOMETRY.pop(next(iter(_LOS_GEOMETRY)))
        _LOS_GEOMETRY[key] = geo
    seg_off, seg_layer, pt_off, x_cm, alts = geo
    nd = np.exp(np.interp(alts, zz, ln))
    vmr = np.array([np.interp(alts, zz, v) for v in vv])
    return dict(seg_off=seg_off, seg_layer=seg_layer, pt_off=pt_off, x=x_cm, nd=nd, vmr=vmr, alt=alts)


def sun_in_local_frame(tangent_lat_deg, subsolar_lat_deg, sza_tangent_deg):
    """Unit vector to the sun in the (up, north, east) frame of a tangent point at the given latitude that sees the sun
    at sza_tangent: the hour angle H follows from cos SZA = sin lat sin dec + cos lat cos dec cos H (afternoon side).
    Raises if that SZA does not occur at this latitude."""
    lat, dec, sza = (np.deg2rad(np.asarray(v, float)) for v in (tangent_lat_deg, subsolar_lat_deg, sza_tangent_deg))
    ch = (np.cos(sza) - np.sin(lat) * np.sin(dec)) / (np.cos(lat) * np.cos(dec))
    if np.any(np.abs(ch) > 1.0):
        raise ValueError("a solar zenith angle of %s deg does not occur at latitude %s (subsolar latitude %s)"
                         % (sza_tangent_deg, tangent_lat_deg, subsolar_lat_deg))
    H = np.arccos(ch)
    sun = np.stack([np.cos(dec) * np.cos(H), np.cos(dec) * np.sin(H), np.sin(dec) * np.ones_like(H)], axis=-1)   # x: lon 0, z: pole
    up = np.stack([np.cos(lat), np.zeros_like(lat), np.sin(lat)], axis=-1)
    north = np.stack([-np.sin(lat), np.zeros_like(lat), np.cos(lat)], axis=-1)
    east = np.stack([np.zeros_like(lat), np.ones_like(lat), np.zeros_like(lat)], axis=-1)
    return np.stack([(sun * up).sum(-1), (sun * north).sum(-1), (sun * east).sum(-1)], axis=-1)


def path_state_3d(L, z_tans, sun_local, heading_deg, tangent_lat_deg=0.0, R=TITAN_RADIUS_KM):
    """Per segment of the LOS batch L (limb_los / calc_radtran_steps output): cos SZA and latitude (deg) at the middle
    of the segment, for rays whose tangent points lie at tangent_lat (lon 0), head `heading_deg` east of north there and
    see the sun in the direction sun_local = (up, north, east) components (sun_in_local_frame).  The point at path
    coordinate s is p = r_t up + s d, d = cos(heading) north + sin(heading) east:

        cos SZA(s) = (r_t sun_up + s (sun_north cos h + sun_east sin h)) / |p|,   lat(s) = asin(p_z / |p|)."""
    z_tans = np.atleast_1d(np.asarray(z_tans, float))
    n_rays = len(z_tans)
    sun = np.broadcast_to(np.asarray(sun_local, float), (n_rays, 3))
    hd = np.deg2rad(np.broadcast_to(np.asarray(heading_deg, float), (n_rays,)))
    lat_t = np.deg2rad(np.broadcast_to(np.asarray(tangent_lat_deg, float), (n_rays,)))
    seg_off, pt_off = L["seg_off"], L["pt_off"]
    n_seg = len(L["seg_layer"])
    ray = np.repeat(np.arange(n_rays), np.diff(seg_off))
    s_mid = 0.5 * (L["x"][pt_off[:-1]] + L["x"][pt_off[1:] - 1]) * 1e-5      # km from the tangent point
    rt = R + z_tans[ray]
    norm = np.sqrt(rt * rt + s_mid * s_mid)
    mu = (rt * sun[ray, 0] + s_mid * (sun[ray, 1] * np.cos(hd[ray]) + sun[ray, 2] * np.sin(hd[ray]))) / norm
    pz = rt * np.sin(lat_t[ray]) + s_mid * np.cos(hd[ray]) * np.cos(lat_t[ray])
    lat = np.rad2deg(np.arcsin(np.clip(pz / norm, -1.0, 1.0)))
    assert len(mu) == n_seg
    return mu, lat


def lat_box_index(lat_deg, lat_ext=LAT_EXT):
    """Latitude box of the reference's 3-D atmospheres (['box', ...] interpolation on lat_ext[:-1])."""
    return np.clip(np.searchsorted(np.asarray(lat_ext, float), np.asarray(lat_deg, float), side="right") - 1, 0, len(lat_ext) - 2)


def limb_los_3d(z, nd_levels, vmr_levels, z_tans, sza_tangent_deg, azimuth_deg, R=TITAN_RADIUS_KM, n_sub=3,
                tangent_lat_deg=None, subsolar_lat_deg=None):
    """limb_los for a 3-D atmosphere: every LOS step is its own row of the coefficient tables, because the state along
    the path depends on the local illumination (and latitude), not on altitude alone.  Stands in for the absent sbm
    LineOfSight.calc_atm_intersections + calc_SZA_along_los (spect_main_module.py:2746-2757 with use_tangent_sza =
    False).  Two ways to place the sun:

      tangent_lat_deg None (round 3): azimuth_deg is the angle between the ray and the sun's horizontal direction at
        the tangent point:  cos SZA(s) = (r_t cos SZA_t + s sin SZA_t cos az) / sqrt(r_t^2 + s^2);
      tangent_lat_deg given: the tangent points lie at that latitude, the rays head azimuth_deg east of north and the
        sun stands at subsolar_lat_deg with the hour angle that gives SZA_t there (sun_in_local_frame).

    Returns limb_los's dict with seg_layer = 0 .. n_seg-1 (one coefficient row per step) plus, per step, `seg_alt_layer`
    (the altitude shell), `seg_mu` = cos SZA and `seg_lat` (deg) at the middle of the step."""
    L = limb_los(z, nd_levels, vmr_levels, z_tans, R=R, n_sub=n_sub)
    z_tans = np.atleast_1d(np.asarray(z_tans, float))
    if tangent_lat_deg is None:
        szt = np.deg2rad(np.broadcast_to(np.asarray(sza_tangent_deg, float), z_tans.shape))
        sun = np.stack([np.cos(szt), np.sin(szt), np.zeros_like(szt)], axis=-1)     # sun's horizontal direction = "north"
        lat_t = 0.0
    else:
        lat_t = np.broadcast_to(np.asarray(tangent_lat_deg, float), z_tans.shape)
        sun = sun_in_local_frame(lat_t, 0.0 if subsolar_lat_deg is None else subsolar_lat_deg,
                                 np.broadcast_to(np.asarray(sza_tangent_deg, float), z_tans.shape))
    mu, lat = path_state_3d(L, z_tans, sun, azimuth_deg, lat_t, R=R)
    out = dict(L)
    out["seg_alt_layer"] = L["seg_layer"].copy()
    out["seg_layer"] = np.arange(len(mu), dtype=np.int32)
    out["seg_mu"] = mu
    out["seg_lat"] = lat
    return out


def slant_los(z, nd_levels, vmr_levels, zenith_deg, R=TITAN_RADIUS_KM, n_sub=3):
    """Upward-looking-from-below / nadir-viewing paths: rays that leave the lowest level z[0] at the given zenith
    angles (0 = nadir view / vertical path) and cross every shell once, in photon order (bottom -> top, the observer
    is above the atmosphere).  The geometry of the reference's planetary (non-limb) cases -- BASELINE configs[0]
    quotes a "40-layer 1D nadir" CO case: a ray of impact parameter b = (R + z[0]) sin(zenith) has the path length
    sqrt(r_hi^2 - b^2) - sqrt(r_lo^2 - b^2) in the shell [r_lo, r_hi].  Same dict as limb_los; combine with
    LimbLOS(initial_temperature=T_surface) for the surface emission behind the path."""
    z, zz, ln, vv = _profiles(z, nd_levels, vmr_levels)
    seg_off, lay, xs, alts = [0], [], [], []
    for zen in np.atleast_1d(np.asarray(zenith_deg, float)):
        if not 0.0 <= zen < 90.0:
            raise ValueError("zenith angle must be in [0, 90)")
        b = (R + z[0]) * np.sin(np.deg2rad(zen))
        lo, hi = R + zz[:-1], R + zz[1:]
        s = _sample(np.sqrt(lo * lo - b * b), np.sqrt(hi * hi - b * b), n_sub)   # path coordinate from the closest approach
        lay.append(np.arange(len(z), dtype=np.int32))
        xs.append(s.ravel())
        alts.append((np.sqrt(s * s + b * b) - R).ravel())
        seg_off.append(seg_off[-1] + len(z))
    alts = np.clip(np.concatenate(alts), z[0], zz[-1])
    nd = np.exp(np.interp(alts, zz, ln))
    vmr = np.array([np.interp(alts, zz, v) for v in vv])
    n_seg = seg_off[-1]
    return dict(seg_off=np.array(seg_off, np.int32), seg_layer=np.concatenate(lay).astype(np.int32),
                pt_off=(np.arange(n_seg + 1) * (n_sub + 1)).astype(np.int32), x=np.concatenate(xs) * 1e5, nd=nd, vmr=vmr, alt=alts)


def calc_radtran_steps(z, temps, press, nd_levels, vmr_levels, z_tans, R=TITAN_RADIUS_KM, n_sub=3, max_T_variation=None,
                       max_Plog_variation=None, max_opt_depth=None, opt_depth_of=None, max_rounds=12):
    """Adaptive LOS stepping with the reference's knobs (radtran_opt: max_T_variation [K], max_Plog_variation [ln P],
    max_opt_depth): every shell crossing of every limb ray is split into equal ALTITUDE slices until, inside each step,
    the temperature (linear in altitude between the levels, ['lin']) varies by at most max_T_variation and ln P
    (['exp']) by at most max_Plog_variation; then steps whose optical depth exceeds max_opt_depth are halved until none
    does (opt_depth_of(steps) -> largest optical depth of every step over the spectral grid; the caller supplies it
    from the coefficient rows on the device, engine.refine_los_by_opt_depth).  None = that bound is off; all three off
    reproduces limb_los's crossings.

    A step carries the state at its own mean altitude: step_temp (linear), step_pres (exponential in altitude) -- its
    coefficient row; `seg_layer` numbers the steps 0 .. n_seg-1 as in limb_los_3d, `seg_alt_layer` is the shell.
    Returns limb_los's dict + seg_alt_layer, step_temp, step_pres, step_alt (mean altitude, km)."""
    z, zz, ln, vv = _profiles(z, nd_levels, vmr_levels)
    temps, press = np.asarray(temps, float), np.asarray(press, float)
    dz_last = zz[-1] - zz[-2]
    # state on the boundaries: T continued constant, ln P with its last scale height
    tt = np.concatenate([temps, temps[-1:]])
    lp = np.log(press)
    lpp = np.concatenate([lp, [lp[-1] + ((lp[-1] - lp[-2]) / (z[-1] - z[-2]) * dz_last if len(z) > 1 else 0.0)]])
    z_tans = np.atleast_1d(np.asarray(z_tans, float))
    rays = []
    for zt in z_tans:
        k, a, b = _limb_crossings(zz, zt, R)
        rt = R + zt
        alt_a, alt_b = np.sqrt(a * a + rt * rt) - R, np.sqrt(b * b + rt * rt) - R
        dT = np.abs(np.interp(alt_b, zz, tt) - np.interp(alt_a, zz, tt))
        dP = np.abs(np.interp(alt_b, zz, lpp) - np.interp(alt_a, zz, lpp))
        n = np.ones(len(k), int)
        if max_T_variation:
            n = np.maximum(n, np.ceil(dT / max_T_variation - 1e-9).astype(int))
        if max_Plog_variation:
            n = np.maximum(n, np.ceil(dP / max_Plog_variation - 1e-9).astype(int))
        # equal altitude slices of each crossing: slice i of n covers alt_a + (alt_b - alt_a) [i, i + 1] / n
        rep = np.repeat(np.arange(len(k)), n)
        i = np.arange(len(rep)) - np.repeat(np.cumsum(n) - n, n)
        f0, f1 = i / n[rep], (i + 1) / n[rep]
        h0 = alt_a[rep] + (alt_b - alt_a)[rep] * f0
        h1 = alt_a[rep] + (alt_b - alt_a)[rep] * f1
        sign = np.where(b[rep] <= 0.0, -1.0, 1.0)               # far side: s < 0
        s0 = sign * np.sqrt(np.maximum((R + h0) ** 2 - rt * rt, 0.0))
        s1 = sign * np.sqrt(np.maximum((R + h1) ** 2 - rt * rt, 0.0))
        # the crossing's own ends exactly (no drift from the altitude round trip)
        first, last = i == 0, i == n[rep] - 1
        s0[first], s1[last] = a[rep][first], b[rep][last]
        rays.append(dict(k=k[rep], a=s0, b=s1, rt=rt))

    def assemble():
        seg_off, lay, xs, alts = [0], [], [], []
        for r in rays:
            s = _sample(r["a"], r["b"], n_sub)
            lay.append(r["k"])
            xs.append(s.ravel())
            alts.append((np.sqrt(s * s + r["rt"] ** 2) - R).ravel())
            seg_off.append(seg_off[-1] + len(r["k"]))
        n_seg = seg_off[-1]
        alt = np.clip(np.concatenate(alts), z[0], zz[-1])
        pt_off = (np.arange(n_seg + 1) * (n_sub + 1)).astype(np.int32)
        mid = np.concatenate([np.sqrt((0.5 * (r["a"] + r["b"])) ** 2 + r["rt"] ** 2) - R for r in rays])
        mid = np.clip(mid, z[0], zz[-1])
        return dict(seg_off=np.array(seg_off, np.int32), seg_layer=np.arange(n_seg, dtype=np.int32),
                    seg_alt_layer=np.concatenate(lay).astype(np.int32), pt_off=pt_off, x=np.concatenate(xs) * 1e5,
                    nd=np.exp(np.interp(alt, zz, ln)), vmr=np.array([np.interp(alt, zz, v) for v in vv]), alt=alt,
                    step_alt=mid, step_temp=np.interp(mid, zz, tt), step_pres=np.exp(np.interp(mid, zz, lpp)))

    L = assemble()
    if max_opt_depth and opt_depth_of is not None:
        for _ in range(max_rounds):
            tau = np.asarray(opt_depth_of(L), float)
            too = tau > max_opt_depth
            if not too.any():
                break
            at = 0
            for r in rays:                                   # halve (in path length) the steps that are too thick
                m = too[at:at + len(r["k"])]
                at += len(r["k"])
                rep = np.repeat(np.arange(len(m)), np.where(m, 2, 1))
                second = np.concatenate([[False], rep[1:] == rep[:-1]])
                mid_s = 0.5 * (r["a"] + r["b"])
                a2 = np.where(second, mid_s[rep], r["a"][rep])
                b2 = np.where(m[rep] & ~second, mid_s[rep], r["b"][rep])
                r["k"], r["a"], r["b"] = r["k"][rep], a2, b2
            L = assemble()
    return L
