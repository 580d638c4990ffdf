"""ctypes binding of oracle/liboracle.so -- the CPU restatement of the hot path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by spectrobot_amd/ (see oracle/sr_oracle.h).
Parity: PINNED against the reference through tests/golden/ (radiance
recursion excepted: its reference source is absent, parity unpinned).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
IMXSIG = 13010

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build(force=False):
    """Compile liboracle.so with gcc (and oracle/_ref when the reference is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("sr_oracle.c", "sr_oracle.h")]
    if force or not os.path.exists(so) or any(
            os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference") and not os.path.exists(
            os.path.join(_HERE, "_ref", "liblineshape.so")):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return so


class _Lines(C.Structure):
    _fields_ = [("n_lines", C.c_long)] + [(n, _dp) for n in (
        "freq", "a_coeff", "e_lower", "g_up", "g_lo", "air_broad", "t_dep_broad")] + [
        ("lev_up", _ip), ("lev_lo", _ip)]


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        for n in ("sro_h_cgs", "sro_c_cgs", "sro_k_cgs", "sro_c2"):
            getattr(L, n).restype = C.c_double
        L.sro_humliv_bb.argtypes = [_dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, _dp]
        L.sro_humli_bb.restype = C.c_double
        L.sro_humli_bb.argtypes = [C.c_double, C.c_double]
        L.sro_sum_all_lines.argtypes = [_dp, C.c_long, _dp, _ip, _ip, C.c_int, C.c_int]
        L.sro_sum_all_lines.restype = None
        for n, na in (("sro_convert_to_atm", 1), ("sro_lorenz_width", 4), ("sro_doppler_width", 3),
                      ("sro_boltz_ratio_nodeg", 2), ("sro_calc_bb_single", 2),
                      ("sro_linestrength_hitran", 6)):
            f = getattr(L, n)
            f.restype = C.c_double
            f.argtypes = [C.c_double] * na
        L.sro_make_shape.argtypes = [_dp, C.c_int, C.c_double, C.c_double, C.c_double, _dp]
        L.sro_closest_grid.restype = C.c_long
        L.sro_closest_grid.argtypes = [_dp, C.c_long, C.c_double]
        L.sro_calc_gcoeffs.restype = None
        L.sro_calc_gcoeffs.argtypes = [C.c_double] * 8 + [_dp]
        L.sro_calc_partition_sum.restype = C.c_double
        L.sro_calc_partition_sum.argtypes = [_dp, _dp, C.c_int, C.c_double]
        L.sro_curgod_1.restype = C.c_double
        L.sro_curgod_1.argtypes = [_dp, _dp, C.c_int]
        L.sro_curgod_2.restype = C.c_double
        L.sro_curgod_2.argtypes = [_dp, _dp, _dp, C.c_int]
        for n in ("sro_curgod_3", "sro_curgod_4"):
            f = getattr(L, n)
            f.restype = C.c_double
            f.argtypes = [_dp, _dp, _dp, _dp, C.c_int]
        L.sro_abscoeff_layers.restype = C.c_int
        L.sro_abscoeff_layers.argtypes = [C.POINTER(_Lines), C.c_double, C.c_int, _dp, C.c_int,
                                          _dp, _dp, _dp, _dp, _dp, C.c_long, C.c_int, C.c_int,
                                          _dp, _dp]
        L.sro_gcoeff_layers.restype = C.c_int
        L.sro_gcoeff_layers.argtypes = [C.POINTER(_Lines), C.c_double, C.c_int, _dp, C.c_int,
                                        _dp, _dp, _dp, _dp, _dp, C.c_long, C.c_int, _dp, _dp, _dp]
        L.sro_hires_to_lowres.restype = None
        L.sro_hires_to_lowres.argtypes = [_dp, _dp, C.c_long, _dp, _dp, C.c_int, C.c_double, C.c_int, _dp]
        L.sro_radiance_ray.restype = None
        L.sro_radiance_ray.argtypes = [_dp, _dp, C.c_long, C.c_int, _ip, _dp, _dp]
        _LIB = L
    return _LIB


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


def constants():
    L = lib()
    return dict(h_cgs=L.sro_h_cgs(), c_cgs=L.sro_c_cgs(), k_cgs=L.sro_k_cgs(), c2=L.sro_c2())


def humliv_bb(x, i1, i2, x0, lw, dw):
    """lineshape.f:226 -- returns y (same length as x); entries outside i1..i2 are 0."""
    x, xp = _d(x)
    y = np.zeros_like(x)
    rc = lib().sro_humliv_bb(xp, i1, i2, x0, lw, dw, y.ctypes.data_as(_dp))
    if rc:
        raise ValueError("humliv_bb: Fortran would stop (code %d)" % rc)
    return y


def humli_bb(rx, ry):
    return lib().sro_humli_bb(rx, ry)


def sum_all_lines(spe_ini, rows, init, fin):
    spe = np.array(spe_ini, dtype=np.float64)
    rows, rp = _d(rows)
    init, ip = _i(init)
    fin, fp = _i(fin)
    lib().sro_sum_all_lines(spe.ctypes.data_as(_dp), spe.size, rp, ip, fp, rows.shape[0], rows.shape[1])
    return spe


def convert_to_atm(p):
    return lib().sro_convert_to_atm(p)


def lorenz_width(T, P_atm, n_air, gamma_air):
    return lib().sro_lorenz_width(T, P_atm, n_air, gamma_air)


def doppler_width(T, MM, wn0):
    return lib().sro_doppler_width(T, MM, wn0)


def make_shape(xwin, wn0, lw, dw):
    xwin, xp = _d(xwin)
    s = np.zeros_like(xwin)
    rc = lib().sro_make_shape(xp, xwin.size, wn0, lw, dw, s.ctypes.data_as(_dp))
    if rc:
        raise ValueError("make_shape rc=%d" % rc)
    return s


def closest_grid(grid, wn0):
    grid, gp = _d(grid)
    return lib().sro_closest_grid(gp, grid.size, wn0)


def calc_gcoeffs(freq, A, E_low, g_up, g_lo, E_vib_up, E_vib_lo, T):
    G = np.zeros(3)
    lib().sro_calc_gcoeffs(freq, A, E_low, g_up, g_lo, E_vib_up, E_vib_lo, T, G.ctypes.data_as(_dp))
    return G  # sp_emission, ind_emission, absorption


def linestrength_hitran(A, wn, T, Q, g_up, E_low):
    return lib().sro_linestrength_hitran(A, wn, T, Q, g_up, E_low)


def boltz_ratio_nodeg(wn, T):
    return lib().sro_boltz_ratio_nodeg(wn, T)


def calc_bb_single(nu, T):
    return lib().sro_calc_bb_single(nu, T)


def calc_partition_sum(t_grid, q_grid, temp):
    t_grid, tp = _d(t_grid)
    q_grid, qp = _d(q_grid)
    return lib().sro_calc_partition_sum(tp, qp, t_grid.size, temp)


def partition_sums(mol, iso, temps):
    """Q(T) for the CHECKER's side of a comparison from the oracle alone: its Lagrange restatement
    (spect_classes.py:1692-1710) over the table the reference's Fortran returned (tests/golden/tips2003.npz, written by
    tests/golden/make_golden.py) -- nothing of the product enters (VERDICT rounds 3 and 5)."""
    g = np.load(os.path.join(os.path.dirname(_HERE), "tests", "golden", "tips2003.npz"), allow_pickle=False)
    keys = [tuple(int(v) for v in k) for k in g["keys"]]
    tab = g["q_tab"][keys.index((int(mol), int(iso)))]
    return np.array([calc_partition_sum(g["t_grid"], tab, float(t)) for t in np.atleast_1d(np.asarray(temps, float))])


def curgod(which, nd, x, vmr=None, f=None):
    L = lib()
    nd, ndp = _d(nd)
    x, xp = _d(x)
    n = nd.size
    if which == 1:
        return L.sro_curgod_1(ndp, xp, n)
    vmr, vp = _d(vmr)
    if which == 2:
        return L.sro_curgod_2(ndp, vp, xp, n)
    f, fp = _d(f)
    return getattr(L, "sro_curgod_%d" % which)(ndp, vp, fp, xp, n)


def abscoeff_layers(lines, mm, e_lev, temps, press, q_part, tvib, grid, mode=0, n_threads=1):
    """lines: dict of arrays freq,a_coeff,e_lower,g_up,g_lo,air_broad,t_dep_broad,lev_up,lev_lo.
    e_lev: level energies (empty -> the 'all' set). tvib: [n_levels, n_layers] or None (LTE).
    Returns abs[n_layers, n_grid], emi[n_layers, n_grid]."""
    keep = {}
    st = _Lines()
    st.n_lines = len(lines["freq"])
    for n in ("freq", "a_coeff", "e_lower", "g_up", "g_lo", "air_broad", "t_dep_broad"):
        keep[n], p = _d(lines[n])
        setattr(st, n, p)
    for n in ("lev_up", "lev_lo"):
        keep[n], p = _i(lines.get(n, np.zeros(st.n_lines)))
        setattr(st, n, p)
    e_lev = np.ascontiguousarray(e_lev if e_lev is not None else [], dtype=np.float64)
    temps, tp = _d(temps)
    press, pp = _d(press)
    q_part, qp = _d(q_part)
    grid, gp = _d(grid)
    nlay = temps.size
    if tvib is not None:
        tvib, tvp = _d(tvib)
        assert tvib.shape == (e_lev.size, nlay)
    else:
        tvp = None
    ab = np.zeros((nlay, grid.size))
    em = np.zeros((nlay, grid.size))
    rc = lib().sro_abscoeff_layers(C.byref(st), mm, e_lev.size, e_lev.ctypes.data_as(_dp), nlay, tp, pp, qp,
                                   tvp, gp, grid.size, mode, n_threads, ab.ctypes.data_as(_dp),
                                   em.ctypes.data_as(_dp))
    if rc:
        raise ValueError("sro_abscoeff_layers rc=%d" % rc)
    return ab, em


def gcoeff_layers(lines, mm, e_lev, temps, press, q_part, tvib, grid, n_threads=1):
    """abscoeff_layers(mode=0) that also returns the per-level G spectra it materialises:
    (abs, emi, G[n_layers, max(n_levels, 1), 3, n_grid]); ctype 0 sp_emission, 1 ind_emission, 2 absorption."""
    keep = {}
    st = _Lines()
    st.n_lines = len(lines["freq"])
    for n in ("freq", "a_coeff", "e_lower", "g_up", "g_lo", "air_broad", "t_dep_broad"):
        keep[n], p = _d(lines[n])
        setattr(st, n, p)
    for n in ("lev_up", "lev_lo"):
        keep[n], p = _i(lines.get(n, np.zeros(st.n_lines)))
        setattr(st, n, p)
    e_lev = np.ascontiguousarray(e_lev if e_lev is not None else [], dtype=np.float64)
    temps, tp = _d(temps)
    press, pp = _d(press)
    q_part, qp = _d(q_part)
    grid, gp = _d(grid)
    nlay = temps.size
    tvp = None
    if tvib is not None:
        tvib, tvp = _d(tvib)
    ab = np.zeros((nlay, grid.size))
    em = np.zeros((nlay, grid.size))
    G = np.zeros((nlay, max(e_lev.size, 1), 3, grid.size))
    rc = lib().sro_gcoeff_layers(C.byref(st), mm, e_lev.size, e_lev.ctypes.data_as(_dp), nlay, tp, pp, qp, tvp, gp,
                                 grid.size, n_threads, ab.ctypes.data_as(_dp), em.ctypes.data_as(_dp),
                                 G.ctypes.data_as(_dp))
    if rc:
        raise ValueError("sro_gcoeff_layers rc=%d" % rc)
    return ab, em, G


def radiance_ray(abs_c, emi_c, seg_layer, col, rad0=None):
    abs_c, ap = _d(abs_c)
    emi_c, ep = _d(emi_c)
    seg_layer, sp = _i(seg_layer)
    col, cp = _d(col)
    n = abs_c.shape[1]
    rad = np.zeros(n) if rad0 is None else np.array(rad0, dtype=np.float64)
    lib().sro_radiance_ray(ap, ep, n, seg_layer.size, sp, cp, rad.ctypes.data_as(_dp))
    return rad


def hires_to_lowres(grid_cm, spec, centers_nm, widths_nm, out_units="Wm2", n_sigma=5.0):
    grid_cm, gp = _d(grid_cm)
    spec, sp = _d(spec)
    centers_nm, cp = _d(centers_nm)
    widths_nm, wp = _d(widths_nm)
    out = np.zeros(centers_nm.size)
    lib().sro_hires_to_lowres(gp, sp, grid_cm.size, cp, wp, centers_nm.size, n_sigma,
                              {"Wm2": 0, "ergscm2": 1, "nWcm2": 2}[out_units], out.ctypes.data_as(_dp))
    return out
