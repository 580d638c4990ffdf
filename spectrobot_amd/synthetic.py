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


def number_density(P_hpa, T):
    """n = P/(kb*T) with the reference's kb for hPa / cm^-3 (spect_classes.py:34)."""
    return P_hpa / (1.38065e-19 * T)


# The LOS builders are product geometry, not synthetic input: they live in spectrobot_amd.geometry (round 4) and are
# re-exported here for the callers of rounds 1-3.
from .geometry import limb_path, limb_los, limb_los_3d, slant_los, _LOS_GEOMETRY  # noqa: E402,F401
