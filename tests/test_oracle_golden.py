"""The oracle (oracle/sr_oracle.c) against the fixtures generated from the
reference itself (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import relerr


def test_constants(oracle, golden):
    g = golden("spcl_scalars")
    c = oracle.constants()
    for k in ("h_cgs", "c_cgs", "k_cgs", "c2"):
        assert c[k] == float(g["const_" + k]), k


def test_humliv_windows(oracle, golden):
    """A1 lineshape.f:226-569, middle branch, ry from 4e-9 to 258."""
    g = golden("humliv_windows")
    worst = 0.0
    for x, y, p in zip(g["x"], g["y"], g["par"]):
        yo = oracle.humliv_bb(x, 1, 13010, p[0], p[1], p[2])
        worst = max(worst, relerr(yo, y))
    assert worst < 2e-14, worst


def test_humliv_outer_branches(oracle, golden):
    g = golden("humliv_windows")
    for p, y in zip(g["outer_par"], g["outer_y"]):
        yo = oracle.humliv_bb(g["outer_x"], 1, 13010, p[0], p[1], p[2])
        assert relerr(yo, y) < 2e-14


def test_humli_scalar(oracle, golden):
    g = golden("humliv_windows")
    yo = [oracle.humli_bb(a, b) for a, b in zip(g["humli_rx"], g["humli_ry"])]
    assert relerr(yo, g["humli_y"]) < 2e-14


def test_humliv_errors(oracle):
    x = np.linspace(0, 1, 13010)
    with pytest.raises(ValueError):
        oracle.humliv_bb(x, 5, 4, 0.5, 1e-3, 1e-3)
    with pytest.raises(ValueError):
        oracle.humliv_bb(x, 1, 13010, 0.5, 1e-3, 0.0)


def test_scalars(oracle, golden):
    """A3/A4: widths, G coefficients, Planck, Boltzmann, LTE strength, closest_grid."""
    g = golden("spcl_scalars")
    n = len(g["T"])
    lw = [oracle.lorenz_width(g["T"][i], oracle.convert_to_atm(g["P"][i]), g["n_air"][i], g["gam"][i])
          for i in range(n)]
    dw = [oracle.doppler_width(g["T"][i], g["MM"][i], g["nu"][i]) for i in range(n)]
    assert relerr(lw, g["lw"]) < 1e-15
    assert relerr(dw, g["dw"]) < 1e-15
    G = np.array([oracle.calc_gcoeffs(g["nu"][i], g["A"][i], g["El"][i], g["gu"][i], g["gl"][i], g["Evu"][i],
                                      g["Evl"][i], g["T"][i]) for i in range(n)])
    assert relerr(G, g["G"]) < 1e-14
    S = [oracle.linestrength_hitran(g["A"][i], g["nu"][i], g["T"][i], 1.0, g["gu"][i], g["El"][i])
         for i in range(n)]
    assert relerr(S, g["S_hitran_Q1"]) < 1e-14
    assert relerr([oracle.calc_bb_single(g["nu"][i], g["T"][i]) for i in range(n)], g["bb"]) < 1e-14
    assert relerr([oracle.boltz_ratio_nodeg(g["El"][i], g["T"][i]) for i in range(n)], g["boltz"]) < 1e-15
    grid = float(g["grid_w0"]) + float(g["grid_step"]) * np.arange(int(g["grid_n"]))
    idx = [oracle.closest_grid(grid, v) for v in g["closest_in"]]
    assert list(idx) == list(g["closest_idx"])


def test_lte_identity(oracle, golden):
    """(G_abs - G_ind)/Q == S_hitran(T) with E_vib = 0 (spect_classes.py:1806-1842 vs 1856-1863)."""
    g = golden("spcl_scalars")
    for i in range(len(g["T"])):
        G = oracle.calc_gcoeffs(g["nu"][i], g["A"][i], g["El"][i], g["gu"][i], g["gl"][i], 0.0, 0.0, g["T"][i])
        S = oracle.linestrength_hitran(g["A"][i], g["nu"][i], g["T"][i], 1.0, g["gu"][i], g["El"][i])
        assert abs((G[2] - G[1]) - S) <= 1e-12 * abs(S)


def test_tips_partition_sum(oracle, golden):
    """A7: CalcPartitionSum (4-point lagrange through the TIPS-2003 table)."""
    g = golden("tips2003")
    keys = [tuple(k) for k in g["keys"]]
    for mol, iso, T, q in g["samples"]:
        tab = g["q_tab"][keys.index((int(mol), int(iso)))]
        qo = oracle.calc_partition_sum(g["t_grid"], tab, T)
        assert abs(qo - q) <= 1e-13 * abs(q), (mol, iso, T, qo, q)


def test_curgods(oracle, golden):
    """A10 curgods.f:2-98."""
    g = golden("curgods")
    for i in range(3):
        nd, x, vmr, f = (g["%s_%d" % (k, i)] for k in ("nd", "x", "vmr", "f"))
        r = [oracle.curgod(1, nd, x), oracle.curgod(2, nd, x, vmr), oracle.curgod(3, nd, x, vmr, f),
             oracle.curgod(4, nd, x, vmr, f)]
        assert relerr(r, g["res_%d" % i]) < 1e-12


def test_curgod1_analytic(oracle):
    x = np.linspace(0, 3e7, 50)
    nd = 1e15 * np.exp(-x / 5e6)
    want = 1e15 * 5e6 * (1 - np.exp(-3e7 / 5e6))
    assert abs(oracle.curgod(1, nd, x) - want) < 1e-12 * want


def _lines(g):
    return {k[5:]: g[k] for k in g.files if k.startswith("line_")}


@pytest.mark.parametrize("mode", [0, 1])
def test_e2e_ch4_levels(oracle, golden, mode):
    """A2-A8 end to end against the reference Python run (non-LTE levels,
    clipped windows, unidentified / same-level / A=0 lines)."""
    g = golden("e2e_ch4_levels")
    grid = float(g["grid_w0"]) + float(g["grid_step"]) * np.arange(int(g["grid_n"]))
    ab, em = oracle.abscoeff_layers(_lines(g), float(g["mm"]), g["e_lev"], g["temps"], g["press"], g["q_part"],
                                    g["tvib"], grid, mode=mode, n_threads=3 if mode else 1)
    tol = 1e-13 if mode == 0 else 1e-12
    assert relerr(ab, g["abs"]) < tol
    assert relerr(em, g["emi"]) < tol
    ab0, em0 = oracle.abscoeff_layers(_lines(g), float(g["mm"]), g["e_lev"], g["temps"][:1], g["press"][:1],
                                      g["q_part"][:1], None, grid, mode=mode)
    assert relerr(ab0, g["abs_lte0"]) < tol
    assert relerr(em0, g["emi_lte0"]) < tol


def test_e2e_co_all(oracle, golden):
    """BASELINE configs[0] shape: 500 CO-like lines, 1e4 grid, 'all' level set."""
    g = golden("e2e_co_all")
    grid = float(g["grid_w0"]) + float(g["grid_step"]) * np.arange(int(g["grid_n"]))
    sel = g["layer_sel"]
    ab, em = oracle.abscoeff_layers(_lines(g), float(g["mm"]), [], g["temps"][sel], g["press"][sel],
                                    g["q_part"][sel], None, grid, mode=0)
    assert relerr(ab, g["abs"]) < 1e-13
    assert relerr(em, g["emi"]) < 1e-13


def test_sum_all_lines(oracle):
    """A6 lineshape.f:2-25 on a small case (the Fortran's fixed 4 GB argument is not built here)."""
    rng = np.random.default_rng(3)
    rows = rng.random((7, 20))
    init = np.array([1, 5, 31, 81, 2, 60, 11])
    fin = init + 19
    spe = rng.random(100)
    want = spe.copy()
    for r, i in zip(rows, init):
        want[i - 1:i + 19] += r
    assert np.array_equal(oracle.sum_all_lines(spe, rows, init, fin), want)


def test_radiance_slab(oracle):
    """Build's own recursion (parity unpinned): homogeneous slab I = S(1-exp(-tau))."""
    a = np.array([[1e-18, 3e-17, 0.0]])
    e = np.array([[2e-25, 6e-24, 1e-26]])
    u = 4e17
    r = oracle.radiance_ray(a, e, [0], [u])
    tau = a[0] * u
    want = np.where(tau > 0, e[0] / np.where(a[0] > 0, a[0], 1) * (1 - np.exp(-tau)), e[0] * u)
    assert np.allclose(r, want, rtol=1e-13)
    r2 = oracle.radiance_ray(a, e, [0, 0], [u / 2, u / 2])
    assert np.allclose(r2, want, rtol=1e-13)


def test_hires_to_lowres(oracle, golden):
    """N2: SpectralIntensity.hires_to_lowres (spect_classes.py:1180-1191) against the reference run."""
    g = golden("lowres_ils")
    grid = float(g["grid_w0"]) + float(g["grid_step"]) * np.arange(int(g["grid_n"]))
    for u in ("Wm2", "ergscm2", "nWcm2"):
        o = oracle.hires_to_lowres(grid, g["spectrum"], g["centers_nm"], g["widths_nm"], u)
        assert relerr(o, g["low_" + u]) < 1e-13


def _glines(g):
    return {k[5:]: g[k] for k in g.files if k.startswith("line_")}


def _ggrid(g):
    return float(g["grid_w0"]) + float(g["grid_step"]) * np.arange(int(g["grid_n"]))


def test_gcoeff_levels_vs_reference_add_PT(oracle, golden):
    """A5: per-level, per-ctype G spectra against the reference's own LutSet.add_PT -> BuildCoeff run
    (tests/golden/make_golden.py --gcoeff), levels and the 'all' set, and the tracked-level combine."""
    g = golden("gcoeff_levels")
    L, grid = _glines(g), _ggrid(g)
    ab, em, G = oracle.gcoeff_layers(L, float(g["mm"]), g["e_lev"], g["temps"], g["press"], g["q_part"], g["tvib"], grid)
    nz = g["G_lev"] != 0
    assert np.array_equal(G != 0, nz)
    assert relerr(G[nz], g["G_lev"][nz]) < 1e-13
    _, _, Ga = oracle.gcoeff_layers(L, float(g["mm"]), [], g["temps"], g["press"], g["q_part"], None, grid)
    nz = g["G_all"] != 0
    assert np.array_equal(Ga != 0, nz)
    assert relerr(Ga[nz], g["G_all"][nz]) < 1e-13
    lv = int(g["track_level"])
    pop = np.exp(-oracle.constants()["c2"] * g["e_lev"][lv] / g["tvib"][lv]) / g["q_part"]
    ta = G[:, lv, 2] * pop[:, None] - G[:, lv, 1] * pop[:, None]
    assert relerr(ta, g["track_abs"]) < 1e-12
    assert relerr(G[:, lv, 0] * pop[:, None], g["track_emi"]) < 1e-13


def test_outer_lines_vs_reference(oracle, golden):
    """Lines 3.3 - 25 cm-1 outside the grid (humliv_bb's outer branches through the whole coefficient
    path) against the reference run (make_golden.py --outer)."""
    g = golden("e2e_outer_lines")
    L, grid = _glines(g), _ggrid(g)
    for mode in (0, 1):
        ab, em = oracle.abscoeff_layers(L, float(g["mm"]), [], g["temps"], g["press"], g["q_part"], None, grid,
                                        mode=mode, n_threads=3)
        assert relerr(ab, g["abs"]) < 1e-13 and relerr(em, g["emi"]) < 1e-13
    Lo = {k: v[g["outer_sel"]] for k, v in L.items()}
    ab, em = oracle.abscoeff_layers(Lo, float(g["mm"]), [], g["temps"], g["press"], g["q_part"], None, grid, mode=1)
    nz = g["abs_outer_only"] != 0
    assert np.array_equal(ab != 0, nz)
    assert relerr(ab[nz], g["abs_outer_only"][nz]) < 1e-13 and relerr(em[nz], g["emi_outer_only"][nz]) < 1e-13
