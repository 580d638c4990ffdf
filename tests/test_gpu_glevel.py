"""The level-factored route on the GPU (sr_glevel_pairs_dev + sr_glevel_combine_dev): the reference's own structure
for paths whose steps share (P, T) -- per-level G spectra once per (P, T) (spect_main_module.py:1122-1168,
spect_classes.py:1277-1337), then the population-weighted combine per LOS step (:2036-2106, :2200-2276)."""
import os

import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    from spectrobot_amd import engine
    engine.set_device(0)
    return engine


@pytest.fixture(scope="module")
def scene(eng):
    from spectrobot_amd import synthetic as syn
    grid = syn.make_grid(2990.0, 5e-4, 24000)
    L = syn.make_lines(9000, grid, seed=21, n_levels=12, config_id=2)
    atm = syn.make_atmosphere(7, 12)
    ls = eng.LineSet(L, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
    return dict(grid=grid, L=L, atm=atm, ls=ls)


def test_level_pairs_are_the_ctype_spectra_the_combine_uses(eng, scene):
    """tab[L, 0] = Gabs_L - Gind_L and tab[L, 1] = Gsp_L of sr_gcoeff_layers_dev (itself pinned to the reference's
    LutSet.add_PT -> BuildCoeff run, test_gcoeff_levels_golden), for every level; the 'all' set of an iso-molecule
    without levels likewise."""
    import torch
    from spectrobot_amd import synthetic as syn
    ls, atm = scene["ls"], scene["atm"]
    tab = ls.glevel_pairs(atm["temps"], atm["press"])
    assert tuple(tab.shape) == (12, 2, 7, 24000)
    for lv in range(12):
        g = ls.gcoeff_layers(atm["temps"], atm["press"], level=lv)
        a = g[2] - g[1]
        sa = a.abs().amax(dim=1, keepdim=True).clamp_min(1e-300)
        se = g[0].abs().amax(dim=1, keepdim=True).clamp_min(1e-300)
        assert float(((tab[lv, 0] - a).abs() / sa).max()) < 1e-13, lv
        assert float(((tab[lv, 1] - g[0]).abs() / se).max()) < 1e-13, lv
    # an iso-molecule without levels: one pair
    grid = scene["grid"]
    Lc = syn.make_lines(500, grid, seed=3, n_levels=0, co_like=True)
    lc = eng.LineSet(Lc, grid, 5, 1, syn.CO_MM, [])
    t1 = lc.glevel_pairs(atm["temps"], atm["press"])
    g = lc.gcoeff_layers(atm["temps"], atm["press"], level=0)
    assert tuple(t1.shape) == (1, 2, 7, 24000)
    assert float(((t1[0, 0] - (g[2] - g[1])).abs() / (g[2] - g[1]).abs().amax()).max()) < 1e-13
    assert float(((t1[0, 1] - g[0]).abs() / g[0].abs().amax()).max()) < 1e-13
    pop = lc.level_populations(atm["temps"])
    ab, em = eng.glevel_combine(t1, np.arange(7), pop)
    a0, e0 = lc.abscoeff_layers(atm["temps"], atm["press"])
    assert relerr(ab.cpu().numpy(), a0.cpu().numpy()) < 1e-12 and relerr(em.cpu().numpy(), e0.cpu().numpy()) < 1e-12


def test_combine_equals_the_folded_op_step_by_step(eng, oracle, scene):
    """40 LOS steps on 7 (P, T) rows, every step with its own vibrational temperatures (a 3-D path: T_vib by the
    local SZA): the combine of the pair tables against the folded coefficient op run on the steps themselves
    (<= 1e-12 of a row's largest value: the summation order differs) and against the oracle on sampled steps."""
    import torch
    from spectrobot_amd import synthetic as syn
    from test_gpu_configs import _q
    ls, atm, L, grid = scene["ls"], scene["atm"], scene["L"], scene["grid"]
    rng = np.random.default_rng(5)
    n_steps = 40
    row = rng.integers(0, 7, n_steps).astype(np.int32)
    row[:7] = np.arange(7)
    T, P = atm["temps"][row], atm["press"][row]
    tv = atm["tvib"][:, row] + rng.uniform(-15.0, 25.0, (12, n_steps))
    tv[0] = T
    tab = ls.glevel_pairs(atm["temps"], atm["press"])
    pop = ls.level_populations(T, tvib=tv)
    ab, em = eng.glevel_combine(tab, row, pop)
    a0, e0 = ls.abscoeff_layers(T, P, tvib=tv)
    sa, se = a0.abs().amax(dim=1, keepdim=True), e0.abs().amax(dim=1, keepdim=True)
    assert float(((ab - a0).abs() / sa).max()) < 1e-12 and float(((em - e0).abs() / se).max()) < 1e-12
    sel = np.array([0, 9, 23, 39])
    abo, emo = oracle.abscoeff_layers(L, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES, T[sel], P[sel], _q(T[sel]), tv[:, sel], grid,
                                      mode=1, n_threads=4)
    assert float(np.max(np.abs(ab[sel].cpu().numpy() - abo) / np.abs(abo).max(axis=1, keepdims=True))) < 1e-11
    assert relerr(em[sel].cpu().numpy(), emo) < 1e-10
    # populations: the host mirror of smm:2073 against what the folded op used (q_part pinned = the same numbers)
    q = _q(T)
    assert relerr(ls.level_populations(T, tvib=tv, q_part=q), np.exp(-1.4387768775039338 * syn.CH4_LEVEL_ENERGIES[None, :] / tv.T) / q[:, None]) < 1e-14
    # argument checks
    with pytest.raises(RuntimeError):
        eng.glevel_combine(tab, row + 7, pop)
    with pytest.raises(ValueError):
        eng.glevel_combine(tab, row, pop[:, :5])


def test_combine_temperature_derivative(eng, scene):
    """d abs / d T and d emi / d T of every step from two table builds (T and T + dT, boundaries frozen at T) with the
    population part analytic (d pop_L / d T through Q's own interpolant), against the frozen central difference of the
    folded op at +-0.01 K (engine.coefficients_dT).  The error is that of the forward difference of the G spectra
    (the folded op's own two-op scheme lands on the same figures): 1e-7 |c| / dT of staircase noise from the
    reference's single-precision cmplx(ry, -rx) against a derivative of ~0.05 |c| / K."""
    import torch
    ls, atm = scene["ls"], scene["atm"]
    rng = np.random.default_rng(6)
    n_steps = 20
    row = rng.integers(0, 7, n_steps).astype(np.int32)
    T, P = atm["temps"][row], atm["press"][row]
    tv = atm["tvib"][:, row] + rng.uniform(-10.0, 20.0, (12, n_steps))
    co, (da_ref, de_ref) = eng.coefficients_dT(ls, T, P, tvib=tv, scheme="central", dT=0.01)
    _, (da_r5, de_r5) = eng.coefficients_dT(ls, T, P, tvib=tv, coeffs=co, scheme="central", dT=0.05)
    _, (da_f, de_f) = eng.coefficients_dT(ls, T, P, tvib=tv, coeffs=co, scheme="forward")

    def rel(x, y):
        return float(((x - y).abs().amax(dim=1) / y.abs().amax(dim=1)).max())
    print("the two central references against each other: %.1e %.1e" % (rel(da_r5, da_ref), rel(de_r5, de_ref)))
    errs = {}
    for name, dT, lin in (("exact weights, dT 0.002 K", 0.002, False), ("linearised weights, dT 0.05 K", 0.05, True)):
        lf = eng.LevelFactored(ls, atm["temps"], atm["press"], dT=dT, linear_weights=lin)
        (ab, em), (da, de) = lf.steps(row, tvib=tv, derivative=True)
        assert rel(ab, co[0]) < 1e-12 and rel(em, co[1]) < 1e-12
        errs[name] = (min(rel(da, da_ref), rel(da, da_r5)), min(rel(de, de_ref), rel(de, de_r5)))
    print("T derivative of 20 steps against the frozen central difference of 0.01 K: folded forward difference %.1e %.1e; "
          "level-factored, two table builds, analytic populations: %s" % (rel(da_f, da_ref), rel(de_f, de_ref), errs))
    e_x, e_l = errs["exact weights, dT 0.002 K"], errs["linearised weights, dT 0.05 K"]
    assert max(e_x) < 2e-3 and e_x[0] < 1.5 * rel(da_f, da_ref) + 1e-5 and e_x[1] < 1.5 * rel(de_f, de_ref) + 1e-5
    assert max(e_l) < 4e-4 and max(e_l) < max(e_x)        # the linearised weights are what two builds can do
    # LTE (no vibrational temperatures given): the Boltzmann factors follow T too
    pop2, dpop2 = ls.level_populations(T, derivative=True)
    h = 1e-3
    fd = (ls.level_populations(T + h) - ls.level_populations(T - h)) / (2 * h)
    assert relerr(dpop2, fd) < 1e-6


def test_make_abscoeff_isomolec_takes_the_factored_route_for_shared_rows(eng, scene):
    """smm.make_abscoeff_isomolec(useLUTs=False) on a step list whose (P, T) couples repeat (12 steps on 3 rows, every
    step with its own vibrational temperatures): the level-factored route it then takes gives what the folded op gives
    on the steps themselves."""
    import torch
    from spectrobot_amd import spect_main_module as smm, spect_base_module as sbm, synthetic as syn
    ls, atm, grid = scene["ls"], scene["atm"], scene["grid"]
    rng = np.random.default_rng(8)
    row = np.repeat(np.array([1, 3, 5]), 4)
    T, P = atm["temps"][row], atm["press"][row]
    tv = atm["tvib"][:, row] + rng.uniform(-10.0, 20.0, (12, 12))
    tv[0] = T
    iso = sbm.IsoMolec(6, 1, syn.CH4_MM, mol_name="CH4")
    for i, e in enumerate(syn.CH4_LEVEL_ENERGIES):
        iso.add_level("L%02d" % i, e, local_vibtemp=tv[i])
    a_set, e_set = smm.make_abscoeff_isomolec([grid[0], grid[-1]], iso, T, P, LTE=False, lineset=ls, to_host=False)
    a0, e0 = ls.abscoeff_layers(T, P, tvib=tv)
    sa, se = a0.abs().amax(dim=1, keepdim=True), e0.abs().amax(dim=1, keepdim=True)
    assert float(((a_set.device - a0).abs() / sa).max()) < 1e-12 and float(((e_set.device - e0).abs() / se).max()) < 1e-12
    assert not torch.equal(a_set.device, a0)          # (it did take the other route)


def test_level_pairs_on_a_shard_with_lines_beyond_the_grid(eng):
    """The level-factored route on a spectral shard (one rank's [g_lo, g_hi) of a multi-GPU run) and with lines whose
    centre lies outside their own window (the outer branches of humliv_bb): the shard's tables are the whole grid's
    tables cut to the shard, and their combine equals the folded op on the shard -- outer lines included."""
    import torch
    from spectrobot_amd import synthetic as syn
    grid = syn.make_grid(2990.0, 5e-4, 26000)
    L = syn.make_lines(2500, grid, seed=31, n_levels=12, config_id=2)
    step = grid[1] - grid[0]
    # eight lines 3.3 .. 9 cm-1 outside the grid ends: their windows sit on the first / last grid point
    L["freq"][:4] = grid[0] - np.array([3.3, 4.1, 6.0, 9.0])
    L["freq"][-4:] = grid[-1] + np.array([3.3, 3.9, 5.5, 8.0])
    atm = syn.make_atmosphere(5, 12)
    ls = eng.LineSet(L, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
    T, P = atm["temps"], atm["press"] * 20.0      # wide Lorentz wings: the outer lines reach well into the grid
    full = ls.glevel_pairs(T, P)
    lo, hi = 9000, 21000
    part = ls.glevel_pairs(T, P, g_lo=lo, g_hi=hi)
    sc = full[..., lo:hi].abs().amax(dim=-1, keepdim=True).clamp_min(1e-300)
    assert float(((part - full[..., lo:hi]).abs() / sc).max()) < 1e-12
    rng = np.random.default_rng(3)
    row = rng.integers(0, 5, 16).astype(np.int32)
    tv = atm["tvib"][:, row] + rng.uniform(-8.0, 15.0, (12, 16))
    pop = ls.level_populations(T[row], tvib=tv)
    for (a, b, tab) in ((0, 26000, full), (lo, hi, part)):
        ab, em = eng.glevel_combine(tab, row, pop)
        a0, e0 = ls.abscoeff_layers(T[row], P[row], tvib=tv, g_lo=a, g_hi=b)
        sa, se = a0.abs().amax(dim=1, keepdim=True), e0.abs().amax(dim=1, keepdim=True)
        assert float(((ab - a0).abs() / sa).max()) < 1e-12 and float(((em - e0).abs() / se).max()) < 1e-12
    # the outer lines do contribute at the grid ends (the check would be empty otherwise)
    Li = {k: v[4:-4] for k, v in L.items()}
    lsi = eng.LineSet(Li, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
    a_in, _ = lsi.abscoeff_layers(T[:1], P[:1], tvib=atm["tvib"][:, :1])
    a_all, _ = ls.abscoeff_layers(T[:1], P[:1], tvib=atm["tvib"][:, :1])
    assert float(((a_all - a_in).abs() / a_all.abs()).max()) > 1e-6


def _plane_err(a, b):
    """max |a - b| relative to the largest |b| of each (level, channel, row) spectrum."""
    s = b.abs().amax(dim=-1, keepdim=True).clamp_min(1e-300)
    return float(((a - b).abs() / s).max())


@pytest.mark.parametrize("case", ["dense", "sparse_shard", "frozen_linear", "outer_lines", "low_pressure", "forty_levels"])
def test_multichannel_route_equals_the_per_level_route(eng, case):
    """Round 6: the level tables by the multi-channel pass (every line ONCE: sr_zones_mc_kernel / sr_wings_mc_kernel add
    its three weighted contributions to the LDS planes of its two levels; far field by far-only passes of the level
    sub-linesets) against one coefficient op per level (sr_set_level_route(0): the route of rounds 4-5, itself pinned
    to the reference's add_PT -> BuildCoeff run by test_gcoeff_levels_golden).  Pair tables AND the three ctypes; whole
    grids and shards whose lines reach beyond the grid; frozen boundaries with linearised weights (the T + dT build of
    configs[3]); lines whose centre lies outside their window; Doppler-dominated rows (wide region-3 cores); an
    iso-molecule with forty levels (80 / 120 planes: the zones kernel's 128-point images).  The two routes differ by the
    order of summation only: <= 2e-12 of a spectrum's largest value."""
    import torch
    from spectrobot_amd import synthetic as syn
    n_grid, n_lines, nl, lo, hi, w0 = 40000, 30000, 6, 0, None, 2985.0
    kw = {}
    if case == "sparse_shard":
        n_lines, lo, hi = 5000, 3001, 33333
    elif case == "outer_lines":
        n_grid, n_lines, lo, hi = 20000, 8000, 0, None
    n_lev, e_lev = 12, syn.CH4_LEVEL_ENERGIES
    if case == "forty_levels":
        n_grid, n_lines, nl, n_lev = 12000, 9000, 3, 40
        e_lev = np.concatenate([[0.0], np.linspace(1300.0, 6000.0, 39)])
    grid = syn.make_grid(w0, 5e-4, n_grid)
    L = syn.make_lines(n_lines, grid, seed=61, n_levels=n_lev, config_id=2)
    if case == "outer_lines":   # some lines up to 6 cm-1 beyond the grid ends: their windows sit on the end points
        rng = np.random.default_rng(4)
        k = rng.choice(n_lines, 400, replace=False)
        L = dict(L)
        L["freq"] = L["freq"].copy()
        L["freq"][k[:200]] = grid[0] - rng.uniform(0.0, 6.0, 200)
        L["freq"][k[200:]] = grid[-1] + rng.uniform(0.0, 6.0, 200)
    atm = syn.make_atmosphere(nl, 12)
    T, P = atm["temps"], atm["press"] * (1e-3 if case == "low_pressure" else 1.0)
    ls = eng.LineSet(L, grid, 6, 1, syn.CH4_MM, e_lev)
    try:
        if case == "frozen_linear":
            ls.set_bounds_temps(T, linear_weights=True)
            T = T + 0.02
        res = {}
        for route in (1, 0):
            eng.set_level_route(route)
            res[route] = (ls.glevel_pairs(T, P, g_lo=lo, g_hi=hi), ls.gcoeff_levels(T, P, g_lo=lo, g_hi=hi))
            torch.cuda.synchronize()
    finally:
        eng.set_level_route(1)
        ls.set_bounds_temps(None)
    assert float(res[0][0].abs().max()) > 0 and float(res[0][1].abs().max()) > 0
    assert _plane_err(res[1][0], res[0][0]) < 2e-12, case
    assert _plane_err(res[1][1], res[0][1]) < 2e-12, case
    # the per-level entry point of the three ctypes is untouched by the route: level 5 of the all-levels call
    if case != "frozen_linear":   # (the boundaries were released above)
        assert _plane_err(res[1][1][5], ls.gcoeff_layers(T, P, level=5, g_lo=lo, g_hi=hi)) < 2e-12


def test_multichannel_route_in_row_batches_and_after_other_calls(eng):
    """The multi-channel pass under a small table budget (row batches: every batch writes its rows of every channel)
    and interleaved with folded ops on the same handle (shared scratch of the level sub-linesets, own tables of the
    pass) on unsynchronised streams: same tables as one unbatched call."""
    import torch
    from spectrobot_amd import synthetic as syn
    grid = syn.make_grid(2990.0, 5e-4, 16000)
    L = syn.make_lines(6000, grid, seed=8, n_levels=12, config_id=2)
    atm = syn.make_atmosphere(9, 12)
    ls = eng.LineSet(L, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
    ref = ls.glevel_pairs(atm["temps"], atm["press"])
    a0, e0 = ls.abscoeff_layers(atm["temps"], atm["press"], tvib=atm["tvib"])
    torch.cuda.synchronize()
    try:
        eng.set_table_budget(6000 * 112 * 3 * 4 + 4 * 16000 * 100)   # a few rows per batch
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        with torch.cuda.stream(s1):
            t1 = ls.glevel_pairs(atm["temps"], atm["press"])
        with torch.cuda.stream(s2):
            a1, e1 = ls.abscoeff_layers(atm["temps"], atm["press"], tvib=atm["tvib"])
        with torch.cuda.stream(s1):
            t2 = ls.glevel_pairs(atm["temps"], atm["press"])
        torch.cuda.synchronize()
    finally:
        eng.set_table_budget(48 << 30)
    assert _plane_err(t1, ref) < 2e-12 and _plane_err(t2, ref) < 2e-12
    assert torch.equal(a1, a0) and torch.equal(e1, e0)


@pytest.mark.parametrize("far", [1, 2])
def test_multichannel_route_under_forced_far_field_modes(eng, far):
    """sr_set_far_field(1) / (2) force the per-line expansions / the box-pair chain for EVERY sub-lineset: the far-only
    passes of the multi-channel route then all run through coef_op's far-only option on their own CoefWork lanes (no
    sparse batch: that is mode 3's rule), and the tables must still equal the per-level route's under the same mode and
    the default mode's."""
    import torch
    from spectrobot_amd import synthetic as syn
    grid = syn.make_grid(2988.0, 5e-4, 20000)
    L = syn.make_lines(12000, grid, seed=17, n_levels=12, config_id=2)
    atm = syn.make_atmosphere(5, 12)
    ls = eng.LineSet(L, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
    ref = ls.glevel_pairs(atm["temps"], atm["press"])
    try:
        eng.set_far_field(far)
        t_mc = ls.glevel_pairs(atm["temps"], atm["press"])
        eng.set_level_route(0)
        t_pl = ls.glevel_pairs(atm["temps"], atm["press"])
        torch.cuda.synchronize()
    finally:
        eng.set_level_route(1)
        eng.set_far_field(eng.FAR_FIELD_DEFAULT)
    assert _plane_err(t_mc, t_pl) < 2e-12 and _plane_err(t_mc, ref) < 5e-12
