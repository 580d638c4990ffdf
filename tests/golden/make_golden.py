#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference and amdflang); the
fixtures it writes are committed, this script is committed, nothing of the
reference is.  Two sources of truth are exercised:

  * the reference's Fortran (lineshape.f, curgods.f, fparts_mod.f) compiled as
    they lie under /root/reference into oracle/_ref/ (make -C oracle ref) and
    called through ctypes (oracle/ref_fortran.py);
  * the reference's Python, spect_classes.py, imported under Python 3 with the
    f2py modules replaced by those ctypes wrappers and with a stub for the
    module the reference tree does not contain (spect_base_module: only
    isclose / extract_quanta_HITRAN / find_molec_metadata are touched here).

Outputs (all numpy .npz, inputs + expected outputs side by side):
  humliv_windows.npz   A1  humliv_bb on 13010-point windows, many (lw, dw, x0)
  tips2003.npz         A7  bd_tips_2003 tables + CalcPartitionSum samples
  curgods.npz          A10 curgod_fort_1..4 on three profiles
  spcl_scalars.npz     A3/A4 widths, G-coefficients, Planck, LTE line strength
  e2e_ch4_levels.npz   A2-A8 end to end, non-LTE levels, 3 (P,T), clipped windows
  e2e_co_all.npz       A2-A8 end to end, 'all' level set (BASELINE configs[0] shape)
"""
import math
import os
import pickle
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

REF = "/root/reference"


def import_reference_spcl():
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    from oracle import ref_fortran as RF
    import matplotlib
    matplotlib.use("Agg")

    m_ls = types.ModuleType("lineshape")
    m_ls.humliv_bb = lambda x, i1, i2, x0, lw, dw: RF.humliv_bb(x, i1, i2, x0, lw, dw)
    m_fp = types.ModuleType("fparts_mod")
    m_fp.bd_tips_2003 = lambda mol, iso: RF.bd_tips_2003(mol, iso)
    m_sbm = types.ModuleType("spect_base_module")
    m_sbm.isclose = lambda a, b, rtol=1e-9, atol=0.0: bool(np.isclose(a, b, rtol=rtol, atol=atol))
    # the level label of a synthetic line IS its minimal string
    m_sbm.extract_quanta_HITRAN = lambda mol, iso, s: (s.strip(), None, None)
    m_sbm.find_molec_metadata = lambda mol, iso: {"iso_MM": float("nan"), "iso_ratio": float("nan")}
    sys.modules.update({"lineshape": m_ls, "fparts_mod": m_fp, "spect_base_module": m_sbm,
                        "cPickle": pickle})
    sys.path.insert(0, REF)
    import spect_classes as spcl  # the reference, unmodified
    return spcl, RF


class Level(object):
    def __init__(self, name, energy):
        self.name, self.energy = name, energy

    def minimal_level_string(self):
        return self.name


class IsoMolec(object):
    def __init__(self, mol, iso, MM, level_energies):
        self.mol, self.iso, self.MM = mol, iso, MM
        self.levels = []
        for i, e in enumerate(level_energies):
            nm = "lev_%02d" % i
            self.levels.append(nm)
            setattr(self, nm, Level("L%02d" % i, float(e)))


def ref_lines(spcl, mol, iso, L, labels=True):
    out = []
    for i in range(len(L["freq"])):
        up = ("L%02d" % L["lev_up"][i]) if (labels and L["lev_up"][i] >= 0) else "??"
        lo = ("L%02d" % L["lev_lo"][i]) if (labels and L["lev_lo"][i] >= 0) else "??"
        vals = [mol, iso, float(L["freq"][i]), 0.0, float(L["a_coeff"][i]), float(L["air_broad"][i]), 0.0,
                float(L["e_lower"][i]), float(L["t_dep_broad"][i]), 0.0, up, lo, "", "", "",
                float(L["g_up"][i]), float(L["g_lo"][i])]
        out.append(spcl.SpectLine(vals, nomi=spcl.cose_hit))
    return out


def ref_abscoeff(spcl, grid, lines, isomolec, temps, press, tvib):
    """The reference's useLUTs=False path with its own objects: PrepareCalcShapes
    (spcl:1440) after the LinkToMolec filter of calc_shapes_lines (spcl:1384-1388),
    one SpectralGcoeff per level and ctype filled as BuildCoeff selects lines
    (spcl:1304-1321) -- accumulated with the reference's NumPy accumulate
    add_to_spectrum (spcl:929) because add_lines_to_spectrum needs Python-2
    integer division and a 4 GB matrix -- then the population-weighted combine of
    make_abscoeff_isomolec (smm:2036-2080) with the reference's operators."""
    sg = spcl.SpectralGrid(grid, units="cm_1")
    ctypes_ = ["sp_emission", "ind_emission", "absorption"]
    abs_out, emi_out, integ = [], [], []
    for k, (P, T) in enumerate(zip(press, temps)):
        lin = lines
        if len(isomolec.levels) > 0:
            oks = [l.LinkToMolec(isomolec) for l in lin]
            lin = [l for l, ok in zip(lin, oks) if ok]
        lin = spcl.PrepareCalcShapes(sg, lin, T, P, isomolec)
        integ.append([l.shape.integrate() for l in lin[:5]])
        zero = lambda: spcl.SpectralObject(np.zeros(len(grid)), sg)
        abs_c, emi_c = zero(), zero()
        Q = spcl.CalcPartitionSum(isomolec.mol, isomolec.iso, temp=T)
        if len(isomolec.levels) == 0:
            G = {}
            for ct in ctypes_:
                g = spcl.SpectralGcoeff(ct, sg, isomolec.mol, isomolec.iso, isomolec.MM, "",
                                        unidentified_lines=True)
                for l in lin:
                    g.add_to_spectrum(l.shape, Strength=l.G_coeffs[ct])
                G[ct] = g
            pop = 1 / Q
            abs_c += G["absorption"] * pop
            abs_c -= G["ind_emission"] * pop
            emi_c += G["sp_emission"] * pop
        else:
            for li, lev in enumerate(isomolec.levels):
                levello = getattr(isomolec, lev)
                G = {}
                for ct in ctypes_:
                    g = spcl.SpectralGcoeff(ct, sg, isomolec.mol, isomolec.iso, isomolec.MM,
                                            levello.minimal_level_string())
                    if ct in ("sp_emission", "ind_emission"):
                        sel = [l for l in lin if g.lev_string == l.minimal_level_string_up()]
                    else:
                        sel = [l for l in lin if g.lev_string == l.minimal_level_string_lo()]
                    for l in sel:
                        g.add_to_spectrum(l.shape, Strength=l.G_coeffs[ct])
                    G[ct] = g
                vibt = T if tvib is None else tvib[li][k]
                pop = spcl.Boltz_ratio_nodeg(levello.energy, vibt) / Q
                abs_c += G["absorption"] * pop
                abs_c -= G["ind_emission"] * pop
                emi_c += G["sp_emission"] * pop
        abs_out.append(abs_c.spectrum.copy())
        emi_out.append(emi_c.spectrum.copy())
    return np.array(abs_out), np.array(emi_out), np.array(integ)


def main():
    spcl, RF = import_reference_spcl()
    from spectrobot_amd import synthetic as syn
    rng = np.random.default_rng(20260001)
    ln2 = math.log(2.0)

    consts = dict(h_cgs=spcl.h_cgs, c_cgs=spcl.c_cgs, k_cgs=spcl.k_cgs, c2=spcl.c2,
                  hpa_to_atm=spcl.hpa_to_atm, T_ref=spcl.T_ref)
    print("constants", consts)

    # ---------------- A1: humliv windows (compiled Fortran) ----------------
    grid = syn.make_grid(2975.0, 5e-4, 100000)
    sg = spcl.SpectralGrid(grid, units="cm_1")
    lin_grid = np.arange(-spcl.imxsig * sg.step() / 2, spcl.imxsig * sg.step() / 2, sg.step(), dtype=float)
    assert len(lin_grid) == 13010
    cases = []
    # (T, P_hPa, n_air, gamma, MM) -> ry from ~1e-8 (Doppler) to ~80 (Lorentz)
    for P in (2e-7, 1e-4, 1e-2, 0.5, 3.0, 10.0, 150.0, 1013.25, 8000.0):
        for frac in (0.0, 0.37):
            T = float(rng.uniform(90, 200))
            nu0 = float(grid[40000] + frac * sg.step() + rng.integers(0, 2000) * sg.step())
            cases.append((T, P, float(rng.uniform(.55, .85)), float(rng.uniform(.04, .08)), 16.0313, nu0))
    cases.append((70.0, 1.0, 0.7, 0.06, 27.994915, float(grid[50000] - 0.49 * sg.step())))  # CO-like
    X, Y, par = [], [], []
    for (T, P, n_air, gam, MM, nu0) in cases:
        ic, fr = spcl.closest_grid(sg, nu0)
        x = lin_grid + fr
        lw = spcl.Lorenz_width(T, spcl.convert_to_atm(P), n_air, gam)
        dw = spcl.Doppler_width(T, MM, nu0)
        y = RF.humliv_bb(x, 1, 13010, nu0, lw, dw / math.sqrt(ln2))
        X.append(x)
        Y.append(y)
        par.append([nu0, lw, dw / math.sqrt(ln2), T, P, n_air, gam, MM, ic])
    # outer branches of humliv_bb (x0 outside the window) and the scalar humli_bb
    xo = lin_grid + grid[50000]
    outer = []
    for x0, lw, dwp in ((xo[0] - 0.013, 2e-3, 4e-3), (xo[0], 1e-4, 4e-3), (xo[-1] + 0.02, 3e-3, 3.5e-3),
                        (xo[-1], 5e-2, 4e-3), (xo[0] - 5.0, 1e-3, 4e-3), (xo[-1] + 4.0, 1e-3, 4e-3)):
        outer.append((x0, lw, dwp, RF.humliv_bb(xo, 1, 13010, x0, lw, dwp)))
    rxs = rng.uniform(0, 20, 64)
    rys = 10.0 ** rng.uniform(-6, 1.3, 64)
    hum = np.array([RF.humli_bb(float(a), float(b)) for a, b in zip(rxs, rys)])
    np.savez_compressed(os.path.join(HERE, "humliv_windows.npz"), x=np.array(X), y=np.array(Y),
                        par=np.array(par), par_names="nu0 lw dw_over_sqrtln2 T P n_air gamma MM ic",
                        outer_x=xo, outer_par=np.array([o[:3] for o in outer]),
                        outer_y=np.array([o[3] for o in outer]), humli_rx=rxs, humli_ry=rys, humli_y=hum)
    print("humliv_windows: %d cases, ry range %.2e..%.2e" % (len(cases), min(p[1] / p[2] for p in par),
                                                            max(p[1] / p[2] for p in par)))

    # ---------------- A7: TIPS tables + CalcPartitionSum ----------------
    tips = {}
    molparam_isos = {1: 6, 2: 9, 3: 5, 4: 5, 5: 6, 6: 3, 7: 3, 8: 3, 9: 2, 10: 1, 11: 2, 12: 1, 13: 3,
                     14: 1, 15: 2, 16: 2, 17: 1, 18: 2, 19: 5, 20: 3, 21: 2, 22: 1, 23: 3, 24: 2,
                     25: 1, 26: 2, 27: 1, 28: 1, 29: 1, 31: 3, 32: 1, 33: 1, 34: 1, 35: 2, 36: 1,
                     37: 2, 38: 2}
    keys, gis, tabs = [], [], []
    tgrid = None
    for mol, niso in molparam_isos.items():
        for iso in range(1, niso + 1):
            gi, t, q = RF.bd_tips_2003(mol, iso)
            if not np.all(np.isfinite(q)) or q[0] <= 0:
                continue
            tgrid = t
            keys.append((mol, iso))
            gis.append(gi)
            tabs.append(q)
    smp = []
    for (mol, iso) in ((6, 1), (5, 1), (23, 1), (26, 1), (27, 1), (38, 1), (6, 2)):
        for T in (60.0, 70.0, 84.99, 85.0, 110.0, 149.3, 175.0, 212.654, 296.0, 300.0, 1000.0):
            smp.append([mol, iso, T, float(spcl.CalcPartitionSum(mol, iso, temp=T))])
    np.savez_compressed(os.path.join(HERE, "tips2003.npz"), keys=np.array(keys), gi=np.array(gis),
                        t_grid=tgrid, q_tab=np.array(tabs), samples=np.array(smp))
    print("tips2003: %d (mol,iso) tables; Q(CH4,150)=%r" % (len(keys), spcl.CalcPartitionSum(6, 1, 150.0)))

    # ---------------- A10: curgods ----------------
    cg = []
    for n_p, H in ((6, 45.0), (40, 60.0), (300, 33.0)):
        xk = np.sort(rng.uniform(0, 400, n_p)) * 1e5  # cm
        z = np.linspace(100, 500, n_p)
        nd = 1e16 * np.exp(-(z - 100) / H) * (1 + 0.05 * rng.standard_normal(n_p))
        vmr = 1e-2 * (1 + 0.3 * np.sin(z / 50.0))
        f = 0.5 + 0.4 * np.cos(z / 70.0)
        r = [RF.curgod(1, nd, xk), RF.curgod(2, nd, xk, vmr), RF.curgod(3, nd, xk, vmr, f),
             RF.curgod(4, nd, xk, vmr, f)]
        cg.append(dict(nd=nd, x=xk, vmr=vmr, f=f, res=np.array(r)))
    np.savez_compressed(os.path.join(HERE, "curgods.npz"),
                        **{"%s_%d" % (k, i): v for i, c in enumerate(cg) for k, v in c.items()})

    # ---------------- A3/A4: scalar functions of spect_classes.py ----------------
    n = 64
    T = rng.uniform(60, 320, n)
    P = 10.0 ** rng.uniform(-7, 3, n)
    n_air = rng.uniform(.4, .9, n)
    gam = rng.uniform(.02, .1, n)
    MM = rng.choice([16.0313, 27.994915, 27.010899, 26.01565], n)
    nu = rng.uniform(600, 4500, n)
    A = 10.0 ** rng.uniform(-3, 2, n)
    El = rng.uniform(0, 3000, n)
    gu = rng.integers(1, 200, n).astype(float)
    gl = rng.integers(1, 200, n).astype(float)
    Evu = rng.uniform(0, 3000, n)
    Evl = rng.uniform(0, 1500, n)
    lw = np.array([spcl.Lorenz_width(T[i], spcl.convert_to_atm(P[i]), n_air[i], gam[i]) for i in range(n)])
    dw = np.array([spcl.Doppler_width(T[i], MM[i], nu[i]) for i in range(n)])
    G = np.zeros((n, 3))
    S = np.zeros(n)
    for i in range(n):
        l = spcl.SpectLine([6, 1, nu[i], 0.0, A[i], gam[i], 0.0, El[i], n_air[i], 0.0, "a", "b", "", "", "",
                            gu[i], gl[i]], nomi=spcl.cose_hit)
        G[i] = [spcl.Einstein_A_to_Gcoeff_spem(l, T[i], Evu[i]), spcl.Einstein_A_to_Gcoeff_indem(l, T[i], Evu[i]),
                spcl.Einstein_A_to_Gcoeff_abs(l, T[i], Evl[i])]
        S[i] = spcl.Einstein_A_to_LineStrength_hitran(A[i], nu[i], T[i], 1.0, gu[i], El[i])
    bb = np.array([spcl.Calc_BB_single(nu[i], T[i]) for i in range(n)])
    br = np.array([spcl.Boltz_ratio_nodeg(El[i], T[i]) for i in range(n)])
    cgi = np.array([spcl.closest_grid(sg, float(v))[0] for v in
                    list(grid[1000] + sg.step() * np.array([0.0, 0.5, 0.4999, 0.5001, -0.5, 1e-9])) +
                    [grid[0] - 1.0, grid[-1] + 1.0]])
    cgv = np.array(list(grid[1000] + sg.step() * np.array([0.0, 0.5, 0.4999, 0.5001, -0.5, 1e-9])) +
                   [grid[0] - 1.0, grid[-1] + 1.0])
    np.savez_compressed(os.path.join(HERE, "spcl_scalars.npz"), T=T, P=P, n_air=n_air, gam=gam, MM=MM, nu=nu,
                        A=A, El=El, gu=gu, gl=gl, Evu=Evu, Evl=Evl, lw=lw, dw=dw, G=G, S_hitran_Q1=S, bb=bb,
                        boltz=br, closest_in=cgv, closest_idx=cgi, grid_w0=grid[0], grid_step=sg.step(),
                        grid_n=len(grid), **{"const_" + k: v for k, v in consts.items()})

    # ---------------- end to end, CH4-like non-LTE levels ----------------
    n_grid = 16000  # 8 cm^-1: windows (6.5 cm^-1) are clipped at one or both ends for most lines
    g2 = syn.make_grid(2990.0, 5e-4, n_grid)
    L = syn.make_lines(200, g2, config_id=101, n_levels=4)
    # pin the quirks: unidentified levels and lev_up == lev_lo lines are dropped
    L["lev_up"][5] = -1
    L["lev_lo"][9] = -1
    L["lev_lo"][12] = L["lev_up"][12]
    L["lev_lo"][40] = L["lev_up"][40]
    L["a_coeff"][17] = 0.0            # 'linea non defined' -> zero G (spcl:326-337)
    L["freq"][0] = g2[0] + 1e-7       # window clipped left
    L["freq"][-1] = g2[-1] - 1e-7     # window clipped right
    L["freq"] = np.sort(L["freq"])
    e_lev = np.array([0.0, 1311.0, 1533.0, 3019.0])
    iso = IsoMolec(6, 1, syn.CH4_MM, e_lev)
    temps = np.array([172.3, 148.9, 131.0])
    press = np.array([8.0, 0.31, 2.2e-5])
    tvib = np.array([temps, temps + 11.0, temps + 23.5, temps + 35.25])
    lines = ref_lines(spcl, 6, 1, L)
    ab, em, integ = ref_abscoeff(spcl, g2, lines, iso, temps, press, tvib)
    ab_lte, em_lte, _ = ref_abscoeff(spcl, g2, ref_lines(spcl, 6, 1, L), iso, temps[:1], press[:1], None)
    qpart = np.array([float(spcl.CalcPartitionSum(6, 1, temp=t)) for t in temps])
    np.savez_compressed(os.path.join(HERE, "e2e_ch4_levels.npz"), grid_w0=g2[0], grid_step=g2[1] - g2[0],
                        grid_n=n_grid, mm=syn.CH4_MM, mol=6, iso=1, e_lev=e_lev, temps=temps, press=press,
                        tvib=tvib, q_part=qpart, abs=ab, emi=em, abs_lte0=ab_lte, emi_lte0=em_lte,
                        shape_integrals=integ, **{"line_" + k: v for k, v in L.items()})
    print("e2e_ch4_levels: abs max %.3e, shape integrals %s" % (ab.max(), integ[0][:3]))

    # ---------------- end to end, CO-like 'all' set (BASELINE configs[0] shape) ----------------
    g3 = syn.make_grid(2100.0, 5e-4, 10000)
    Lc = syn.make_lines(500, g3, config_id=1, n_levels=0, co_like=True)
    iso_co = IsoMolec(5, 1, syn.CO_MM, [])
    atm = syn.make_atmosphere(40, 0)
    sel = np.array([0, 7, 15, 23, 31, 39])
    abc, emc, _ = ref_abscoeff(spcl, g3, ref_lines(spcl, 5, 1, Lc, labels=False), iso_co, atm["temps"][sel],
                               atm["press"][sel], None)
    qco = np.array([float(spcl.CalcPartitionSum(5, 1, temp=t)) for t in atm["temps"]])
    np.savez_compressed(os.path.join(HERE, "e2e_co_all.npz"), grid_w0=g3[0], grid_step=g3[1] - g3[0], grid_n=10000,
                        mm=syn.CO_MM, mol=5, iso=1, temps=atm["temps"], press=atm["press"], q_part=qco,
                        layer_sel=sel, abs=abc, emi=emc, **{"line_" + k: v for k, v in Lc.items()})
    print("e2e_co_all: abs max %.3e" % abc.max())


if __name__ == "__main__" and not any(a in sys.argv for a in ("--lowres", "--hitran", "--inversion", "--retrieval", "--gcoeff", "--outer", "--containers")):
    main()


def golden_lowres():
    """N2: SpectralIntensity.hires_to_lowres (spect_classes.py:1180-1191) = unit/grid conversion
    cm-1 -> nm, Gaussian ILS (5 sigma) by np.trapz on the irregular nm grid, conversion of the
    result to the observation's units.  Run with the reference's own classes."""
    spcl, RF = import_reference_spcl()
    from spectrobot_amd import synthetic as syn
    rng = np.random.default_rng(20260002)
    grid = syn.make_grid(2990.0, 5e-4, 24000)  # 12 cm-1
    spec = np.abs(rng.standard_normal(24000)) * 1e-3 + 1e-3 * np.exp(-((grid - 2996.0) / 0.05) ** 2)
    centers = np.array([3328.5, 3331.0, 3333.3, 3336.0, 3338.2, 3340.9, 3343.0])  # nm
    widths = np.array([0.8, 1.1, 0.9, 1.3, 1.0, 0.7, 1.2])                            # nm (gaussian sigma)

    class Obs(object):
        pass
    out = {}
    for units in ("Wm2", "nWcm2", "ergscm2"):
        hi = spcl.SpectralIntensity(spec.copy(), spcl.SpectralGrid(grid, units="cm_1"), units="ergscm2")
        obs = Obs()
        obs.spectral_grid = spcl.SpectralGrid(centers, units="nm")
        obs.units = units
        low = hi.hires_to_lowres(obs, spectral_widths=list(widths))
        out[units] = np.array(low.spectrum, dtype=float)
        assert low.units == units
    np.savez_compressed(os.path.join(HERE, "lowres_ils.npz"), grid_w0=grid[0], grid_step=grid[1] - grid[0],
                        grid_n=len(grid), spectrum=spec, centers_nm=centers, widths_nm=widths,
                        low_Wm2=out["Wm2"], low_nWcm2=out["nWcm2"], low_ergscm2=out["ergscm2"])
    print("lowres_ils:", out["Wm2"])


if __name__ == "__main__" and "--lowres" in sys.argv:
    golden_lowres()


def golden_hitran():
    """N3: read_line_database (spect_classes.py:1532-1601) on a HITRAN-2012 160-column file written by
    the reference's own SpectLine.Print_hitran (spect_classes.py:100-109) from synthetic values."""
    spcl, RF = import_reference_spcl()
    rng = np.random.default_rng(20260003)
    path = os.path.join(HERE, "hitran_sample.par")
    n = 60
    freqs = np.sort(rng.uniform(2900.0, 3100.0, n))
    with open(path, "w") as f:
        for i in range(n):
            mol, iso = (6, 1) if i % 5 else (23, 2)
            vals = [mol, iso, float(freqs[i]), float(10 ** rng.uniform(-28, -19)), float(10 ** rng.uniform(-3, 2)),
                    float(rng.choice([0.0, rng.uniform(0.03, 0.09)])), float(rng.choice([0.0, rng.uniform(0.05, 0.1)])),
                    float(rng.uniform(0, 3000)), float(rng.uniform(0.4, 0.9)), float(rng.uniform(-0.01, 0.0)),
                    "    0 0 1 0 1F2", "    0 0 0 0 1A1", ("    %2d F2 %2d" % (i % 20, i % 7)).ljust(15)[:15],
                    ("    %2d F1 %2d" % ((i + 1) % 20, i % 5)).ljust(15)[:15], " 465540 5 6 2 2 1 0", float(2 * (i % 20) + 1) * 3.0,
                    float(2 * ((i + 1) % 20) + 1) * 5.0]
            l = spcl.SpectLine(vals, nomi=spcl.cose_hit)
            l.Print_hitran(ofile=f)
    lines = spcl.read_line_database(path)
    sel = spcl.read_line_database(path, mol=6, iso=1, freq_range=[2950.0, 3050.0])
    frac = spcl.read_line_database(path, fraction_to_keep=0.5)
    fields = ("Mol", "Iso", "Freq", "Strength", "A_coeff", "Air_broad", "Self_broad", "E_lower", "T_dep_broad",
              "P_shift", "g_up", "g_lo")
    out = {k: np.array([getattr(l, k) for l in lines], dtype=float) for k in fields}
    out["Up_lev_str"] = np.array([l.Up_lev_str.decode() for l in lines])
    out["Q_num_lo"] = np.array([l.Q_num_lo.decode() for l in lines])
    out["sel_freq"] = np.array([l.Freq for l in sel])
    out["frac_freq"] = np.array([l.Freq for l in frac])
    np.savez_compressed(os.path.join(HERE, "hitran_sample.npz"), **out)
    print("hitran_sample: %d lines, %d selected, %d kept at fraction 0.5" % (len(lines), len(sel), len(frac)))


if __name__ == "__main__" and "--hitran" in sys.argv:
    golden_hitran()


def golden_inversion():
    """N4 (algebra only): inversion_algebra / chicalc (spect_main_module.py:3399-3469) run with the
    reference's own code on a small synthetic problem; bayes_set is a stub exposing the five methods
    the function calls."""
    spcl, RF = import_reference_spcl()
    m_mp = types.ModuleType("memory_profiler")
    m_mp.profile = lambda f: f
    sys.modules["memory_profiler"] = m_mp
    import spect_main_module as smm
    rng = np.random.default_rng(20260004)
    n_obs, n_par = 40, 6
    K = rng.standard_normal((n_obs, n_par)) * 1e-7
    xi = rng.uniform(1, 2, n_par)
    x_ap = xi * (1 + 0.1 * rng.standard_normal(n_par))
    A = rng.standard_normal((n_par, n_par))
    S_ap = A @ A.T * 0.01 + np.eye(n_par) * 0.04
    noise = np.abs(rng.standard_normal(n_obs)) * 1e-8 + 2e-8
    sim = np.abs(rng.standard_normal(n_obs)) * 1e-6
    obs = sim + K @ (0.05 * rng.standard_normal(n_par)) + noise * rng.standard_normal(n_obs)

    class BS(object):
        def build_jacobian(self, masks=None): return K
        def param_vector(self): return xi
        def VCM_apriori(self): return S_ap
        def apriori_vector(self): return x_ap
        def update_params(self, dx): self.dx = np.array(dx)
        def store_avk(self, a): self.avk = np.array(a)
        def store_VCM(self, s): self.vcm = np.array(s)

    class Sp(object):
        def __init__(self, v): self.spectrum = v
    half = n_obs // 2
    o = [Sp(obs[:half]), Sp(obs[half:])]
    s = [Sp(sim[:half]), Sp(sim[half:])]
    nz = [Sp(noise[:half]), Sp(noise[half:])]
    res = {}
    for lam in (0.1, 1.0):
        bs = BS()
        smm.inversion_algebra(o, s, nz, bs, lambda_LM=lam)
        res["dx_%g" % lam], res["avk_%g" % lam], res["vcm_%g" % lam] = bs.dx, bs.avk, bs.vcm
    chi = smm.chicalc(o, s, nz, None, n_par)
    np.savez_compressed(os.path.join(HERE, "inversion_algebra.npz"), K=K, xi=xi, x_ap=x_ap, S_ap=S_ap, noise=noise,
                        sim=sim, obs=obs, chi=chi, **res)
    print("inversion_algebra: chi", chi, "dx", res["dx_0.1"][:3])


if __name__ == "__main__" and "--inversion" in sys.argv:
    golden_inversion()


def golden_retrieval():
    """N4 (parameter space) and N2 (FOV): the reference's own BayesSet / RetSet / RetParam /
    LinearProfile_1D_new / alt_triangle / lat_box / centre_boxes (spect_main_module.py:169-665) and
    FOV_integr_1D (:3342-3374), run under Python 3.  The absent spect_base_module is stubbed with the
    least that these call: AtmGrid / AtmGridMask as plain holders, rad = degrees -> radians."""
    spcl, RF = import_reference_spcl()
    m_mp = types.ModuleType("memory_profiler")
    m_mp.profile = lambda f: f
    sys.modules["memory_profiler"] = m_mp
    sbm = sys.modules["spect_base_module"]

    class AtmGrid(object):
        def __init__(self, name, coords):
            self.grid = [np.array(coords, dtype=float)]
            self.coords = {name: self.grid[0]}

    class AtmGridMask(object):
        def __init__(self, grid, mask, interp):
            self.grid, self.mask, self.interp = grid, np.array(mask, dtype=float), interp

    sbm.AtmGrid, sbm.AtmGridMask = AtmGrid, AtmGridMask
    sbm.rad = lambda deg: deg * np.pi / 180.0
    import spect_main_module as smm
    rng = np.random.default_rng(20260005)
    out = {}

    alts = np.linspace(100.0, 890.0, 80)
    alt_grid = AtmGrid("alt", alts)
    nodes_a = [150.0, 300.0, 450.0, 600.0, 800.0]
    ap_a = [1.5e-2, 1.4e-2, 1.2e-2, 1.0e-2, 0.8e-2]
    er_a = [0.5e-2, 0.5e-2, 0.4e-2, 0.4e-2, 0.3e-2]
    fg_a = [1.2e-2, 1.5e-2, 1.1e-2, 1.2e-2, 0.7e-2]
    nodes_b = [200.0, 500.0, 700.0]
    ap_b = [2e-7, 5e-7, 9e-7]
    er_b = [1e-7, 3e-7, 5e-7]
    pa = smm.LinearProfile_1D_new("CH4", alt_grid, nodes_a, ap_a, er_a, first_guess_prof=fg_a)
    pb = smm.LinearProfile_1D_new("HCN", alt_grid, nodes_b, ap_b, er_b)
    out["alts"], out["nodes_a"], out["nodes_b"] = alts, np.array(nodes_a), np.array(nodes_b)
    out["ap_a"], out["er_a"], out["fg_a"], out["ap_b"], out["er_b"] = map(np.array, (ap_a, er_a, fg_a, ap_b, er_b))
    out["masks_a"] = np.array([p.maskgrid.mask for p in pa.set])
    out["masks_b"] = np.array([p.maskgrid.mask for p in pb.set])
    out["tri_step"] = smm.alt_triangle(alts, 420.0, step=75.0).mask
    out["involved_a"] = np.array([[pa.check_involved(k, {"alt": (lo, lo + 50.0)}) for lo in (100.0, 320.0, 650.0, 850.0)]
                                  for k in nodes_a], dtype=bool)
    lat_limits = [-90.0, -60.0, -30.0, 30.0, 60.0]
    out["lat_limits"] = np.array(lat_limits)
    out["lat_boxes"] = np.array([smm.lat_box(lat_limits, la).mask for la in (-75.0, -30.0, 10.0, 59.9, 75.0)])
    out["lat_centres"] = np.array(smm.centre_boxes(lat_limits))

    bs = smm.BayesSet(tag="golden")
    bs.add_set(pa)
    bs.add_set(pb)
    n_par, n_pix, n_low = bs.n_tot, 2, 24

    class Sp(object):
        def __init__(self, v):
            self.spectrum = np.array(v, dtype=float)

    ders = rng.standard_normal((n_par, n_pix, n_low)) * 1e-7
    for ip, par in enumerate(bs.params()):
        for num in range(n_pix):
            par.store_deriv(Sp(ders[ip, num]), num)
    masks = [rng.random(n_low) > 0.2 for _ in range(n_pix)]
    out["ders"], out["pix_masks"] = ders, np.array(masks)
    out["jac"] = bs.build_jacobian()
    out["jac_masked"] = bs.build_jacobian(masks=masks)
    out["S_ap"], out["x_ap"], out["x0"] = bs.VCM_apriori(), bs.apriori_vector(), bs.param_vector()
    # positivity: parameters 1 and 6 are pushed below zero and have their step halved until positive
    dx = np.array([1e-3, -4.9e-2, 2e-3, -3e-3, 1e-3, 1e-7, -9e-7, 2e-7])
    bs.update_params(dx)
    out["dx_pos"], out["x_after_pos"] = dx, bs.param_vector()
    out["old_params_0"] = np.array(bs.old_params[0])
    # one Levenberg-Marquardt step with the reference's algebra on this parameter space
    sim = [Sp(np.abs(rng.standard_normal(n_low)) * 1e-6) for _ in range(n_pix)]
    noi = [Sp(np.abs(rng.standard_normal(n_low)) * 1e-8 + 2e-8) for _ in range(n_pix)]
    obs = [Sp(s_.spectrum + 3e-8 * rng.standard_normal(n_low)) for s_ in sim]
    for par in bs.params():
        par.set_used()
    out["sim"], out["noi"], out["obs"] = (np.array([o.spectrum for o in x]) for x in (sim, noi, obs))
    out["chi"] = smm.chicalc(obs, sim, noi, masks, bs.n_used_par())
    smm.inversion_algebra(obs, sim, noi, bs, lambda_LM=0.1, masks=masks)
    bs.update_parerror()
    out["x_after_lm"], out["vcm"], out["avk"] = bs.param_vector(), np.array(bs.VCM), np.array(bs.av_kernel)
    out["ret_error"] = np.array([p.ret_error for p in bs.params()])

    # FOV integration of three LOS spectra (lower, centre, upper) for three pixel rotations
    class Grid(object):
        def __init__(self, g):
            self.grid = g

    class Rad(object):
        def __init__(self, g, v):
            self.spectral_grid, self.spectrum = Grid(g), np.array(v, dtype=float)

    wl = np.linspace(3.2, 3.45, 17)
    spe = np.array([1e-6 * (1.0 + 0.3 * np.sin(7 * wl + ph)) * sc for ph, sc in ((0.0, 0.7), (0.4, 1.0), (0.9, 1.6))])
    out["fov_wl"], out["fov_spe"], out["fov_rot"] = wl, spe, np.array([0.0, 20.0, -45.0])
    out["fov_out"] = np.array([smm.FOV_integr_1D([Rad(wl, v) for v in spe], pixel_rot=r).spectrum
                               for r in out["fov_rot"]])
    # LinearProfile_1D (older constructor, with its unsliced zip) and LinearProfile_2D (alt nodes x lat boxes).
    # AtmGridMask.merge is in the absent module: the stub below merges as the outer product lat x alt.
    class Atmo(object):
        pass
    atmo = Atmo()
    atmo.grid = alt_grid

    def merge(self, other):
        m = AtmGridMask((other.grid, self.grid), np.outer(other.mask, self.mask), {"lat": other.interp, "alt": self.interp})
        return m
    AtmGridMask.merge = merge
    p1 = smm.LinearProfile_1D("CH4", atmo, nodes_a, ap_a, er_a, first_guess_prof=fg_a)
    out["lp1d_masks"] = np.array([p.maskgrid.mask for p in p1.set])
    out["lp1d_apriori"] = np.array([p.apriori for p in p1.set])
    out["lp1d_err"] = np.array([p.apriori_err for p in p1.set])
    out["lp1d_value"] = np.array([p.value for p in p1.set])
    lat_lim2 = [-90.0, -30.0, 30.0]
    aps = [np.array(ap_a) * f for f in (1.0, 1.5, 0.5)]
    ers = [np.array(er_a) * f for f in (1.0, 1.2, 0.8)]
    p2 = smm.LinearProfile_2D("CH4", atmo, nodes_a, lat_lim2, aps, ers, first_guess_profs=[np.array(fg_a)] * 3)
    out["lp2d_lat_limits"] = np.array(lat_lim2)
    out["lp2d_aps"], out["lp2d_ers"] = np.array(aps), np.array(ers)
    out["lp2d_masks"] = np.array([p.maskgrid.mask for p in p2.set])
    out["lp2d_keys"] = np.array([list(p.key) for p in p2.set])
    out["lp2d_apriori"] = np.array([p.apriori for p in p2.set])
    out["lp2d_value"] = np.array([p.value for p in p2.set])
    out["lp2d_involved"] = np.array([[p2.check_involved(p.key, {"alt": (lo, lo + 50.0), "lat": la}) for p in p2.set]
                                     for lo, la in ((100.0, (-80.0, -70.0)), (320.0, (-40.0, -20.0)), (650.0, (40.0, 50.0)),
                                                    (850.0, (-10.0, 10.0)))], dtype=bool)
    np.savez_compressed(os.path.join(HERE, "retrieval_classes.npz"), **out)
    print("retrieval: x_after_pos", out["x_after_pos"][:3], "fov", out["fov_out"][:, 0])


if __name__ == "__main__" and "--retrieval" in sys.argv:
    golden_retrieval()


def _patch_add_lines_py3(spcl, RF):
    """SpectralObject.add_lines_to_spectrum (spect_classes.py:1016-1097) cannot run under Python 3 as it
    lies: it slices with n_lines/n_threads (a float) and forks one process per slice.  The harness
    replaces ONLY that fan-out: the rows are packed by the reference's own prepare_fortran_sum
    (spect_classes.py:1100-1147, one call, a list standing in for the Queue) and summed by the reference's
    compiled sum_all_lines (lineshape.f:2-25) on its fixed (40000, 13010) matrix and 2e6-point spectrum,
    exactly as lines 1079-1095 arrange them.  BuildCoeff / LutSet.add_PT then run unmodified."""
    import ctypes as C
    lib = RF._lib("lineshape")
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    matrix = np.zeros((spcl.imxlines, spcl.imxsig), dtype=float, order="F")   # 4.16 GB, reused
    state = {"rows": 0}

    class Coda(object):
        def put(self, v):
            self.v = v

    def add_lines_to_spectrum(self, lines, Strengths=None, fix_length=spcl.imxsig, n_threads=1):
        n_lines = len(lines)
        if n_lines == 0:
            return self.spectrum
        if n_lines > spcl.imxlines:
            raise ValueError("too many lines")
        lines_ok = []
        if Strengths is not None:
            for line, strength in zip(lines, Strengths):
                lines_ok.append(line.multiply(strength, save=False))          # spcl:1040-1042
        coda = Coda()
        self.prepare_fortran_sum(lines_ok, 0, coda)
        rows, initarr, finarr = coda.v
        matrix[:state["rows"], :] = 0.0
        matrix[:n_lines, :] = rows
        state["rows"] = n_lines
        init = np.zeros(spcl.imxlines, dtype=np.int32)
        fin = np.zeros(spcl.imxlines, dtype=np.int32)
        init[:n_lines], fin[:n_lines] = initarr, finarr
        spe_ini = np.zeros(spcl.imxsig_long)
        spe_ini[:self.n_points()] = self.spectrum
        spe_fin = np.zeros(spcl.imxsig_long)
        lib.sum_all_lines_(spe_ini.ctypes.data_as(dp), matrix.ctypes.data_as(dp), init.ctypes.data_as(ip),
                           fin.ctypes.data_as(ip), C.byref(C.c_int(n_lines)), C.byref(C.c_int(self.n_points())),
                           spe_fin.ctypes.data_as(dp))
        self.spectrum = spe_fin[:self.n_points()]
        return self.spectrum

    spcl.SpectralObject.add_lines_to_spectrum = add_lines_to_spectrum


def golden_gcoeff():
    """A5: per-level, per-ctype G-coefficient spectra from the reference's own LutSet.add_PT
    (spect_main_module.py:1122-1168) -> SpectralGcoeff.BuildCoeff(preCalc_shapes=True)
    (spect_classes.py:1277-1337) on lines prepared by the reference's PrepareCalcShapes, for an
    iso-molecule with levels and for the 'all' set; plus the tracked-level coefficients assembled as
    make_abscoeff_isomolec does (spect_main_module.py:2073-2087) with the reference's operators."""
    import io
    spcl, RF = import_reference_spcl()
    m_mp = types.ModuleType("memory_profiler")
    m_mp.profile = lambda f: f
    sys.modules["memory_profiler"] = m_mp
    import spect_main_module as smm
    from spectrobot_amd import synthetic as syn
    _patch_add_lines_py3(spcl, RF)
    ctypes_ = ["sp_emission", "ind_emission", "absorption"]
    # longer than one 13010-point window: on a shorter grid prepare_fortran_sum shifts `init` below 1
    # and sum_all_lines writes in front of its array (spect_classes.py:1132-1134)
    n_grid = 14000
    grid = syn.make_grid(2992.0, 5e-4, n_grid)
    sg = spcl.SpectralGrid(grid, units="cm_1")
    L = syn.make_lines(90, grid, config_id=102, n_levels=3)
    L["lev_up"][4] = -1                      # dropped by the LinkToMolec filter
    L["lev_lo"][7] = L["lev_up"][7]          # same-level line: dropped too
    L["freq"][0] = grid[0] + 3e-7            # windows clipped at both grid ends
    L["freq"][-1] = grid[-1] - 3e-7
    L["freq"] = np.sort(L["freq"])
    e_lev = np.array([0.0, 1311.0, 3019.0])
    temps = np.array([168.4, 141.2])
    press = np.array([4.0, 3.7e-3])
    tvib = np.array([temps, temps + 14.0, temps + 31.5])
    out = {}
    for tag, levels in (("lev", e_lev), ("all", np.array([]))):
        iso = IsoMolec(6, 1, syn.CH4_MM, levels)
        G = np.zeros((len(temps), max(len(levels), 1), 3, n_grid))
        for k, (P, T) in enumerate(zip(press, temps)):
            lines = ref_lines(spcl, 6, 1, L, labels=len(levels) > 0)
            if len(levels) > 0:
                oks = [l.LinkToMolec(iso) for l in lines]                       # spcl:1384-1388
                lines = [l for l, ok in zip(lines, oks) if ok]
            lines = spcl.PrepareCalcShapes(sg, lines, T, P, iso)
            names = iso.levels if len(levels) > 0 else [None]
            for li, lev in enumerate(names):
                ls = smm.LutSet(6, 1, syn.CH4_MM, level=None if lev is None else getattr(iso, lev), filename="unused")
                ls.temp_file = io.BytesIO()                                      # add_PT pickles into it
                ls.add_PT(sg, lines, P, T, keep_memory=True, n_threads=1)
                for ci, ct in enumerate(ctypes_):
                    G[k, li, ci] = ls.sets[0][ct].spectrum
        out["G_" + tag] = G
    # tracked level 1 (non-LTE populations), spect_main_module.py:2073-2087
    iso = IsoMolec(6, 1, syn.CH4_MM, e_lev)
    trk_abs, trk_emi = [], []
    for k, T in enumerate(temps):
        Q = spcl.CalcPartitionSum(6, 1, temp=T)
        pop = spcl.Boltz_ratio_nodeg(e_lev[1], tvib[1][k]) / Q
        mk = lambda v: spcl.SpectralObject(v.copy(), sg)
        a = spcl.SpectralObject(np.zeros(n_grid), sg)
        e = spcl.SpectralObject(np.zeros(n_grid), sg)
        a += mk(out["G_lev"][k, 1, 2]) * pop
        a -= mk(out["G_lev"][k, 1, 1]) * pop
        e += mk(out["G_lev"][k, 1, 0]) * pop
        trk_abs.append(a.spectrum.copy())
        trk_emi.append(e.spectrum.copy())
    # A9: the reference's LutSet.calculate (spect_main_module.py:997-1066) on a small table of level 1 built
    # with its own add_PT: bilinear inside the pressure range, T only below it.  sbm.weight is in the absent
    # module: linear weights (the only reading of "itype='lin'").
    sys.modules["spect_base_module"].weight = lambda x, x1, x2, itype="lin": (1.0 - (x - x1) / (x2 - x1), (x - x1) / (x2 - x1))
    lut_P, lut_T = [1.0, 4.0], [140.0, 150.0, 160.0]
    lut = smm.LutSet(6, 1, syn.CH4_MM, level=getattr(iso, "lev_01"), filename="unused")
    lut.temp_file = io.BytesIO()
    lut.PTcouples = []
    for P in lut_P:
        for T in lut_T:
            lines = ref_lines(spcl, 6, 1, L)
            oks = [l.LinkToMolec(iso) for l in lines]
            lines = spcl.PrepareCalcShapes(sg, [l for l, ok in zip(lines, oks) if ok], T, P, iso)
            lut.add_PT(sg, lines, P, T, keep_memory=True, n_threads=1)
            lut.PTcouples.append([P, T])
            for ct in ctypes_:
                lut.sets[-1][ct].restore_grid(sg)
    lut_q = [(2.2, 146.0), (0.5, 151.0), (4.0, 157.5)]
    cut = slice(5000, 9000)
    lut_res = np.array([[lut.calculate(P, T)[ct].spectrum[cut] for ct in ctypes_] for P, T in lut_q])
    qpart = np.array([float(spcl.CalcPartitionSum(6, 1, temp=t)) for t in temps])
    np.savez_compressed(os.path.join(HERE, "gcoeff_levels.npz"), lut_P=np.array(lut_P), lut_T=np.array(lut_T),
                        lut_query=np.array(lut_q), lut_cut=np.array([5000, 9000]), lut_result=lut_res,
                        grid_w0=grid[0], grid_step=grid[1] - grid[0],
                        grid_n=n_grid, mm=syn.CH4_MM, mol=6, iso=1, e_lev=e_lev, temps=temps, press=press, tvib=tvib,
                        q_part=qpart, G_lev=out["G_lev"], G_all=out["G_all"], track_level=1,
                        track_abs=np.array(trk_abs), track_emi=np.array(trk_emi),
                        **{"line_" + k: v for k, v in L.items()})
    print("gcoeff_levels: G_lev max per ctype", out["G_lev"].max(axis=(0, 1, 3)), "G_all", out["G_all"].max(axis=(0, 1, 3)))


if __name__ == "__main__" and "--gcoeff" in sys.argv:
    golden_gcoeff()


def golden_outer():
    """Lines 3.3 - 25 cm-1 OUTSIDE the grid (the usual far-wing margin of a HITRAN extraction): the
    reference keeps them (make_abscoeff_isomolec filters on Mol / Iso only, spect_main_module.py:1968),
    closest_grid puts their window on the first / last grid point and humliv_bb runs one of its outer
    branches (lineshape.f:272-442).  Same route as the other end-to-end fixtures (ref_abscoeff)."""
    spcl, RF = import_reference_spcl()
    from spectrobot_amd import synthetic as syn
    n_grid = 15000
    grid = syn.make_grid(2100.0, 5e-4, n_grid)
    L = syn.make_lines(60, grid, config_id=103, n_levels=0, co_like=True)
    far = np.array([-25.0, -11.0, -6.0, -3.4, -3.26, -3.2531, 3.2527, 3.27, 3.6, 7.5, 14.0, 25.0])
    for i, d in enumerate(far):
        L["freq"][i] = (grid[0] + d) if d < 0 else (grid[-1] + d)
        L["a_coeff"][i] *= 30.0
    order = np.argsort(L["freq"])
    L = {k: v[order] for k, v in L.items()}
    iso = IsoMolec(5, 1, syn.CO_MM, [])
    temps = np.array([150.0, 210.0, 120.0])
    press = np.array([1013.25, 40.0, 0.02])       # Lorentz wings reach far at the first two
    ab, em, _ = ref_abscoeff(spcl, grid, ref_lines(spcl, 5, 1, L, labels=False), iso, temps, press, None)
    # the outer lines alone (their contribution is small next to the in-grid lines)
    sel = (L["freq"] < grid[0] - 3.0) | (L["freq"] > grid[-1] + 3.0)
    Lo = {k: v[sel] for k, v in L.items()}
    abo, emo, _ = ref_abscoeff(spcl, grid, ref_lines(spcl, 5, 1, Lo, labels=False), iso, temps, press, None)
    qp = np.array([float(spcl.CalcPartitionSum(5, 1, temp=t)) for t in temps])
    np.savez_compressed(os.path.join(HERE, "e2e_outer_lines.npz"), grid_w0=grid[0], grid_step=grid[1] - grid[0],
                        grid_n=n_grid, mm=syn.CO_MM, mol=5, iso=1, temps=temps, press=press, q_part=qp, abs=ab, emi=em,
                        outer_sel=sel, abs_outer_only=abo, emi_outer_only=emo,
                        **{"line_" + k: v for k, v in L.items()})
    print("e2e_outer_lines: %d outer lines, outer-only abs max %.3e of %.3e" % (sel.sum(), abo.max(), ab.max()))


if __name__ == "__main__" and "--outer" in sys.argv:
    golden_outer()


def golden_containers():
    """A11 remainder + the LUT planner, with the reference's own classes: SpectralGrid / SpectralObject unit
    conversions (spect_classes.py:395-432, 757-807), __getitem__ / __div__ (Python-2 name) / interp_to_grid
    (:451-507), SpectralIntensity.convertto (:1200-1235), Calc_BB (:1881-1892), SpectralGcoeff.interpolate
    (:1349-1375, linear sbm.weight) and calc_PT_couples_atmosphere (spect_main_module.py:1746-1844)."""
    spcl, RF = import_reference_spcl()
    m_mp = types.ModuleType("memory_profiler")
    m_mp.profile = lambda f: f
    sys.modules["memory_profiler"] = m_mp
    sbm = sys.modules["spect_base_module"]
    sbm.weight = lambda x, x1, x2, itype="lin": (1.0 - (x - x1) / (x2 - x1), (x - x1) / (x2 - x1))

    class Molec(object):
        pass

    class IsoMolecS(object):
        def __init__(self, MM):
            self.MM = MM

    sbm.Molec, sbm.IsoMolec = Molec, IsoMolecS
    import spect_main_module as smm
    rng = np.random.default_rng(20260006)
    out = {}
    g0 = np.arange(2000.0, 2003.0, 0.01)
    sp0 = np.abs(rng.standard_normal(len(g0))) + 0.1
    out["grid_cm"], out["spec_cm"] = g0, sp0
    for units in ("nm", "mum", "hz", "cm_1"):
        o = spcl.SpectralObject(sp0.copy(), spcl.SpectralGrid(g0, units="cm_1"), units="cm_1")
        o.convert_grid_to(units)
        out["conv_%s_grid" % units], out["conv_%s_spec" % units] = np.array(o.spectral_grid.grid), np.array(o.spectrum)
        if units != "cm_1":      # and back
            o.convertto_cm_1()
            out["back_%s_grid" % units], out["back_%s_spec" % units] = np.array(o.spectral_grid.grid), np.array(o.spectrum)
    o = spcl.SpectralObject(sp0.copy(), spcl.SpectralGrid(g0, units="cm_1"), units="cm_1")
    sub = o[(2000.5, 2001.25)]
    out["getitem_grid"], out["getitem_spec"] = np.array(sub.spectral_grid.grid), np.array(sub.spectrum)
    out["getitem_none"] = np.array([o[(3000.0, 3001.0)] is None])
    out["div_scalar"] = o.__div__(2.5).spectrum
    out["div_obj"] = o.__div__(spcl.SpectralObject(sp0[::-1].copy(), o.spectral_grid)).spectrum
    ng = spcl.SpectralGrid(np.linspace(1999.5, 2003.5, 97), units="cm_1")
    out["interp_grid"], out["interp_spec"] = ng.grid, o.interp_to_grid(ng).spectrum
    for u in ("Wm2", "nWcm2"):
        si = spcl.SpectralIntensity(sp0.copy(), spcl.SpectralGrid(g0, units="cm_1"), units="ergscm2")
        out["intens_" + u] = np.array(si.convertto(u))
    out["bb_T"] = np.array([150.0, 5777.0])
    out["bb"] = np.array([spcl.Calc_BB(spcl.SpectralGrid(g0, units="cm_1"), T).spectrum for T in out["bb_T"]])
    out["bb_Wm2"] = spcl.Calc_BB(spcl.SpectralGrid(g0, units="cm_1"), 150.0, units="Wm2").spectrum
    sg = spcl.SpectralGrid(g0, units="cm_1")
    a = spcl.SpectralGcoeff("absorption", sg, 6, 1, 16.0, "L01", spectrum=sp0.copy(), Pres=1.0, Temp=150.0)
    b = spcl.SpectralGcoeff("absorption", sg, 6, 1, 16.0, "L01", spectrum=sp0[::-1].copy(), Pres=4.0, Temp=150.0)
    c = spcl.SpectralGcoeff("absorption", sg, 6, 1, 16.0, "L01", spectrum=sp0[::-1].copy(), Pres=1.0, Temp=160.0)
    out["gint_P"] = a.interpolate(b, Pres=2.2).spectrum
    out["gint_T"] = a.interpolate(c, Temp=153.0).spectrum
    # LUT planner on a Titan-like profile
    from spectrobot_amd import synthetic as syn

    class Atm(object):
        pass
    atm = syn.make_atmosphere(80, 0)
    A = Atm()
    A.pres, A.temp = atm["press"], atm["temps"]
    g = syn.make_grid(2990.0, 5e-4, 4000)
    L = syn.make_lines(40, g, config_id=104, n_levels=0)
    lines = ref_lines(spcl, 6, 1, L, labels=False)
    for l in lines:
        l.P_shift = 0.0
    for key, kw in (("a", dict()), ("b", dict(pres_step_log=1.0, temp_step=10.0, max_pres=2.0)),
                    ("c", dict(thres=0.5, add_lowpres=False))):
        pt = smm.calc_PT_couples_atmosphere(lines, IsoMolecS(syn.CH4_MM), A, **kw)
        out["pt_" + key] = np.array(pt)
    out["pt_press"], out["pt_temps"] = atm["press"], atm["temps"]
    np.savez_compressed(os.path.join(HERE, "containers.npz"), **out, **{"line_" + k: v for k, v in L.items()})
    print("containers: PT couples", [len(out["pt_" + k]) for k in "abc"], "bb", out["bb"][:, 0])


if __name__ == "__main__" and "--containers" in sys.argv:
    golden_containers()


def golden_group_obs():
    """The group_observations route of the reference's drivers (What's missing 2):
    make_group_observations (spect_main_module.py:3290-3338: the coarse set of tangent altitudes a pixel set is
    simulated on) and make_radtran_spline (:3377-3396: RectBivariateSpline(kx=2, ky=2) over tangent altitude x spectral
    grid, evaluated at a pixel's three LOS altitudes), run under Python 3.  The absent spect_base_module is stubbed with
    what the two call: Coords / LineOfSight as plain holders; the pixels are plain objects with the VIMSPixel members
    the function reads (limb_tg_alt / lat / lon / sza, LOS() / low_LOS() / up_LOS(), sub_solar_point())."""
    spcl, RF = import_reference_spcl()
    m_mp = types.ModuleType("memory_profiler")
    m_mp.profile = lambda f: f
    sys.modules["memory_profiler"] = m_mp
    sbm = sys.modules["spect_base_module"]

    class Coords(object):
        def __init__(self, c, s_ref="Spherical"):
            self.c = list(c)

        def Spherical(self):
            return self.c

    class LineOfSight(object):
        def __init__(self, start, tg):
            self.starting_point, self.tg = start, tg

        def get_tangent_altitude(self):
            return self.tg.c[2]

        def get_tangent_point(self):
            return self.tg

    sbm.Coords, sbm.LineOfSight = Coords, LineOfSight
    sbm.rad = lambda deg: deg * np.pi / 180.0
    import spect_main_module as smm

    class Pix(object):
        def __init__(self, alt, half, lat, lon, sza):
            self.limb_tg_alt, self.half, self.limb_tg_lat, self.limb_tg_lon, self.limb_tg_sza = alt, half, lat, lon, sza

        def _los(self, alt):
            return LineOfSight("spacecraft", Coords([self.limb_tg_lat, self.limb_tg_lon, alt]))

        def LOS(self):
            return self._los(self.limb_tg_alt)

        def low_LOS(self):
            return self._los(self.limb_tg_alt - self.half)

        def up_LOS(self):
            return self._los(self.limb_tg_alt + self.half)

        def sub_solar_point(self):
            return (0.0, 10.0)

    rng = np.random.default_rng(20260006)
    out = {}
    cases = [dict(alts=[412.0, 187.5, 655.0, 301.0, 533.0], half=12.5, step=50.0, first=None),
             dict(alts=[250.0, 275.0, 300.0], half=20.0, step=30.0, first=200.0),
             dict(alts=[150.0, 480.0], half=7.0, step=75.0, first=400.0)]      # alt_first_los above the range: clamped
    for i, c in enumerate(cases):
        pix = [Pix(a, c["half"], -40.0 + 3.0 * k, 120.0 + k, 55.0 + 2.0 * k) for k, a in enumerate(c["alts"])]
        los, alts, ssps, fszas = smm.make_group_observations(pix, alt_step=c["step"], alt_first_los=c["first"])
        out["go%d_pix_alts" % i] = np.array(c["alts"])
        out["go%d_half" % i], out["go%d_step" % i] = c["half"], c["step"]
        out["go%d_first" % i] = np.nan if c["first"] is None else c["first"]
        out["go%d_alts" % i] = np.array(alts, dtype=float)
        out["go%d_los_alts" % i] = np.array([l.get_tangent_altitude() for l in los])
        out["go%d_sza" % i] = np.array(fszas, dtype=float)
        out["go%d_latlon" % i] = np.array([los[0].tg.c[0], los[0].tg.c[1]])
        out["go%d_sorted" % i] = np.array([p.limb_tg_alt for p in pix])          # (the function sorts the caller's list)
    # make_radtran_spline: 9 simulated altitudes x 40 bands, smooth + noisy columns, evaluated inside and at the nodes
    alts = np.arange(150.0, 551.0, 50.0)
    grid = np.linspace(3.2, 3.6, 40)

    class Rad(object):
        pass
    rads = []
    for a in alts:
        r = Rad()
        r.spectrum = np.exp(-a / 180.0) * (1.0 + 0.3 * np.sin(7.0 * grid)) + 1e-3 * rng.standard_normal(grid.size)
        r.spectral_grid = Rad()
        r.spectral_grid.grid = grid
        rads.append(r)
    f = smm.make_radtran_spline(alts, rads)
    xs = np.array([150.0, 171.3, 200.0, 337.5, 349.9, 512.0, 550.0])
    out["spl_alts"], out["spl_grid"] = alts, grid
    out["spl_spectra"] = np.array([r.spectrum for r in rads])
    out["spl_x"] = xs
    out["spl_values"] = np.array([f(x).spectrum for x in xs])
    np.savez_compressed(os.path.join(HERE, "group_obs.npz"), **out)
    print("group_obs: alts", [list(out["go%d_alts" % i]) for i in range(3)], "spline", out["spl_values"][:, 0])


if __name__ == "__main__" and "--group-obs" in sys.argv:
    golden_group_obs()
