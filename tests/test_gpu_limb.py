"""Device LOS pipeline on the HIP kernels: Curtis-Godson columns per segment, recursion with
the call-site options, Jacobians.  The reference's own radiative-transfer code is in the absent
spect_base_module, so this is the build's definition -- PARITY UNPINNED -- checked against analytic cases
(homogeneous slab, two slabs, pure absorption of a Planck source), the pinned curgod_fort_2, the oracle's
recursion and finite differences.  Needs a real MI355X."""
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


def _atm(n_layers=30):
    from spectrobot_amd import synthetic as syn
    atm = syn.make_atmosphere(n_layers, 1)
    atm["nd"] = syn.number_density(atm["press"], atm["temps"])
    return atm


def test_columns_are_curgod_fort_2(eng, oracle):
    """Per-segment columns of the device pipeline = the reference's curgod_fort_2 (curgods.f:24-45, pinned by
    tests/golden/curgods.npz) on the segment's sample points, times col_scale; observer order too."""
    from spectrobot_amd import synthetic as syn
    from spectrobot_amd.compat import curgods
    atm = _atm()
    vmr = np.array([0.0148 * (1 + 0.2 * np.sin(atm["z"] / 90.0)), 2e-7 * np.exp(atm["z"] / 400.0)])
    L = syn.limb_los(atm["z"], atm["nd"], vmr, [atm["z"][0] + 3.0, atm["z"][9] + 1.0], n_sub=4)
    for order in ("photon", "observer"):
        los = eng.LimbLOS(L["seg_off"], L["seg_layer"], L["pt_off"], L["x"], L["nd"], L["vmr"], col_scale=[0.98827, 0.5],
                          LOS_order=order)
        col = los.columns()
        assert col.shape == (2, len(L["seg_layer"]))
        for g, sc in ((0, 0.98827), (1, 0.5)):
            for s in (0, 5, len(L["seg_layer"]) // 2, len(L["seg_layer"]) - 1):
                a, b = L["pt_off"][s], L["pt_off"][s + 1]
                want = sc * oracle.curgod(2, L["nd"][a:b], L["x"][a:b], vmr=L["vmr"][g][a:b])
                assert abs(col[g, s] - want) < 1e-12 * abs(want)
                shim = sc * curgods.curgod_fort_2(L["nd"][a:b], L["vmr"][g][a:b], L["x"][a:b], b - a)
                assert abs(col[g, s] - shim) < 1e-13 * abs(want)
    # the column of an isothermal exponential atmosphere with constant VMR is n * vmr integrated exactly
    a, b = L["pt_off"][0], L["pt_off"][1]
    assert col[0, 0] > 0 and np.all(col > 0)


def test_analytic_slabs_and_planck_absorption(eng):
    """Homogeneous slab I = S (1 - e^-tau); two slabs I = S1 (1 - e^-tau1) e^-tau2 + S2 (1 - e^-tau2); pure
    absorption of a Planck source I = B(T) e^-tau (radtran_3D_ch4.py:297-315: solo_absorption + initial_intensity)."""
    import torch
    from spectrobot_amd import spect_classes as spcl, synthetic as syn
    rng = np.random.default_rng(3)
    n = 4000
    grid = syn.make_grid(2990.0, 5e-4, n)
    a = np.stack([10.0 ** rng.uniform(-21, -17, n), 10.0 ** rng.uniform(-21, -17, n)])
    e = np.stack([a[0] * rng.uniform(1e-7, 1e-6, n), a[1] * rng.uniform(1e-7, 1e-6, n)])
    ad, ed = torch.tensor(a, device="cuda"), torch.tensor(e, device="cuda")
    # segments of constant density: curgod needs n_{i+1} != n_i, so a 1e-9 gradient; column = n vmr L
    def seg(nd0, length, npt=3):
        x = np.linspace(0.0, length, npt)
        return x, nd0 * np.exp(-1e-9 * np.arange(npt)), np.full(npt, 1e-2)
    x1, n1, v1 = seg(1e15, 4e6)
    x2, n2, v2 = seg(3e14, 7e6)
    # one slab: three segments of layer 0
    los = eng.LimbLOS([0, 3], [0, 0, 0], [0, 3, 6, 9], np.concatenate([x1, x1, x1]), np.concatenate([n1, n1, n1]),
                      np.concatenate([v1, v1, v1]))
    u1 = los.columns()[0]
    assert relerr(u1, np.full(3, 1e15 * 1e-2 * 4e6)) < 1e-8
    tau = a[0] * u1.sum()
    rad = eng.limb_rays((ad, ed), los).cpu().numpy()[0]
    assert relerr(rad, e[0] / a[0] * -np.expm1(-tau)) < 1e-12
    # two slabs
    los2 = eng.LimbLOS([0, 2], [0, 1], [0, 3, 6], np.concatenate([x1, x2]), np.concatenate([n1, n2]), np.concatenate([v1, v2]))
    ua, ub = los2.columns()[0]
    t1, t2 = a[0] * ua, a[1] * ub
    want = e[0] / a[0] * -np.expm1(-t1) * np.exp(-t2) + e[1] / a[1] * -np.expm1(-t2)
    assert relerr(eng.limb_rays((ad, ed), los2).cpu().numpy()[0], want) < 1e-12
    # observer order lists the same path from the other end
    los2r = eng.LimbLOS([0, 2], [1, 0], [0, 3, 6], np.concatenate([x2, x1]), np.concatenate([n2, n1]), np.concatenate([v2, v1]),
                        LOS_order="observer")
    assert np.array_equal(eng.limb_rays((ad, ed), los2r).cpu().numpy(), eng.limb_rays((ad, ed), los2).cpu().numpy())
    # pure absorption of a Planck source, on a shard of the grid
    Tsun = 5777.0
    los3 = eng.LimbLOS([0, 2], [0, 1], [0, 3, 6], np.concatenate([x1, x2]), np.concatenate([n1, n2]), np.concatenate([v1, v2]),
                       solo_absorption=True, initial_temperature=Tsun)
    bb = spcl.Calc_BB(spcl.SpectralGrid(grid, units="cm_1"), Tsun).spectrum
    got = eng.limb_rays((ad, ed), los3, grid=grid).cpu().numpy()[0]
    assert relerr(got, bb * np.exp(-(t1 + t2))) < 1e-12
    lo = 1000
    got_s = eng.limb_rays((ad[:, lo:].contiguous(), ed[:, lo:].contiguous()), los3, grid=grid, g_lo=lo).cpu().numpy()[0]
    assert np.array_equal(got_s, got[lo:])
    # initial intensity handed over in the radiance buffer (the chained vertical path of radtran_3D_ch4.py:305-312)
    r0 = torch.tensor(bb[None, :].copy(), device="cuda")
    los4 = eng.LimbLOS([0, 2], [0, 1], [0, 3, 6], np.concatenate([x1, x2]), np.concatenate([n1, n2]), np.concatenate([v1, v2]),
                       solo_absorption=True)
    assert relerr(eng.limb_rays((ad, ed), los4, rad0=r0).cpu().numpy()[0], got) < 1e-14


def test_pipeline_vs_oracle_recursion_and_gas_mixture(eng, oracle):
    """Synthetic limb geometry, 3 rays: device columns + recursion against the oracle's recursion fed with the same
    columns; two gases against the explicit per-gas optical depths in numpy."""
    import torch
    from spectrobot_amd import synthetic as syn
    rng = np.random.default_rng(11)
    atm = _atm(24)
    n = 3000
    a1, a2 = rng.uniform(0, 3e-18, (24, n)), rng.uniform(0, 2e-17, (24, n))
    a1[3, :40] = 0.0
    e1, e2 = a1 * rng.uniform(1e-8, 1e-7, (24, n)), a2 * rng.uniform(1e-8, 1e-7, (24, n))
    vmr = np.array([np.full(24, 0.0148), 1e-3 * (1 + 0.5 * np.cos(atm["z"] / 120.0))])
    L = syn.limb_los(atm["z"], atm["nd"], vmr, [atm["z"][0] + 2.0, atm["z"][5] + 7.0, atm["z"][17] + 1.0])
    t = lambda v: torch.tensor(v, device="cuda")
    los1 = eng.LimbLOS(L["seg_off"], L["seg_layer"], L["pt_off"], L["x"], L["nd"], L["vmr"][:1], col_scale=[0.98827])
    col = los1.columns()[0]
    rad = eng.limb_rays((t(a1), t(e1)), los1).cpu().numpy()
    for r in range(3):
        s = slice(L["seg_off"][r], L["seg_off"][r + 1])
        assert relerr(rad[r], oracle.radiance_ray(a1, e1, L["seg_layer"][s], col[s])) < 1e-13
    # the host-column entry point gives the same
    rad_h = eng.radiance_rays(t(a1), t(e1), L["seg_off"], L["seg_layer"], col).cpu().numpy()
    assert relerr(rad, rad_h) < 1e-14
    los2 = eng.LimbLOS(L["seg_off"], L["seg_layer"], L["pt_off"], L["x"], L["nd"], L["vmr"], col_scale=[0.98827, 1.0])
    c2 = los2.columns()
    rad2 = eng.limb_rays([(t(a1), t(e1)), (t(a2), t(e2))], los2).cpu().numpy()
    for r in range(3):
        I = np.zeros(n)
        for s in range(L["seg_off"][r], L["seg_off"][r + 1]):
            k = L["seg_layer"][s]
            tau = a1[k] * c2[0, s] + a2[k] * c2[1, s]
            E = e1[k] * c2[0, s] + e2[k] * c2[1, s]
            I = I * np.exp(-tau) + np.where(tau > 1e-12, E * -np.expm1(-tau) / np.where(tau > 0, tau, 1.0), E)
        assert relerr(rad2[r], I) < 1e-12


def test_limb_jacobians_finite_differences(eng):
    """Profile-parameter Jacobian (two gases, parameters of both) and per-layer Jacobian of the device pipeline
    against central finite differences of the pipeline itself."""
    import torch
    from spectrobot_amd import synthetic as syn, spect_main_module as smm
    rng = np.random.default_rng(5)
    nl, n = 20, 700
    atm = _atm(nl)
    z = atm["z"]
    a = [rng.uniform(0, 4e-18, (nl, n)), rng.uniform(0, 3e-17, (nl, n))]
    e = [a[0] * rng.uniform(1e-8, 1e-7, (nl, n)), a[1] * rng.uniform(1e-8, 1e-7, (nl, n))]
    t = lambda v: torch.tensor(np.ascontiguousarray(v), device="cuda")
    coeffs = [(t(a[0]), t(e[0])), (t(a[1]), t(e[1]))]
    nodes = [z[0] + 50.0, z[nl // 2], z[-1] - 40.0]
    x = [np.array([1.5e-2, 1.2e-2, 0.9e-2]), np.array([1.0e-3, 2.0e-3, 0.7e-3])]
    profs = [smm.LinearProfile_1D_new("g%d" % g, z, nodes, x[g], 0.5 * x[g]) for g in range(2)]
    # densities scaled so that segment optical depths are O(1): both terms of the sensitivities matter and
    # nothing is saturated away (at tau ~ 1e3 a difference quotient is exactly 0)
    L = syn.limb_los(z, atm["nd"] * 1e-6, [profs[0].profile(), profs[1].profile()], [z[0] + 5.0, z[6] + 3.0])
    # parameter weights at the LOS sample points: the masks are piecewise linear in altitude, like the VMR
    top = z[-1] + (z[-1] - z[-2])
    W = np.array([np.interp(L["alt"], np.append(z, top), np.append(p.maskgrid.mask, p.maskgrid.mask[-1]))
                  for g in range(2) for p in profs[g].set])
    par_gas = [0, 0, 0, 1, 1, 1]
    xs = np.concatenate(x)

    def los_for(xv):
        vm = np.array([W[:3].T @ xv[:3], W[3:].T @ xv[3:]])
        return eng.LimbLOS(L["seg_off"], L["seg_layer"], L["pt_off"], L["x"], L["nd"], vm, col_scale=[0.98827, 1.0])

    rad, jac = eng.limb_rays_jacobian(coeffs, los_for(xs), par_gas, W)
    assert relerr(rad.cpu().numpy(), eng.limb_rays(coeffs, los_for(xs)).cpu().numpy()) < 1e-14
    for p in range(6):
        # gas 1 is a minor contributor: its derivative is ~1e-5 I / x, so the difference quotient carries
        # rounding noise ~1e-15 I / h next to its O((h/x)^2 tau^2) truncation; h/x = 1e-3 balances the two at ~1e-7
        h = 1e-3 * xs[p]
        xp, xm = xs.copy(), xs.copy()
        xp[p] += h
        xm[p] -= h
        fd = (eng.limb_rays(coeffs, los_for(xp)) - eng.limb_rays(coeffs, los_for(xm))) / (2 * h)
        scale = fd.abs().amax(dim=1, keepdim=True).clamp_min(1e-300)
        assert float(((jac[:, p] - fd).abs() / scale).max()) < 3e-6, p
    # per-layer scalar acting through the coefficients: abs_g[k] -> abs_g[k] + h dabs_g[k]
    da = [rng.uniform(-1, 1, (nl, n)) * a[g] for g in range(2)]
    de = [rng.uniform(-1, 1, (nl, n)) * e[g] for g in range(2)]
    los = los_for(xs)
    jl = eng.limb_rays_layer_jacobian(coeffs, [(t(da[0]), t(de[0])), (t(da[1]), t(de[1]))], los)
    # the top layer contributes ~1e-8 of the radiance: a relative step of 1e-5 there drowns in rounding; it is
    # optically thin (linear response), so a large step is exact enough
    for k, h, tol in ((0, 1e-5, 1e-6), (7, 1e-5, 1e-6), (19, 1e-1, 1e-5)):
        def run(sign):
            aa = [a[g].copy() for g in range(2)]
            ee = [e[g].copy() for g in range(2)]
            for g in range(2):
                aa[g][k] += sign * h * da[g][k]
                ee[g][k] += sign * h * de[g][k]
            return eng.limb_rays([(t(aa[0]), t(ee[0])), (t(aa[1]), t(ee[1]))], los)
        fd = (run(+1) - run(-1)) / (2 * h)
        scale = fd.abs().amax(dim=1, keepdim=True).clamp_min(1e-300)
        assert float(((jl[:, k] - fd).abs() / scale).max()) < tol, k
    assert float(jl[1, :6].abs().max()) == 0.0    # layers below the second ray's tangent height
    # the one-pass kernel (default for > 8 layers) against the forward-sensitivity kernel
    eng.set_jac_layer_mode(1)
    try:
        jf = eng.limb_rays_layer_jacobian(coeffs, [(t(da[0]), t(de[0])), (t(da[1]), t(de[1]))], los)
    finally:
        eng.set_jac_layer_mode(0)
    scale = jf.abs().amax(dim=2, keepdim=True).clamp_min(1e-300)
    assert float(((jl - jf).abs() / scale).max()) < 1e-11


def test_one_pass_jacobians_vs_forward_sensitivity(eng):
    """sr_limb_rays_jacobians_dev (radiances + per-layer + column-parameter Jacobians in one pass per ray, the host's
    store / add / carry plan) against the forward-sensitivity kernels, which share no code with it beyond the
    segment's attenuation: two gases, 24 level parameters with triangular weights (two per segment and gas), both
    LOS orders, solo_absorption, a Planck initial intensity, rays that miss the lower layers (zero-filled rows), and
    broad weights that make a segment touch more than four parameters (the call then falls back)."""
    import torch
    from spectrobot_amd import synthetic as syn
    rng = np.random.default_rng(11)
    nl, n = 24, 900
    atm = _atm(nl)
    z = atm["z"]
    grid = syn.make_grid(2975.0, 5e-4, n)
    a = [rng.uniform(0, 4e-18, (nl, n)), rng.uniform(0, 3e-17, (nl, n))]
    e = [a[0] * rng.uniform(1e-8, 1e-7, (nl, n)), a[1] * rng.uniform(1e-8, 1e-7, (nl, n))]
    t = lambda v: torch.tensor(np.ascontiguousarray(v), device="cuda")
    coeffs = [(t(a[0]), t(e[0])), (t(a[1]), t(e[1]))]
    dco = [(t(rng.uniform(-1, 1, (nl, n)) * a[g]), t(rng.uniform(-1, 1, (nl, n)) * e[g])) for g in range(2)]
    vm = [np.full(nl, 1.2e-2), np.linspace(2e-3, 5e-4, nl)]
    L = syn.limb_los(z, atm["nd"] * 1e-6, vm, [z[0] + 5.0, z[6] + 3.0, z[15] + 1.0])
    top = z[-1] + (z[-1] - z[-2])
    zz = np.append(z, top)
    tri = []
    for k in range(0, nl, 2):                 # 12 level parameters per gas
        m = np.zeros(nl + 1)
        m[k] = 1.0
        if k == nl - 2:
            m[-2:] = 1.0
        tri.append(np.interp(L["alt"], zz, m))
    W = np.array(tri + tri)
    par_gas = np.array([0] * 12 + [1] * 12, np.int32)

    def cmp(x, y, tol=2e-12):
        sc = y.abs().amax(dim=-1, keepdim=True).clamp_min(1e-300)
        return float(((x - y).abs() / sc).max()) < tol

    for opts in (dict(), dict(LOS_order="observer"), dict(solo_absorption=True, initial_temperature=200.0),
                 dict(initial_temperature=180.0)):
        los = eng.LimbLOS(L["seg_off"], L["seg_layer"], L["pt_off"], L["x"], L["nd"], L["vmr"], col_scale=[0.98827, 1.0], **opts)
        g = grid if "initial_temperature" in opts else None
        rad, jl, jp = eng.limb_rays_jacobians(coeffs, los, dcoeffs=dco, par_gas=par_gas, par_w=W, grid=g)
        # each kind alone through the same entry, and the two older entries (which route to the same kernel)
        r2, jl2, _ = eng.limb_rays_jacobians(coeffs, los, dcoeffs=dco, grid=g)
        _, _, jp2 = eng.limb_rays_jacobians(coeffs, los, par_gas=par_gas, par_w=W, grid=g, want_rad=False)
        assert torch.equal(jl, jl2) and torch.equal(jp, jp2) and torch.equal(rad, r2), opts
        assert torch.equal(jl, eng.limb_rays_layer_jacobian(coeffs, dco, los, grid=g))
        assert torch.equal(jp, eng.limb_rays_jacobian(coeffs, los, par_gas, W, grid=g)[1])
        eng.set_jac_layer_mode(1)
        try:
            rf, jpf = eng.limb_rays_jacobian(coeffs, los, par_gas, W, grid=g)
            jlf = eng.limb_rays_layer_jacobian(coeffs, dco, los, grid=g)
            r3, jl3, jp3 = eng.limb_rays_jacobians(coeffs, los, dcoeffs=dco, par_gas=par_gas, par_w=W, grid=g)
        finally:
            eng.set_jac_layer_mode(0)
        assert torch.equal(jl3, jlf) and torch.equal(jp3, jpf)              # mode 1: the forward kernels
        # round 4: the default is the FOLDED one-pass kernel (both segments of a ray in a shell together, every value
        # stored once; the intensity entering a near-side segment from the observed radiance); mode 2 = path order, one
        # ray per thread; mode 3 = path order, two rays per thread sharing the loads: the bits of mode 2
        eng.set_jac_layer_mode(2)
        try:
            r4, jl4, jp4 = eng.limb_rays_jacobians(coeffs, los, dcoeffs=dco, par_gas=par_gas, par_w=W, grid=g)
            eng.set_jac_layer_mode(3)
            r5, jl5, jp5 = eng.limb_rays_jacobians(coeffs, los, dcoeffs=dco, par_gas=par_gas, par_w=W, grid=g)
        finally:
            eng.set_jac_layer_mode(0)
        assert torch.equal(r5, r4) and torch.equal(jl5, jl4) and torch.equal(jp5, jp4), opts
        assert cmp(rad, r4, 1e-14) and cmp(jl, jl4, 1e-13) and cmp(jp, jp4, 1e-13), opts
        assert cmp(rad, rf, 1e-13) and cmp(jl, jlf) and cmp(jp, jpf), opts
        assert cmp(rad, eng.limb_rays(coeffs, los, grid=g), 1e-13)
        assert float(jl[2, :15].abs().max()) == 0.0 and float(jp[2, :7].abs().max()) == 0.0   # rows the third ray never touches
        assert float(jl[0].abs().max()) > 0 and float(jp[0, 3].abs().max()) > 0
    # broad masks: every parameter touches every segment -> more than four entries per segment: falls back
    Wb = np.array([np.interp(L["alt"], zz, np.exp(-0.5 * ((zz - z[k]) / 300.0) ** 2)) for k in range(0, nl, 2)])
    los = eng.LimbLOS(L["seg_off"], L["seg_layer"], L["pt_off"], L["x"], L["nd"], L["vmr"], col_scale=[0.98827, 1.0])
    rb, _, jb = eng.limb_rays_jacobians(coeffs, los, par_gas=np.zeros(12, np.int32), par_w=Wb)
    assert torch.equal(jb, eng.limb_rays_jacobians(coeffs, los, par_gas=np.zeros(12, np.int32), par_w=Wb, want_rad=False)[2])
    eng.set_jac_layer_mode(1)
    try:
        rbf, jbf = eng.limb_rays_jacobian(coeffs, los, np.zeros(12, np.int32), Wb)
    finally:
        eng.set_jac_layer_mode(0)
    assert torch.equal(jb, jbf) and torch.equal(rb, rbf)


def test_few_broad_parameters_folded_vs_forward(eng):
    """sr_limb_rays_jac_dev with up to eight parameters on 1-D limb / slant rays runs the folded recursion in one
    sweep (sr_limb_fold_sens_lds_kernel: forward sensitivities in fold order); the path-order forward-sensitivity kernel
    (mode 1) shares nothing with it but the segment's attenuation.  Broad masks (every parameter acts on every segment), two gases, both LOS
    orders, solo absorption with a Planck background, an opaque case, slant rays (the outward half alone).

    PARITY UNPINNED: the reference's radiance recursion and Jacobians live in the absent
    spect_base_module; this test checks product kernels against OTHER product kernels and finite differences, not
    against the reference."""
    import torch
    from spectrobot_amd import synthetic as syn
    rng = np.random.default_rng(5)
    nl, n = 24, 700
    atm = _atm(nl)
    z = atm["z"]
    grid = syn.make_grid(2975.0, 5e-4, n)
    t = lambda v: torch.tensor(np.ascontiguousarray(v), device="cuda")
    vm = [np.full(nl, 1.2e-2), np.linspace(2e-3, 5e-4, nl)]
    top = z[-1] + (z[-1] - z[-2])
    zz = np.append(z, top)

    def cmp(x, y, tol=2e-12):
        sc = y.abs().amax(dim=-1, keepdim=True).clamp_min(1e-300)
        return float(((x - y).abs() / sc).max()) < tol

    for scale in (1.0, 300.0):                      # 300: line centres with tau of a few hundred per segment
        a = [rng.uniform(0, 4e-18, (nl, n)) * scale, rng.uniform(0, 3e-17, (nl, n)) * scale]
        e = [a[0] * rng.uniform(1e-8, 1e-7, (nl, n)), a[1] * rng.uniform(1e-8, 1e-7, (nl, n))]
        coeffs = [(t(a[0]), t(e[0])), (t(a[1]), t(e[1]))]
        for kind in ("limb", "slant"):
            if kind == "limb":
                L = syn.limb_los(z, atm["nd"] * 1e-6, vm, [z[0] + 5.0, z[6] + 3.0, z[15] + 1.0])
            else:
                L = syn.slant_los(z, atm["nd"] * 1e-6, vm, [0.0, 40.0, 75.0])
            W = np.array([np.interp(L["alt"], zz, np.exp(-0.5 * ((zz - z[k]) / 250.0) ** 2)) for k in (0, 5, 11, 17, 22)] * 1
                         + [np.interp(L["alt"], zz, np.exp(-0.5 * ((zz - z[k]) / 400.0) ** 2)) for k in (3, 14)])
            par_gas = np.array([0, 0, 0, 0, 0, 1, 1], np.int32)
            for opts in (dict(), dict(LOS_order="observer"), dict(solo_absorption=True, initial_temperature=200.0)):
                los = eng.LimbLOS(L["seg_off"], L["seg_layer"], L["pt_off"], L["x"], L["nd"], L["vmr"], col_scale=[0.98827, 1.0], **opts)
                g = grid if "initial_temperature" in opts else None
                rad, jac = eng.limb_rays_jacobian(coeffs, los, par_gas, W, grid=g)
                eng.set_jac_layer_mode(1)
                try:
                    rf, jf = eng.limb_rays_jacobian(coeffs, los, par_gas, W, grid=g)
                finally:
                    eng.set_jac_layer_mode(0)
                assert not torch.equal(jac, jf)        # two kernels
                assert cmp(rad, rf, 1e-13) and cmp(jac, jf), (scale, kind, opts)
                # joint=True: the same values in ONE buffer, radiances' rows first (one instrument-step call per iteration)
                r2, j2, buf = eng.limb_rays_jacobian(coeffs, los, par_gas, W, grid=g, joint=True)
                assert torch.equal(r2, rad) and torch.equal(j2, jac) and buf.shape == (los.n_rays * (1 + len(par_gas)), n)
                assert buf.data_ptr() == r2.data_ptr() and torch.equal(buf[los.n_rays:].view_as(jac), jac)
                assert cmp(rad, eng.limb_rays(coeffs, los, grid=g), 1e-13)
                assert float(jac.abs().max()) > 0


@pytest.mark.parametrize("n_g", [3, 4])
def test_few_broad_parameters_many_shells_three_gases_asymmetric(eng, n_g):
    """The one-sweep kernel beyond what the retrieval case exercises: 150 layers (the ray's records pass through LDS in
    three chunks of 64 shells, the coefficient prefetch restarts at every chunk), three or four gases with eight parameters, a
    point count that is no multiple of the block (threads beyond the grid keep the block's barriers), and VMRs that
    differ between the two halves of the path (a shell's two segments then have different columns: the far pass and the
    near pass of a shell run one after the other) -- against the path-order forward-sensitivity kernel and against
    central differences of the radiance in one parameter.

    PARITY UNPINNED: the reference's radiance recursion and Jacobians live in the absent
    spect_base_module; this test checks product kernels against OTHER product kernels and finite differences, not
    against the reference."""
    import torch
    from spectrobot_amd import synthetic as syn
    rng = np.random.default_rng(11)
    nl, n = 150, 777
    atm = _atm(nl)
    z = atm["z"]
    t = lambda v: torch.tensor(np.ascontiguousarray(v), device="cuda")
    vm = [np.full(nl, 1.2e-2), np.linspace(2e-3, 5e-4, nl), np.linspace(1e-4, 3e-4, nl), np.linspace(4e-5, 1e-5, nl)][:n_g]
    top = z[-1] + (z[-1] - z[-2])
    zz = np.append(z, top)
    a = [rng.uniform(0, s_, (nl, n)) for s_ in (4e-18, 3e-17, 2e-16, 9e-16)[:n_g]]
    e = [a_ * rng.uniform(1e-8, 1e-7, (nl, n)) for a_ in a]
    coeffs = [(t(a_), t(e_)) for a_, e_ in zip(a, e)]
    L = syn.limb_los(z, atm["nd"] * 1e-6, vm, [z[0] + 5.0, z[40] + 3.0, z[100] + 1.0, z[140] + 2.0])
    vmr = np.array(L["vmr"], dtype=float)
    # the second half of every ray's sample points 7 % richer in gas 1: the path is no longer symmetric
    for r in range(len(L["seg_off"]) - 1):
        p0, p1 = L["pt_off"][L["seg_off"][r]], L["pt_off"][L["seg_off"][r + 1]]
        vmr[1, (p0 + p1) // 2:p1] *= 1.07
    W = np.array([np.interp(L["alt"], zz, np.exp(-0.5 * ((zz - z[k]) / (3.0 * (z[1] - z[0]) + 100.0)) ** 2)) for k in (0, 20, 45, 70, 95, 120, 149, 60)])
    par_gas = np.array([0, 0, 1, 1, 1, 2, 2, n_g - 1], np.int32)

    def cmp(x, y, tol):
        sc = y.abs().amax(dim=-1, keepdim=True).clamp_min(1e-300)
        return float(((x - y).abs() / sc).max())

    for opts in (dict(), dict(LOS_order="observer")):
        los = eng.LimbLOS(L["seg_off"], L["seg_layer"], L["pt_off"], L["x"], L["nd"], vmr, col_scale=[0.98827, 1.0, 1.0, 1.0][:n_g], **opts)
        rad, jac = eng.limb_rays_jacobian(coeffs, los, par_gas, W)
        eng.set_jac_layer_mode(1)
        try:
            rf, jf = eng.limb_rays_jacobian(coeffs, los, par_gas, W)
        finally:
            eng.set_jac_layer_mode(0)
        assert not torch.equal(jac, jf)
        assert cmp(rad, rf, 0) < 1e-13 and cmp(jac, jf, 0) < 2e-12, opts
        assert float(jac.abs().max()) > 0 and jac.shape == (los.n_rays, len(par_gas), n)
        # parameter 3 (gas 1) by central differences of the radiance: VMR +- h W_3
        h = 1e-6
        rp, rm = [], []
        for sgn, out in ((1.0, rp), (-1.0, rm)):
            v2 = vmr.copy()
            v2[1] += sgn * h * W[3]
            l2 = eng.LimbLOS(L["seg_off"], L["seg_layer"], L["pt_off"], L["x"], L["nd"], v2, col_scale=[0.98827, 1.0, 1.0, 1.0][:n_g], **opts)
            out.append(eng.limb_rays(coeffs, l2))
        fd = (rp[0] - rm[0]) / (2 * h)
        assert cmp(jac[:, 3, :], fd, 0) < 1e-6, opts


@pytest.mark.parametrize("n_par,order", [(5, "photon"), (5, "observer"), (8, "photon"), (11, "photon"), (40, "photon")])
def test_retrieval_forward_against_its_parts(eng, n_par, order):
    """engine.retrieval_forward (sr_retrieval_forward_dev: parameter vector -> VMRs on the device -> columns -> radiances
    + Jacobians -> instrument bands) against the same steps taken one by one from the host: the VMR of the retrieved
    gas as sum_p x_p w_p on the host, a fresh LimbLOS with it, limb_rays_jacobian, hires_to_lowres.  Observer order (the
    batch re-lists its sample points; the device sums its own rows) and more parameters than the one-sweep kernel
    takes (the path-order forward sensitivities behind the same entry point; 40: the parameter vector travels as kernel
    arguments in two blocks).  Up to 8 parameters the one-sweep kernel integrates the bands itself (sr_set_band_fusion).
    Two gases, only the second retrieved.

    PARITY UNPINNED: the reference's radiance recursion and Jacobians live in the absent
    spect_base_module; this test checks product kernels against OTHER product kernels and finite differences, not
    against the reference."""
    import torch
    from spectrobot_amd import synthetic as syn
    rng = np.random.default_rng(21)
    nl, n = 30, 3000
    atm = _atm(nl)
    z = atm["z"]
    grid = syn.make_grid(2975.0, 5e-4, n)
    t = lambda v: torch.tensor(np.ascontiguousarray(v), device="cuda")
    vm = [np.full(nl, 1.2e-2), np.linspace(2e-3, 5e-4, nl)]
    a = [rng.uniform(0, 4e-18, (nl, n)), rng.uniform(0, 3e-17, (nl, n))]
    e = [a[0] * rng.uniform(1e-8, 1e-7, (nl, n)), a[1] * rng.uniform(1e-8, 1e-7, (nl, n))]
    coeffs = [(t(a[0]), t(e[0])), (t(a[1]), t(e[1]))]
    L = syn.limb_los(z, atm["nd"] * 1e-6, vm, [z[0] + 5.0, z[8] + 3.0, z[20] + 1.0])
    top = z[-1] + (z[-1] - z[-2])
    zz = np.append(z, top)
    nodes = np.linspace(z[0], z[-1], n_par)
    W = np.array([np.interp(L["alt"], zz, np.clip(1.0 - np.abs(zz - c) / (nodes[1] - nodes[0]), 0.0, None)) for c in nodes])
    par_gas = np.full(n_par, 1, np.int32)
    x = np.linspace(2e-3, 5e-4, n_par) * rng.uniform(0.8, 1.2, n_par)
    opts = dict() if order == "photon" else dict(LOS_order="observer")
    bands = np.linspace(1e7 / grid[-1] + 0.05, 1e7 / grid[0] - 0.05, 7)
    widths = np.full(7, 0.08)
    los = eng.LimbLOS(L["seg_off"], L["seg_layer"], L["pt_off"], L["x"], L["nd"], L["vmr"], col_scale=[0.98827, 1.0], **opts)
    out, _ = eng.retrieval_forward(coeffs, los, par_gas, W, x, grid, bands, widths)
    assert out.shape == (los.n_rays, 1 + n_par, 7)
    vmr2 = np.array(L["vmr"], dtype=float)
    vmr2[1] = x @ W
    los2 = eng.LimbLOS(L["seg_off"], L["seg_layer"], L["pt_off"], L["x"], L["nd"], vmr2, col_scale=[0.98827, 1.0], **opts)
    rad, jac = eng.limb_rays_jacobian(coeffs, los2, par_gas, W)
    lo_r = eng.hires_to_lowres(rad, grid, bands, widths)
    lo_j = eng.hires_to_lowres(jac.reshape(-1, n), grid, bands, widths).reshape(los.n_rays, n_par, 7)
    assert np.max(np.abs(out[:, 0] - lo_r)) <= 1e-12 * np.max(np.abs(lo_r))
    assert np.max(np.abs(out[:, 1:] - lo_j)) <= 1e-11 * np.max(np.abs(lo_j)) and np.max(np.abs(lo_j)) > 0
    # the first gas (no parameters) kept its VMRs: the same call again gives the same doubles
    again, _ = eng.retrieval_forward(coeffs, los, par_gas, W, x, grid, bands, widths)
    assert np.array_equal(again, out)


@pytest.mark.parametrize("n,n_par,n_g,n_extra", [(3000, 7, 2, 0), (256, 3, 1, 0), (70001, 8, 2, 0), (5000, 1, 3, 0), (9000, 4, 2, 30)])
def test_band_fusion_equals_the_instrument_step(eng, n, n_par, n_g, n_extra):
    """sr_retrieval_forward_dev with the instrument bands integrated in the recursion kernel's epilogue (default) against
    spectra + sr_hires_to_lowres_shard_dev's kernels (sr_set_band_fusion(0)): the same band integrals up to the order of
    the sums.  Grids that do not fill their last 256-point block, one block only; a band outside the grid (exactly zero
    both ways), a band over the whole grid, narrow bands at both ends, overlapping bands, 37 bands (three tiles of 16 in
    the kernel's MFMA product); with and without the field-of-view integral; the fused call leaves `buf` untouched.  PARITY UNPINNED (product kernels against product
    kernels: see test_retrieval_forward_against_its_parts)."""
    import torch
    from spectrobot_amd import synthetic as syn
    rng = np.random.default_rng(n + n_par)
    nl = 22
    atm = _atm(nl)
    z = atm["z"]
    grid = syn.make_grid(2975.0, 5e-4, n)
    t = lambda v: torch.tensor(np.ascontiguousarray(v), device="cuda")
    vm = [np.full(nl, 1.2e-2), np.linspace(2e-3, 5e-4, nl), np.full(nl, 3e-4)][:n_g]
    a = [rng.uniform(0, 4e-18, (nl, n)) * (10.0 ** (g - 1)) for g in range(n_g)]
    e = [a[g] * rng.uniform(1e-8, 1e-7, (nl, n)) for g in range(n_g)]
    coeffs = [(t(a[g]), t(e[g])) for g in range(n_g)]
    L = syn.limb_los(z, atm["nd"] * 1e-6, vm, [z[0] + 5.0, z[2] + 3.0, z[5] + 1.0, z[9] + 2.0, z[12] + 1.0, z[15] + 4.0])
    zz = np.append(z, z[-1] + (z[-1] - z[-2]))
    nodes = np.linspace(z[0], z[-1], max(n_par, 2))[:n_par]
    W = np.array([np.interp(L["alt"], zz, np.clip(1.0 - np.abs(zz - c) / max(nodes[-1] - nodes[0], 50.0) * max(n_par - 1, 1), 0.0, None) + 0.05)
                  for c in nodes])
    par_gas = (np.arange(n_par) % n_g).astype(np.int32)
    x = rng.uniform(0.5, 1.5, n_par) * np.array([vm[g][0] for g in par_gas])
    lam_lo, lam_hi = 1e7 / grid[-1], 1e7 / grid[0]
    span = lam_hi - lam_lo
    bands = np.array([lam_lo - 50 * span - 1.0, 0.5 * (lam_lo + lam_hi), lam_lo + 0.01 * span, lam_hi - 0.01 * span,
                      lam_lo + 0.4 * span, lam_lo + 0.45 * span, lam_lo + 0.8 * span])
    widths = np.array([0.05 * span, 3.0 * span, 0.004 * span, 0.004 * span, 0.06 * span, 0.06 * span, 0.02 * span])
    if n_extra:
        bands = np.concatenate([bands, lam_lo + span * rng.uniform(0.02, 0.98, n_extra)])
        widths = np.concatenate([widths, span * rng.uniform(0.003, 0.2, n_extra)])
    nb = bands.size
    los = eng.LimbLOS(L["seg_off"], L["seg_layer"], L["pt_off"], L["x"], L["nd"], L["vmr"], col_scale=[0.98827, 1.0, 1.0][:n_g])
    fov = np.array([[0.8, 0.8 ** 3, 2 * 1.3 ** 2, 0.0, 0.0, 0.6, 0.0], [0.5, 0.5 ** 3, 2 * 1.1 ** 2, 0.3, 0.02, 0.7, 1.0]])
    res = {}
    try:
        for on in (1, 0):
            eng.set_band_fusion(on)
            buf = torch.full((los.n_rays * (1 + n_par), n), -7.0, dtype=torch.float64, device="cuda")
            res[on] = (eng.retrieval_forward(coeffs, los, par_gas, W, x, grid, bands, widths, buf=buf)[0],
                       eng.retrieval_forward(coeffs, los, par_gas, W, x, grid, bands, widths, fov=fov, buf=buf)[0])
            assert bool((buf == -7.0).all()) == bool(on), on
    finally:
        eng.set_band_fusion(1)
    for k in (0, 1):
        f, u = res[1][k], res[0][k]
        assert f.shape == u.shape == ((2 if k else los.n_rays), 1 + n_par, nb)
        assert np.all(f[..., 0] == 0.0) and np.all(u[..., 0] == 0.0)          # the band outside the grid
        scale = np.max(np.abs(u), axis=(0, 2), keepdims=True)
        assert np.all(scale[:, :, :] > 0)
        assert np.max(np.abs(f - u) / scale) <= 1e-12, (k, np.max(np.abs(f - u) / scale))
        assert not np.array_equal(f, u) or n <= 256                            # (two routes, not one run twice)


def test_ray_batch_radiances_folded_vs_path_order(eng):
    """limb_rays on a launch of more than 2048 waves (ray batches: BASELINE configs[2]) runs the folded sweep
    (sr_limb_fold_fwd_kernel: a shell's coefficients and attenuation once for the ray's two segments); mode 2 keeps the
    path-order kernel.  Limb rays in both LOS orders, slant rays, an opaque case, a Planck background."""
    import torch
    from spectrobot_amd import synthetic as syn
    rng = np.random.default_rng(9)
    nl, n = 24, 16384
    atm = _atm(nl)
    z = atm["z"]
    grid = syn.make_grid(2975.0, 5e-4, n)
    t = lambda v: torch.tensor(np.ascontiguousarray(v), device="cuda")
    vm = [np.full(nl, 1.2e-2), np.linspace(2e-3, 5e-4, nl)]
    for scale in (1.0, 300.0):
        a = [rng.uniform(0, 4e-18, (nl, n)) * scale, rng.uniform(0, 3e-17, (nl, n)) * scale]
        e = [a[0] * rng.uniform(1e-8, 1e-7, (nl, n)), a[1] * rng.uniform(1e-8, 1e-7, (nl, n))]
        coeffs = [(t(a[0]), t(e[0])), (t(a[1]), t(e[1]))]
        for L in (syn.limb_los(z, atm["nd"] * 1e-6, vm, z[0] + 5.0 + 37.0 * np.arange(9)),
                  syn.slant_los(z, atm["nd"] * 1e-6, vm, np.linspace(0.0, 80.0, 9))):
            for opts in (dict(), dict(LOS_order="observer"), dict(solo_absorption=True, initial_temperature=200.0),
                         dict(initial_temperature=150.0)):
                los = eng.LimbLOS(L["seg_off"], L["seg_layer"], L["pt_off"], L["x"], L["nd"], L["vmr"], col_scale=[0.98827, 1.0], **opts)
                g = grid if "initial_temperature" in opts else None
                rad = eng.limb_rays(coeffs, los, grid=g)
                eng.set_jac_layer_mode(2)
                try:
                    ref = eng.limb_rays(coeffs, los, grid=g)
                finally:
                    eng.set_jac_layer_mode(0)
                if not opts and scale == 1.0:
                    assert not torch.equal(rad, ref)   # two kernels (slant rays walked from the observer have no near side: same operations)
                sc = ref.abs().amax(dim=-1, keepdim=True).clamp_min(1e-300)
                assert float(((rad - ref).abs() / sc).max()) < 1e-13, (scale, opts)
                assert float(((rad - ref).abs() / ref.abs().clamp_min(1e-300)).max()) < 1e-11, (scale, opts)   # pointwise too


def test_per_step_batches_keep_the_path_order_kernel(eng):
    """A 3-D batch lists a coefficient row per LOS step (seg_layer = arange(n_seg_total)): every ray then looks like a
    slant ray through rows of its own, and folded it would walk the steps of ALL rays (n_rays x the traffic, n_rays x
    n_seg_total plan records).  Such batches must take the path-order kernel (sr_last_limb_route == 1) while the same
    rays on shared shells fold (== 2); both give the same radiances."""
    import torch
    from spectrobot_amd import synthetic as syn
    rng = np.random.default_rng(10)
    nl, n = 24, 16384
    atm = _atm(nl)
    z = atm["z"]
    vm = [np.full(nl, 1.2e-2)]
    L = syn.limb_los(z, atm["nd"] * 1e-6, vm, z[0] + 5.0 + 23.0 * np.arange(8))
    a = rng.uniform(0, 4e-18, (nl, n))
    e = a * rng.uniform(1e-8, 1e-7, (nl, n))
    t = lambda v: torch.tensor(np.ascontiguousarray(v), device="cuda")
    los1 = eng.LimbLOS(L["seg_off"], L["seg_layer"], L["pt_off"], L["x"], L["nd"], L["vmr"], col_scale=[0.98827])
    r1 = eng.limb_rays((t(a), t(e)), los1)
    assert eng.last_limb_route() == 2
    steps = np.arange(len(L["seg_layer"]), dtype=np.int32)         # a row per LOS step, as geometry.limb_los_3d lists them
    los3 = eng.LimbLOS(L["seg_off"], steps, L["pt_off"], L["x"], L["nd"], L["vmr"], col_scale=[0.98827])
    r3 = eng.limb_rays((t(a[L["seg_layer"]]), t(e[L["seg_layer"]])), los3)
    assert eng.last_limb_route() == 1
    sc = r1.abs().amax(dim=-1, keepdim=True)
    assert float(((r3 - r1).abs() / sc).max()) < 1e-13
    # two rays of a 3-D batch: the union of their steps is twice the longer ray's -- still folded, empty visits cost no loads
    two = np.array([0, L["seg_off"][1], L["seg_off"][2]], dtype=np.int32)
    k = int(two[-1])
    los2 = eng.LimbLOS(two, steps[:k], L["pt_off"][:k + 1], L["x"][:L["pt_off"][k]], L["nd"][:L["pt_off"][k]],
                       [v[:L["pt_off"][k]] for v in L["vmr"]], col_scale=[0.98827])
    r2 = eng.limb_rays((t(a[L["seg_layer"][:k]]), t(e[L["seg_layer"][:k]])), los2)
    assert float(((r2 - r1[:2]).abs() / sc[:2]).max()) < 1e-13


def test_resident_los_equals_per_call_staging(eng):
    """A LOS batch made resident once (sr_los_create: staged, columns integrated, folded records packed) gives the
    radiances of the per-call route (sr_limb_rays_dev: all of that on every call) bit for bit -- the same kernels on
    the same columns -- for one ray (split kernel), a ray batch (folded sweep), both LOS orders, a Planck background on
    a shard (g_lo per call), two gases and a given initial intensity; and one sr_limb_step_dev call equals the
    coefficient op followed by the recursion."""
    import torch
    from spectrobot_amd import synthetic as syn
    rng = np.random.default_rng(21)
    nl, n = 24, 16384
    atm = _atm(nl)
    z = atm["z"]
    grid = syn.make_grid(2975.0, 5e-4, 2 * n)
    t = lambda v: torch.tensor(np.ascontiguousarray(v), device="cuda")
    vm = [np.full(nl, 1.2e-2), np.linspace(2e-3, 5e-4, nl)]
    a = [rng.uniform(0, 4e-17, (nl, n)), rng.uniform(0, 3e-17, (nl, n))]
    e = [a[0] * rng.uniform(1e-8, 1e-7, (nl, n)), a[1] * rng.uniform(1e-8, 1e-7, (nl, n))]
    two = [(t(a[0]), t(e[0])), (t(a[1]), t(e[1]))]
    for zt in (z[0] + 5.0 + 37.0 * np.arange(9), [z[3] + 2.0]):
        L = syn.limb_los(z, atm["nd"] * 1e-6, vm, zt)
        for opts in (dict(), dict(LOS_order="observer"), dict(solo_absorption=True, initial_temperature=200.0), dict(initial_temperature=150.0)):
            los = eng.LimbLOS(L["seg_off"], L["seg_layer"], L["pt_off"], L["x"], L["nd"], L["vmr"], col_scale=[0.98827, 1.0], **opts)
            g = grid if "initial_temperature" in opts else None
            for g_lo in (0, 5000):
                ref = eng.limb_rays(two, los, grid=g, g_lo=g_lo, resident=False)
                got = [eng.limb_rays(two, los, grid=g, g_lo=g_lo) for _ in range(2)]     # made resident, then reused
                assert torch.equal(got[0], ref) and torch.equal(got[1], ref), (len(zt), opts, g_lo)
            assert len(los._handles) == 1
            los.refresh_columns()                     # the columns integrated again on the device: the same bits
            assert torch.equal(eng.limb_rays(two, los, grid=g, g_lo=5000), ref)
        r0 = t(rng.uniform(1e-9, 1e-8, (len(zt), n)))
        los = eng.LimbLOS(L["seg_off"], L["seg_layer"], L["pt_off"], L["x"], L["nd"], L["vmr"])
        assert torch.equal(eng.limb_rays(two, los, rad0=r0.clone()), eng.limb_rays(two, los, rad0=r0.clone(), resident=False))
    # the whole step in one call
    grid = syn.make_grid(2990.0, 5e-4, 30000)
    Lns = syn.make_lines(3000, grid, seed=4, n_levels=12)
    am = syn.make_atmosphere(16, 12)
    ls = eng.LineSet(Lns, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
    Lr = syn.limb_los(am["z"], syn.number_density(am["press"], am["temps"]), [np.full(16, 0.0148)], [am["z"][0] + 4.0, am["z"][8] + 4.0])
    los = eng.LimbLOS(Lr["seg_off"], Lr["seg_layer"], Lr["pt_off"], Lr["x"], Lr["nd"], Lr["vmr"], col_scale=[syn.CH4_ISO_RATIO])
    T = np.ascontiguousarray(am["temps"])
    for lo, hi in ((0, 30000), (7000, 19000)):
        ab, em = ls.abscoeff_layers(T, am["press"], tvib=am["tvib"], g_lo=lo, g_hi=hi)
        rad = eng.limb_rays((ab, em), los, resident=False)
        for rep in range(2):
            ab2, em2, rad2 = ls.limb_step(T, am["press"], los, tvib=am["tvib"], g_lo=lo, g_hi=hi)
            assert torch.equal(ab2, ab) and torch.equal(em2, em) and torch.equal(rad2, rad), (lo, rep)
    T += 2.0          # values changed in place are seen (the layer descriptor is cached on the arrays, not on their values)
    ab, em = ls.abscoeff_layers(T, am["press"], tvib=am["tvib"])
    ab2, _, rad2 = ls.limb_step(T, am["press"], los, tvib=am["tvib"])
    assert torch.equal(ab2, ab) and torch.equal(rad2, eng.limb_rays((ab, em), los))
    with pytest.raises(RuntimeError):
        ls.limb_step(T[:5], am["press"][:5], los, tvib=am["tvib"][:, :5])     # 5 layers against a LOS through 16


def test_resident_los_with_parameters_and_vmr_updates(eng):
    """The batch of a retrieval: made resident once with its column parameters (sr_los_create_par), new VMRs every
    iteration through LimbLOS.set_vmr (one small copy + the column kernel).  Radiances and parameter Jacobians equal
    the per-call route's (everything staged on every call) bit for bit, before and after VMR updates, for the folded
    kernel (7 broad parameters) and the forward-sensitivity fallback (12 parameters); the radiance handle of the same
    object follows the update too; observer-order batches rebuild their resident form."""
    import torch
    from spectrobot_amd import synthetic as syn
    rng = np.random.default_rng(33)
    nl, n = 24, 8192
    atm = _atm(nl)
    z = atm["z"]
    top = z[-1] + (z[-1] - z[-2])
    zz = np.append(z, top)
    t = lambda v: torch.tensor(np.ascontiguousarray(v), device="cuda")
    a = [rng.uniform(0, 4e-17, (nl, n)), rng.uniform(0, 3e-17, (nl, n))]
    e = [a[0] * rng.uniform(1e-8, 1e-7, (nl, n)), a[1] * rng.uniform(1e-8, 1e-7, (nl, n))]
    two = [(t(a[0]), t(e[0])), (t(a[1]), t(e[1]))]
    vm = [np.full(nl, 1.2e-2), np.linspace(2e-3, 5e-4, nl)]
    L = syn.limb_los(z, atm["nd"] * 1e-6, vm, z[0] + 5.0 + 31.0 * np.arange(6))
    for n_par in (7, 12):
        centres = np.linspace(zz[0], zz[-1], n_par)
        W = np.array([np.interp(L["alt"], zz, np.exp(-0.5 * ((zz - c) / 150.0) ** 2)) for c in centres])
        pg = (np.arange(n_par) % 2).astype(np.int32)
        los = eng.LimbLOS(L["seg_off"], L["seg_layer"], L["pt_off"], L["x"], L["nd"], L["vmr"], col_scale=[0.98827, 1.0])
        for it in range(3):
            if it:
                los.set_vmr(L["vmr"] * (1.0 + 0.3 * it) + 1e-5 * it)
            fresh = eng.LimbLOS(L["seg_off"], L["seg_layer"], L["pt_off"], L["x"], L["nd"], los.vmr, col_scale=[0.98827, 1.0])
            r_ref, j_ref = eng.limb_rays_jacobian(two, fresh, pg, W)
            r, j = eng.limb_rays_jacobian(two, los, pg, W, resident=True)
            assert torch.equal(r, r_ref) and torch.equal(j, j_ref), (n_par, it)
            assert torch.equal(eng.limb_rays(two, los), eng.limb_rays(two, fresh, resident=False)), (n_par, it)
        assert len(los._handles) == 2 and float(j.abs().max()) > 0
    # observer-order batches re-list their sample points: sr_los_set_vmr refuses them, so LimbLOS.set_vmr drops their
    # resident forms BEFORE it changes the host copy (the two never disagree) and the next call rebuilds
    obs = eng.LimbLOS(L["seg_off"], L["seg_layer"], L["pt_off"], L["x"], L["nd"], L["vmr"], LOS_order="observer")
    eng.limb_rays(two, obs)
    assert len(obs._handles) == 1
    obs.set_vmr(L["vmr"] * 2.0)
    assert len(obs._handles) == 0
    fresh = eng.LimbLOS(L["seg_off"], L["seg_layer"], L["pt_off"], L["x"], L["nd"], L["vmr"] * 2.0, LOS_order="observer")
    assert torch.equal(eng.limb_rays(two, obs), eng.limb_rays(two, fresh, resident=False))


def test_per_level_partial_radiances_sum_to_total(eng):
    """single_rad[(gas, iso, lev)] (spect_main_module.py:2883-2887): the radiance emitted by one level and
    absorbed by the whole gas -- the level's emission share (sr_abscoeff_level_dev) with the total
    absorption.  The shares of all levels add up to the radiance."""
    import torch
    from spectrobot_amd import synthetic as syn
    grid = syn.make_grid(2990.0, 5e-4, 8000)
    Lns = syn.make_lines(600, grid, seed=4, n_levels=12)
    atm = syn.make_atmosphere(16, 12)
    nd = syn.number_density(atm["press"], atm["temps"])
    ls = eng.LineSet(Lns, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
    ab, em = ls.abscoeff_layers(atm["temps"], atm["press"] * 30, tvib=atm["tvib"])
    L = syn.limb_los(atm["z"], nd * 30, [np.full(16, 0.0148)], [atm["z"][0] + 4.0, atm["z"][8] + 4.0])
    los = eng.LimbLOS(L["seg_off"], L["seg_layer"], L["pt_off"], L["x"], L["nd"], L["vmr"], col_scale=[syn.CH4_ISO_RATIO])
    total = eng.limb_rays((ab, em), los)
    parts = torch.zeros_like(total)
    for lv in range(12):
        _, em_l = ls.abscoeff_level(atm["temps"], atm["press"] * 30, lv, tvib=atm["tvib"])
        parts += eng.limb_rays((ab, em_l), los)
    assert float(((parts - total).abs() / total.abs()).max()) < 1e-11
    assert float(total.min()) > 0


def test_adaptive_los_stepping_converges(eng):
    """engine.calc_radtran_steps: the reference's radtran_opt knobs (max_T_variation, max_Plog_variation,
    max_opt_depth: radtran_test_CO.py:184-186, spect_main_module.py:2760-2762) on the device LOS pipeline.  A step
    carries the coefficient row of its own mean (P, T); as the bounds tighten the radiance converges (fixed stepping,
    one row per shell at the level values, is the coarsest member of the family) and the optical-depth bound leaves no
    step above it.  (A constant-density test atmosphere is not an option: curgod_fort_2 is singular for n(i+1) = n(i),
    curgods.f:33-41, as in the reference.)"""
    import torch
    from spectrobot_amd import synthetic as syn
    grid = syn.make_grid(2990.0, 5e-4, 12000)
    Ls = syn.make_lines(1500, grid, seed=9, n_levels=12)
    atm = syn.make_atmosphere(16, 12)
    z, T, P, tv = atm["z"], atm["temps"], atm["press"], atm["tvib"]
    ls = eng.LineSet(Ls, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
    gas = [dict(lineset=ls, vmr=np.full(16, 0.0148), iso_ratio=syn.CH4_ISO_RATIO, tvib=tv)]
    zt = [z[1] + 5.0, z[7] + 20.0]
    rads, n_steps = [], []
    for opt in (None, dict(max_T_variation=4.0, max_Plog_variation=0.5), dict(max_T_variation=1.0, max_Plog_variation=0.12),
                dict(max_T_variation=0.25, max_Plog_variation=0.03)):
        S = eng.calc_radtran_steps(gas, z, T, P, zt, radtran_opt=opt)
        rads.append(eng.limb_rays(S["coeffs"], S["los"]))
        n_steps.append(len(S["L"]["seg_layer"]))
    assert n_steps[0] < n_steps[1] < n_steps[2] < n_steps[3]
    err = [float(((r - rads[3]).abs().amax(dim=1) / rads[3].abs().amax(dim=1)).max()) for r in rads[:3]]
    print("steps %s: deviation of the radiance from the finest stepping %s" % (n_steps, ["%.1e" % e for e in err]))
    assert err[2] < err[1] < err[0] and err[2] < 0.02
    # optical depth (largest over the grid, i.e. at the strongest line centre: a thin gas here -- at Titan's CH4 abundance
    # the centres of the lowest shells would need 2^30 halvings, which max_rounds cuts off)
    thin = [dict(gas[0], vmr=np.full(16, 2e-9))]
    S0 = eng.calc_radtran_steps(thin, z, T, P, zt)
    tau0 = S0["coeffs"][0][0].abs().amax(dim=1).cpu().numpy() * S0["los"].columns()[0]
    bound = 0.2 * tau0.max()
    S = eng.calc_radtran_steps(thin, z, T, P, zt, radtran_opt=dict(max_opt_depth=bound))
    col = S["los"].columns()[0]
    tau = S["coeffs"][0][0].abs().amax(dim=1).cpu().numpy() * col
    assert tau.max() <= bound and len(col) > n_steps[0] and tau0.max() > bound
    # halving steps by optical depth refines the same path: the radiance moves towards the finely stepped one
    r0, r1 = eng.limb_rays(S0["coeffs"], S0["los"]), eng.limb_rays(S["coeffs"], S["los"])
    Sf = eng.calc_radtran_steps(thin, z, T, P, zt, radtran_opt=dict(max_T_variation=0.25, max_Plog_variation=0.03))
    rf = eng.limb_rays(Sf["coeffs"], Sf["los"])
    dev = lambda r: float(((r - rf).abs().amax(dim=1) / rf.abs().amax(dim=1)).max())
    assert dev(r1) < dev(r0)
