"""Soak of the device LOS pipeline: thousands of back-to-back radiance / Jacobian calls over changing ray batches
(folded, few-parameter one-sweep, one-pass and path-order kernels, the one-call forward model of a retrieval iteration,
level-factored tables rebuilt in place); the device memory
in use must not grow and every 200th result must equal the first of its shape."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spectrobot_amd import engine as eng, synthetic as syn


def main(n_calls=3000):
    eng.set_device(0)
    rng = np.random.default_rng(1)
    n = 20000
    grid = syn.make_grid(2980.0, 5e-4, n)
    L = syn.make_lines(6000, grid, seed=3, n_levels=12)
    ls = eng.LineSet(L, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
    shapes = []
    for q in range(6):
        nl = int(rng.integers(8, 40))
        atm = syn.make_atmosphere(nl, 12)
        atm["nd"] = syn.number_density(atm["press"], atm["temps"])
        z = atm["z"]
        co = ls.abscoeff_layers(atm["temps"], atm["press"], tvib=atm["tvib"])
        dco = (co[0] * 0.01, co[1] * 0.01)
        vm = [np.full(nl, 0.0148)]
        nr = int(rng.choice([1, 3, 16, 40]))
        Lr = syn.limb_los(z, atm["nd"] * 1e-6, vm, rng.uniform(z[0] + 1, z[-1], nr)) if q % 3 else syn.slant_los(z, atm["nd"] * 1e-6, vm, rng.uniform(0, 80, nr))
        los = eng.LimbLOS(Lr["seg_off"], Lr["seg_layer"], Lr["pt_off"], Lr["x"], Lr["nd"], Lr["vmr"], col_scale=[syn.CH4_ISO_RATIO],
                          **([dict(), dict(LOS_order="observer")][q % 2]))
        zz = np.append(z, z[-1] + (z[-1] - z[-2]))
        npar = int(rng.choice([3, 7, 20]))
        W = np.array([np.interp(Lr["alt"], zz, np.clip(1 - np.abs(zz - c) / 120.0, 0, None)) for c in rng.uniform(zz[0], zz[-1], npar)])
        shapes.append((co, dco, los, np.zeros(npar, np.int32), W, atm))
    lf = eng.LevelFactored(ls, shapes[0][5]["temps"], shapes[0][5]["press"], dT=0.02)
    bands = np.linspace(1e7 / grid[-1] + 0.3, 1e7 / grid[0] - 0.3, 9)
    first, free0, t0 = {}, None, time.time()
    for c in range(n_calls):
        i = int(rng.integers(0, len(shapes)))
        co, dco, los, pg, W, atm = shapes[i]
        kind = c % 5
        if kind == 4:   # a retrieval iteration's forward model in one call, on the batch resident with its parameters
            x = 1.0 + 0.1 * np.arange(len(pg))
            out = torch.from_numpy(eng.retrieval_forward(co, los, pg, W, x, grid, bands, np.full(bands.size, 0.4))[0]).cuda()
        elif kind == 0:
            out = eng.limb_rays(co, los)
        elif kind == 1:
            out = eng.limb_rays_jacobian(co, los, pg, W)[1]
        elif kind == 2:
            out = eng.limb_rays_jacobians(co, los, dcoeffs=dco, par_gas=pg, par_w=W)[1]
        else:
            if c % 40 == 3:
                lf.rebuild(shapes[0][5]["temps"], shapes[0][5]["press"])
            out = lf.steps(np.arange(len(shapes[0][5]["temps"]), dtype=np.int32), tvib=shapes[0][5]["tvib"], derivative=True)[1][0]
            i = 0
        key = (i, kind)
        if key not in first:
            first[key] = out.clone()
        elif c % 200 < 5:
            if kind == 3:
                # the level tables come from the multi-channel pass (round 6): several waves add into one LDS image, the
                # tables are reproducible to ~1e-15 of a spectrum's largest value, their difference quotient over 0.02 K
                # to ~1e-12 of the derivative's -- not bit for bit
                s_ = first[key].abs().amax(dim=-1, keepdim=True).clamp_min(1e-300)
                assert float(((out - first[key]).abs() / s_).max()) < 1e-9, (c, key)
            else:
                assert torch.equal(out, first[key]), (c, key)
        if c == 400:
            torch.cuda.synchronize()
            free0 = torch.cuda.mem_get_info()[0]
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    print("%d calls in %.1f s; device memory free after 400 calls %.3f GiB, at the end %.3f GiB" % (n_calls, time.time() - t0, free0 / 2**30, free1 / 2**30))
    assert free1 >= free0 - (64 << 20), "device memory in use grew"
    print("soak OK")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 3000)
