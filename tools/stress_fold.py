#!/usr/bin/env python3
"""Randomised check of the folded recursion kernels (radiances of ray batches, few-parameter Jacobians, one-pass
per-layer + column-parameter Jacobians) against the path-order / forward-sensitivity kernels: random layer counts,
gases, tangent heights (rays grazing the top, rays through the lowest layer, repeated heights), slant rays, opacities
from thin to tau ~ 1e4 per segment, LOS orders, solo absorption, Planck backgrounds, 1..30 parameters.
usage: stress_fold.py [first_seed] [n_seeds]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spectrobot_amd import engine as eng, synthetic as syn
eng.set_device(0)
s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t = lambda v: torch.tensor(np.ascontiguousarray(v), device="cuda")


def dev(x, y, floor=None):
    """Largest difference relative to the row's largest value -- rows whose every value is below 1e-10 of the array's
    largest are rounding noise in ALL the kernels (a parameter hidden behind tau of thousands: path order and forward
    sensitivities disagree there by half the row's maximum themselves) and are measured against that floor.
    floor (same shape): what the adjoint weight w_tau = E f' - I_in t can resolve at that entry, see hidden_floor();
    a difference inside it does not count."""
    sc = y.abs().amax(dim=-1, keepdim=True).clamp_min(1e-10 * float(y.abs().max())).clamp_min(1e-250)
    d = (x - y).abs()
    if floor is not None:
        d = (d - floor).clamp_min(0.0)
    return float((d / sc).max())


def hidden_floor(rad, a_list, los_kw, L, pg, W, eps_factor=64.0):
    """The resolution of the one-pass (adjoint) kernels at a column-parameter entry.  Their weight of a segment is
    w_tau = (E f' - I_in t) x transmission behind it: where a segment sits in radiative equilibrium (I_in t = E f' to
    16 digits: the opaque middle of a thick path) the entry is a cancellation residue, good to eps x |I_in t Tn| x
    |d tau / d x_p| whatever the order of the operations -- path order, folded or forward sensitivities, which differ
    there among themselves.  I_in t Tn <= I_obs (what enters a segment, as seen by the observer, is part of the observed
    radiance), so floor[r, p, j] = eps_factor x 2^-53 x I_obs[r, j] x sum_s abs_g(p)[layer_s, j] x dcol[p][s] -- nothing
    of the folded kernels enters it."""
    n_par = W.shape[0]
    # d col / d x_p per segment: the Curtis-Godson column of the parameter's weight profile (engine.LimbLOS.columns)
    kw = {k: v for k, v in los_kw.items() if k == "LOS_order"}
    cols = []
    for p0 in range(0, n_par, 4):   # (a LOS carries at most four gases)
        Wb = W[p0:p0 + 4]
        cols.append(eng.LimbLOS(L["seg_off"], L["seg_layer"], L["pt_off"], L["x"], L["nd"], Wb, col_scale=[1.0] * len(Wb), **kw).columns())
    dcol = torch.as_tensor(np.concatenate(cols, axis=0), device="cuda").abs()          # [n_par, n_seg]
    scale = torch.as_tensor(np.asarray(los_kw["col_scale"]), device="cuda")
    lay = torch.as_tensor(np.asarray(L["seg_layer"]), device="cuda").long()
    so = L["seg_off"]
    out = torch.zeros((len(so) - 1, n_par) + tuple(rad.shape[-1:]), dtype=torch.float64, device="cuda")
    for r in range(len(so) - 1):
        sl = slice(int(so[r]), int(so[r + 1]))
        for p in range(n_par):
            g = int(pg[p])
            out[r, p] = (a_list[g][lay[sl]] * (dcol[p, sl] * scale[g])[:, None]).sum(dim=0)
    return out * rad[:, None, :].abs() * (eps_factor * 2.0 ** -53)


worst = {"rad": 0.0, "jac": 0.0, "jl": 0.0, "jp": 0.0}
worst_raw = worst_ref = 0.0
for seed in range(s0, s0 + ns):
    rng = np.random.default_rng(seed)
    nl = int(rng.integers(3, 40))
    n_gas = int(rng.integers(1, 4))
    n = int(rng.choice([300, 1000, 20000]))
    atm = syn.make_atmosphere(nl, 1)
    atm["nd"] = syn.number_density(atm["press"], atm["temps"])
    z = atm["z"]
    top = z[-1] + (z[-1] - z[-2])
    zz = np.append(z, top)
    scale = 10.0 ** rng.uniform(-2, 4)
    a = [rng.uniform(0, 4e-18, (nl, n)) * scale * rng.uniform(0.1, 3) for _ in range(n_gas)]
    e = [a[g] * rng.uniform(1e-8, 1e-7, (nl, n)) for g in range(n_gas)]
    coeffs = [(t(a[g]), t(e[g])) for g in range(n_gas)]
    dco = [(t(rng.uniform(-1, 1, (nl, n)) * a[g]), t(rng.uniform(-1, 1, (nl, n)) * e[g])) for g in range(n_gas)]
    vm = [np.exp(rng.uniform(-8, -3)) * np.linspace(1, rng.uniform(0.2, 3), nl) for _ in range(n_gas)]
    n_rays = int(rng.choice([1, 2, 3, 9, 33]))
    kind = rng.choice(["limb", "limb", "slant"])
    if kind == "limb":
        zt = rng.uniform(z[0] + 0.01, top - 0.5, n_rays)
        if n_rays > 2:
            zt[1] = zt[0]                       # a repeated ray
            zt[2] = top - 1e-3                  # grazing the top shell
        L = syn.limb_los(z, atm["nd"] * 1e-6, vm, zt)
    else:
        L = syn.slant_los(z, atm["nd"] * 1e-6, vm, rng.uniform(0, 85, n_rays))
    opts = [dict(), dict(LOS_order="observer"), dict(solo_absorption=True, initial_temperature=200.0), dict(initial_temperature=150.0)][int(rng.integers(0, 4))]
    grid = syn.make_grid(2975.0, 5e-4, n)
    rng_scale = rng.uniform(0.9, 1.0, n_gas)
    los = eng.LimbLOS(L["seg_off"], L["seg_layer"], L["pt_off"], L["x"], L["nd"], L["vmr"], col_scale=list(rng_scale), **opts)
    g = grid if "initial_temperature" in opts else None
    n_par = int(rng.choice([1, 3, 7, 8, 12, 30]))
    broad = n_par <= 8 and rng.random() < 0.5
    width = 400.0 if broad else (zz[-1] - zz[0]) / max(n_par, 2) * 0.4
    centres = rng.uniform(zz[0], zz[-1], n_par)
    if broad:
        W = np.array([np.interp(L["alt"], zz, np.exp(-0.5 * ((zz - c) / width) ** 2)) for c in centres])
    else:
        W = np.array([np.interp(L["alt"], zz, np.clip(1 - np.abs(zz - c) / width, 0, None)) for c in centres])
    pg = rng.integers(0, n_gas, n_par).astype(np.int32)
    out = {}
    for mode in (0, 2, 1):
        eng.set_jac_layer_mode(mode)
        try:
            r_ = eng.limb_rays(coeffs, los, grid=g)
            rj, jj = eng.limb_rays_jacobian(coeffs, los, pg, W, grid=g)
            r3, jl, jp = eng.limb_rays_jacobians(coeffs, los, dcoeffs=dco, par_gas=pg, par_w=W, grid=g)
        finally:
            eng.set_jac_layer_mode(0)
        out[mode] = (r_, rj, jj, r3, jl, jp)
    cs_ = list(rng_scale)
    fl = hidden_floor(out[2][0], [c[0] for c in coeffs], dict(opts, col_scale=cs_), L, pg, W)
    d = {"rad": max(dev(out[0][0], out[2][0]), dev(out[0][1], out[2][1]), dev(out[0][3], out[2][3])),
         "jac": max(dev(out[0][2], out[2][2], fl), dev(out[0][2], out[1][2], fl)),
         "jl": max(dev(out[0][4], out[2][4]), dev(out[0][4], out[1][4])),
         "jp": max(dev(out[0][5], out[2][5], fl), dev(out[0][5], out[1][5], fl))}
    raw = {"jac": max(dev(out[0][2], out[2][2]), dev(out[0][2], out[1][2])), "jp": max(dev(out[0][5], out[2][5]), dev(out[0][5], out[1][5]))}
    worst_raw = max(worst_raw, raw["jac"], raw["jp"])
    # a difference counts when it exceeds 1e-11 of the row's largest value AFTER the resolution floor of the adjoint
    # weights (hidden_floor) has been taken off; the two references against each other, same rule, for the record
    ref = {"rad": 0.0, "jac": dev(out[2][2], out[1][2], fl), "jl": dev(out[2][4], out[1][4]), "jp": dev(out[2][5], out[1][5], fl)}
    worst_ref = max(worst_ref, max(ref.values()))
    bad = not all(np.isfinite(d[k]) and d[k] < 1e-11 for k in d)
    for k in d:
        worst[k] = max(worst[k], d[k])
    if bad or seed % 25 == 0:
        print("seed %d: nl %d gases %d n %d rays %d %s %s n_par %d%s scale %.1e -> %s%s" % (
            seed, nl, n_gas, n, n_rays, kind, opts, n_par, " broad" if broad else "", scale,
            " ".join("%s %.1e" % kv for kv in d.items()), "   <-- BAD" if bad else ""), flush=True)
print("seeds %d..%d: worst deviation of the folded kernels from the path-order / forward-sensitivity ones, of a row's largest value, "
      "beyond the resolution floor of the adjoint weights (64 eps x I_obs x |d tau / d x_p|, tools/stress_fold.py hidden_floor): %s; "
      "without the floor (column-parameter Jacobians) %.2e; the two references against each other, same rule: %.2e"
      % (s0, s0 + ns - 1, " ".join("%s %.2e" % kv for kv in worst.items()), worst_raw, worst_ref))
