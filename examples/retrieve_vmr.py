#!/usr/bin/env python3
"""VMR-profile retrieval loop on synthetic limb observations: the shape of the reference's
inversion driver (spect_main_module.py:2700-2990) with the forward model and its Jacobian on the GPU.

  LinearProfile_1D_new / BayesSet            parameter space: VMR at altitude nodes, triangular masks
  engine.LineSet.abscoeff_layers             abs / emi coefficients of every layer (once: T, P fixed)
  engine.radiance_jacobian                   limb radiances and d rad / d x_p for all rays (per iteration)
  engine.hires_to_lowres                     Gaussian ILS onto VIMS-like bands (radiances and derivatives)
  smm.chicalc / inversion_algebra            optimal estimation, Levenberg-Marquardt step
  smm.retrieval_converged                    the reference's stopping rule

The absorber column of a path segment in layer k is n_k * vmr(z_k) * ds, and vmr(z) = sum_p mask_p(z) x_p,
so d col_s / d x_p = n_k * mask_p(z_k) * ds: exactly the input sr_radiance_jac_dev takes.

Needs an MI355X:  python examples/retrieve_vmr.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from spectrobot_amd import engine, synthetic as syn                      # noqa: E402
from spectrobot_amd import spect_main_module as smm                       # noqa: E402


class _Spectrum(object):
    def __init__(self, v):
        self.spectrum = np.asarray(v, dtype=float)


def run(n_lines=1500, n_grid=16000, n_layers=40, n_iter=12, seed=7, verbose=True):
    rng = np.random.default_rng(seed)
    grid = syn.make_grid(2992.0, 5e-4, n_grid)
    L = syn.make_lines(n_lines, grid, config_id=5, n_levels=12)
    atm = syn.make_atmosphere(n_layers, 12)
    z = atm["z"]
    ls = engine.LineSet(L, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
    abs_c, emi_c = ls.abscoeff_layers(atm["temps"], atm["press"], tvib=atm["tvib"])

    # parameter space: CH4 VMR at five altitude nodes
    nodes = [150.0, 300.0, 450.0, 600.0, 800.0]
    x_true = np.array([1.60e-4, 1.45e-4, 1.30e-4, 1.10e-4, 0.90e-4])   # optically thin: radiance ~ column
    x_ap = np.full(5, 1.2e-4)
    prof = smm.LinearProfile_1D_new("CH4", z, nodes, x_ap, 0.5 * x_ap)
    bayes = smm.BayesSet(tag="CH4 limb")
    bayes.add_set(prof)
    M = prof.mask_matrix()                                   # [n_par, n_layers]

    # limb rays and the per-segment geometry factor n_k * ds * iso_ratio
    nd = syn.number_density(atm["press"], atm["temps"])
    offs, lays, geom = [0], [], []
    for zt in (140.0, 220.0, 300.0, 380.0, 460.0, 540.0):
        sl, ln = syn.limb_path(z, zt)
        lays += list(sl)
        geom += list(ln * 1e5 * nd[sl] * syn.CH4_ISO_RATIO)
        offs.append(len(lays))
    lays, geom = np.array(lays, np.int32), np.array(geom)
    dcol = geom[:, None] * M.T[lays]                          # [n_seg, n_par]
    n_rays = len(offs) - 1

    bands_nm = np.linspace(1e7 / grid[-1] + 0.8, 1e7 / grid[0] - 0.8, 10)
    widths = np.full(bands_nm.size, 0.9)

    def forward(x, want_jac):
        col = dcol @ x
        if not want_jac:
            rad = engine.radiance_rays(abs_c, emi_c, offs, lays, col)
            return engine.hires_to_lowres(rad, grid, bands_nm, widths), None
        rad, jac = engine.radiance_jacobian(abs_c, emi_c, offs, lays, col, dcol)
        low = engine.hires_to_lowres(rad, grid, bands_nm, widths)
        dlow = engine.hires_to_lowres(jac.reshape(n_rays * len(x), -1), grid, bands_nm, widths)
        return low, dlow.reshape(n_rays, len(x), -1)

    y_true, _ = forward(x_true, False)
    sigma = 0.005 * np.abs(y_true).max(axis=1, keepdims=True) * np.ones_like(y_true)
    obs = [_Spectrum(y_true[r] + sigma[r] * rng.standard_normal(y_true.shape[1])) for r in range(n_rays)]
    noise = [_Spectrum(sigma[r]) for r in range(n_rays)]
    for par in bayes.params():
        par.set_used()

    chi_old, history, why = None, [], ""
    for it in range(n_iter):
        low, dlow = forward(bayes.param_vector(), True)
        sims = [_Spectrum(low[r]) for r in range(n_rays)]
        for ip, par in enumerate(bayes.params()):
            for r in range(n_rays):
                par.store_deriv(_Spectrum(dlow[r, ip]), r)
        chi = smm.chicalc(obs, sims, noise, None, bayes.n_used_par())
        history.append(chi)
        if verbose:
            print("iteration %2d  chi2 = %10.3f   x = %s" % (it, chi, np.array2string(bayes.param_vector(), precision=5)))
        why = smm.retrieval_converged(chi, chi_old)
        if why:
            break
        chi_old = chi
        smm.inversion_algebra(obs, sims, noise, bayes, lambda_LM=0.1)
    bayes.update_parerror()
    x_ret = bayes.param_vector()
    err = np.array([p.ret_error for p in bayes.params()])
    if verbose:
        print("stopped:", why or "iteration limit")
        print("true      ", np.array2string(x_true, precision=5))
        print("retrieved ", np.array2string(x_ret, precision=5))
        print("1-sigma   ", np.array2string(err, precision=5))
    return dict(history=history, x_true=x_true, x_ret=x_ret, x_ap=x_ap, err=err, why=why,
                avk_trace=float(np.trace(bayes.av_kernel)))


if __name__ == "__main__":
    engine.set_device(0)
    run()
