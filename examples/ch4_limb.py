#!/usr/bin/env python3
"""A CH4 Titan limb forward model end to end with the reference's own object types
(the shape of radtran_3D_ch4.py / spect_radtran_test.py, on synthetic inputs):

  lines (SpectLine) + IsoMolec with non-LTE levels
    -> make_abscoeff_isomolec(..., useLUTs=False)      abs / emi coefficient spectra per LOS step
    -> curgods.curgod_batch                            Curtis-Godson absorber columns per segment
    -> engine.radiance_rays / radiance_jacobian        limb radiances (+ d/d VMR-scale parameters)
    -> SpectralIntensity.hires_to_lowres               Gaussian ILS onto VIMS-like bands

Needs an MI355X:  python examples/ch4_limb.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from spectrobot_amd import engine, synthetic as syn                      # noqa: E402
from spectrobot_amd import spect_base_module as sbm                       # noqa: E402
from spectrobot_amd import spect_classes as spcl                          # noqa: E402
from spectrobot_amd import spect_main_module as smm                       # noqa: E402
from spectrobot_amd.compat import curgods                                 # noqa: E402


def main():
    engine.set_device(0)
    n_layers = 40
    grid = syn.make_grid(2990.0, 5e-4, 40000)                 # 20 cm-1 around the CH4 nu3 band
    soa = syn.make_lines(4000, grid, config_id=3, n_levels=12)
    atm = syn.make_atmosphere(n_layers, 12)

    # the reference's objects: one IsoMolec with 12 vibrational levels, SpectLine list
    iso = sbm.IsoMolec(6, 1, syn.CH4_MM, mol_name="CH4")
    for i, e in enumerate(syn.CH4_LEVEL_ENERGIES):
        iso.add_level("L%02d" % i, e, local_vibtemp=atm["tvib"][i])
    lines = [spcl.SpectLine([6, 1, soa["freq"][i], 0.0, soa["a_coeff"][i], soa["air_broad"][i], 0.0,
                             soa["e_lower"][i], soa["t_dep_broad"][i], 0.0, "L%02d" % soa["lev_up"][i],
                             "L%02d" % soa["lev_lo"][i], "", "", "", soa["g_up"][i], soa["g_lo"][i]],
                            nomi=spcl.cose_hit) for i in range(len(soa["freq"]))]

    abs_c, emi_c = smm.make_abscoeff_isomolec([grid[0], grid[-1]], iso, atm["temps"], atm["press"], LTE=False,
                                              lines=lines, to_host=False)
    print("coefficients:", tuple(abs_c.device.shape), "on", abs_c.device.device)

    # three limb rays; columns by curgod_fort_2 over each path segment (n and vmr sampled at 9 points)
    nd = syn.number_density(atm["press"], atm["temps"])
    vmr = np.full(n_layers, 0.0148)
    offs, lays, seg_nd, seg_vmr, seg_x, seg_off = [0], [], [], [], [], [0]
    for zt in (150.0, 300.0, 450.0):
        sl, ln = syn.limb_path(atm["z"], zt)
        for k, length in zip(sl, ln):
            x = np.linspace(0.0, length * 1e5, 9)             # cm
            hscale = 45e5
            seg_nd.append(nd[k] * np.exp(-(x - x.mean()) / hscale * 0.3))
            seg_vmr.append(np.full(9, vmr[k]))
            seg_x.append(x)
            seg_off.append(seg_off[-1] + 9)
        lays += list(sl)
        offs.append(len(lays))
    col = curgods.curgod_batch(2, np.concatenate(seg_nd), np.concatenate(seg_x), seg_off,
                               vmr=np.concatenate(seg_vmr)) * syn.CH4_ISO_RATIO
    # one retrieval parameter per ray-independent altitude band: a scale factor of the VMR
    bands = np.array_split(np.arange(n_layers), 4)
    D = np.zeros((len(lays), len(bands)))
    for p, b in enumerate(bands):
        D[np.isin(lays, b), p] = col[np.isin(lays, b)]       # d col / d scale_p at scale = 1
    rad, jac = engine.radiance_jacobian(abs_c.device, emi_c.device, offs, lays, col, D)
    print("radiance:", tuple(rad.shape), "max %.3e erg s-1 cm-2 sr-1 (cm-1)-1" % float(rad.max()))
    print("jacobian:", tuple(jac.shape))

    class Obs(object):
        pass
    obs = Obs()
    obs.spectral_grid = spcl.SpectralGrid(np.linspace(3325.0, 3342.0, 12), units="nm")
    obs.units = "Wm2"
    hi = spcl.SpectralIntensity(rad[0].cpu().numpy(), spcl.SpectralGrid(grid, units="cm_1"), units="ergscm2")
    low = hi.hires_to_lowres(obs, spectral_widths=[1.2] * 12)
    print("low-res bands [W m-2 sr-1 nm-1]:", np.array2string(low.spectrum, precision=3))


if __name__ == "__main__":
    main()
