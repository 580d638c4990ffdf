#!/usr/bin/env python3
"""Round-2 features end to end on synthetic inputs (needs an MI355X):

  1. LookUpTable.make          per-level G-coefficient tables built on the GPU, resident in HBM
     (spect_main_module.py:718-788: one sr_gcoeff_layers_dev call per level instead of the loop over
     PT couples x levels x ctypes x lines)
  2. make_abscoeff_isomolec    the same LOS steps through the direct route and through the LUT route
     (useLUTs=True: LutSet.calculate + population-weighted combine on the device), with track_levels
  3. engine.LimbLOS            device LOS pipeline: Curtis-Godson columns per segment + recursion, per-level
     partial radiances (single_rad), absorption of a Planck source (solo_absorption + initial_intensity)
  4. retrieval.inversion_fast_limb   HCN + CH4 retrieval loop (bench_configs' scene), reference stopping rule

  python examples/lut_and_retrieval.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from spectrobot_amd import engine, synthetic as syn, retrieval                      # noqa: E402
from spectrobot_amd import spect_base_module as sbm, spect_classes as spcl        # noqa: E402
from spectrobot_amd import spect_main_module as smm                                 # noqa: E402
import bench_configs as bc                                                          # noqa: E402


def main():
    import torch
    engine.set_device(0)
    n_layers = 40
    grid = syn.make_grid(2990.0, 5e-4, 40000)
    soa = syn.make_lines(4000, grid, config_id=3, n_levels=12)
    atm = syn.make_atmosphere(n_layers, 12)
    iso = sbm.IsoMolec(6, 1, syn.CH4_MM, mol_name="CH4")
    for i, e in enumerate(syn.CH4_LEVEL_ENERGIES):
        iso.add_level("L%02d" % i, e, local_vibtemp=atm["tvib"][i])
    lines = [spcl.SpectLine([6, 1, soa["freq"][i], 0.0, soa["a_coeff"][i], soa["air_broad"][i], 0.0, soa["e_lower"][i],
                             soa["t_dep_broad"][i], 0.0, "L%02d" % soa["lev_up"][i], "L%02d" % soa["lev_lo"][i], "", "", "",
                             soa["g_up"][i], soa["g_lo"][i]], nomi=spcl.cose_hit) for i in range(len(soa["freq"]))]
    sg = spcl.SpectralGrid(grid, units="cm_1")

    # 1. the (P, T) couples the atmosphere needs, and the table
    class Atm(object):
        pres, temp = atm["press"], atm["temps"]
    planned = smm.calc_PT_couples_atmosphere(lines, iso, Atm, pres_step_log=1.0, temp_step=5.0)
    print("calc_PT_couples_atmosphere: %d couples for this atmosphere" % len(planned))
    # a rectangular table here: LutSet.calculate takes the two nearest temperatures of the WHOLE table at the two
    # nearest pressures (nearest in P, not log P) and raises when such a couple is not tabulated
    # (spect_main_module.py:985-995) -- as the reference does on a ragged table
    Ps = np.exp(np.arange(np.floor(np.log(atm["press"].min())), np.log(atm["press"].max()) + 1.0, 1.0))
    Ts = np.arange(5.0 * np.floor(atm["temps"].min() / 5.0) - 5.0, atm["temps"].max() + 10.0, 5.0)
    PT = [[float(P), float(T)] for P in Ps for T in Ts]
    t0 = time.time()
    lut = smm.LookUpTable(iso, [grid[0], grid[-1]], LTE=False)
    lut.make(sg, lines, PT)
    torch.cuda.synchronize()
    gb = sum(s.device.numel() for s in lut.sets.values()) * 8 / 1e9
    print("LUT: %d PT couples x %d levels x 3 ctypes x %d points = %.2f GB in HBM, built in %.2f s "
          "(the reference's own estimate for its path: %.0f min)" % (len(PT), len(iso.levels), len(grid), gb, time.time() - t0,
                                                                   lut.CPU_time_estimate(lines, PT)))

    # 2. direct vs LUT route, tracked level
    a_d, e_d, et, at = smm.make_abscoeff_isomolec([grid[0], grid[-1]], iso, atm["temps"], atm["press"], LTE=False, lines=lines,
                                                  track_levels=["lev_08"], to_host=False)
    a_l, e_l = smm.make_abscoeff_isomolec(None, iso, atm["temps"], atm["press"], LTE=False, useLUTs=True,
                                          allLUTs={(iso.mol_name, iso.iso): lut}, to_host=False)
    rel = ((a_l.device - a_d.device).abs().amax(dim=1) / a_d.device.abs().amax(dim=1)).cpu().numpy()
    print("LUT route vs direct route, max |diff| / max per layer: median %.1e, worst %.1e (interpolation error of the table)"
          % (np.median(rel), rel.max()))

    # 3. device LOS pipeline
    nd = syn.number_density(atm["press"], atm["temps"])
    L = syn.limb_los(atm["z"], nd, [np.full(n_layers, 0.0148)], [150.0, 300.0, 450.0])
    los = engine.LimbLOS(L["seg_off"], L["seg_layer"], L["pt_off"], L["x"], L["nd"], L["vmr"], col_scale=[syn.CH4_ISO_RATIO])
    rad = engine.limb_rays((a_d.device, e_d.device), los)
    part = engine.limb_rays((a_d.device, et["lev_08"].device), los)
    print("limb radiances %s; level lev_08 emits %.1f %% of the band-integrated radiance of the lowest ray"
          % (tuple(rad.shape), 100 * float(part[0].sum() / rad[0].sum())))
    occ = engine.LimbLOS(L["seg_off"], L["seg_layer"], L["pt_off"], L["x"], L["nd"], L["vmr"], col_scale=[syn.CH4_ISO_RATIO],
                         solo_absorption=True, initial_temperature=5777.0)
    sun = engine.limb_rays((a_d.device, e_d.device), occ, grid=grid)
    bb = spcl.Calc_BB(sg, 5777.0).spectrum
    print("solar occultation: mean transmission of the three rays", np.round((sun.cpu().numpy() / bb).mean(axis=1), 4))

    # 4. two-gas retrieval
    scene = bc.two_gas_scene(12000, 2500, 24000, 40)
    bs, pixels, x_true = bc.retrieval_problem(scene)
    t0 = time.time()
    chi, obs, sims, bs = retrieval.inversion_fast_limb(scene, bs, pixels, max_it=20)
    bs.update_parerror()
    print("retrieval: %d iterations in %.2f s, chi2 %s -> %s (%s)" % (len(bs.history), time.time() - t0,
                                                                       "%.1f" % bs.history[0], "%.3f" % bs.history[-1], bs.stop))
    for p, xt in zip(bs.params(), x_true):
        print("   %-4s node %6.1f km: %.3e +- %.1e   (truth %.3e, a priori %.3e)" % (p.nameset, p.key, p.value, p.ret_error, xt, p.apriori))


if __name__ == "__main__":
    main()
