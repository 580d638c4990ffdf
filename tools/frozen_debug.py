#!/usr/bin/env python3
"""Where do the far-field modes and the exact mode differ under frozen boundaries (sr_lineset_set_bounds_temps)?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spectrobot_amd import engine as eng, synthetic as syn
eng.set_device(0)
grid = syn.make_grid(2990.0, 5e-4, 30000)
atm = syn.make_atmosphere(6, 12)
T, P, tv = atm["temps"], atm["press"], atm["tvib"]
L = syn.make_lines(150, grid, seed=11, n_levels=12, config_id=2)
ls = eng.LineSet(L, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
ic = np.rint((np.sort(L["freq"]) - grid[0]) / (grid[1] - grid[0])).astype(int)
for dTb in (8.0, 0.0):
    res = {}
    for mode in (0, 3, 2, 1):
        eng.set_far_field(mode)
        ls.set_bounds_temps(T + dTb)
        res[mode] = ls.abscoeff_layers(T, P, tvib=tv)
        ls.set_bounds_temps(None)
    for mode in (3, 2, 1):
        d = ((res[mode][1] - res[0][1]).abs() / res[0][1].abs().clamp_min(1e-300)).cpu().numpy()
        print("dTb %g mode %d: max rel emi dev %.2e" % (dTb, mode, d.max()))
        for k in range(d.shape[0]):
            bad = np.nonzero(d[k] > 1e-9)[0]
            if bad.size:
                near = ic[np.abs(ic[None, :] - bad[:, None]).argmin(axis=1)]
                off = bad - near
                print("  layer %d: %d points > 1e-9, offsets from the nearest line centre: %s ... max dev %.1e" % (
                    k, bad.size, sorted(set(off.tolist()))[:40], d[k].max()))
