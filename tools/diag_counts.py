#!/usr/bin/env python3
"""Executed-work counters of one coefficient op on config 2 for the library named by SPECTROBOT_HIP_LIB (diagnostic
variants, e.g. -DSR_DIAG_WINGS: rounds and executed steps of the wings kernel's row walk in the window-end /
polynomial counters)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spectrobot_amd import engine, synthetic as syn, spect_classes as spcl
engine.set_device(0)
grid = syn.make_grid(2975.0, 5e-4, 100000)
L = syn.make_lines(100000, grid, config_id=2, n_levels=12)
atm = syn.make_atmosphere(80, 12)
ls = engine.LineSet(L, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
ab = torch.empty((80, 100000), dtype=torch.float64, device="cuda"); em = torch.empty_like(ab)
q = np.atleast_1d(spcl.CalcPartitionSum(6, 1, atm["temps"]))
engine.set_overlap(0)
engine.set_counting(1)
ls.abscoeff_layers(atm["temps"], atm["press"], tvib=atm["tvib"], q_part=q, out=(ab, em))
torch.cuda.synchronize()
for k, v in ls.last_eval_counts().items():
    print("%-26s %.4e" % (k, v))
