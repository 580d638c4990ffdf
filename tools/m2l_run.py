"""A few coefficient ops on config 2 in far-field mode argv[1] (for rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spectrobot_amd import engine as eng, synthetic as syn
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
grid = syn.make_grid(2975.0, 5e-4, n)
L = syn.make_lines(n, grid, config_id=2)
atm = syn.make_atmosphere(80, 12)
ls = eng.LineSet(L, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
eng.set_far_field(mode); eng.set_overlap(0)
for _ in range(6): ls.abscoeff_layers(atm["temps"], atm["press"], tvib=atm["tvib"])
torch.cuda.synchronize()
