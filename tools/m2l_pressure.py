"""Far-field modes on config 2 with the pressures scaled by argv[1] (default 100: an Earth-like 1000 hPa at the
bottom; Lorentz widths of 100+ grid points, the regime where the multipole series of a 64-point source box no longer
converges and the box pairs start at wider levels): accuracy on 8 layers against the exact mode, time per call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spectrobot_amd import engine as eng, synthetic as syn

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
n = 100000
grid = syn.make_grid(2975.0, 5e-4, n)
L = syn.make_lines(n, grid, config_id=2)
atm = syn.make_atmosphere(80, 12)
P_all = atm["press"] * scale
ls = eng.LineSet(L, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
sel = np.arange(0, 80, 10)
T, P, tv = atm["temps"][sel], P_all[sel], atm["tvib"][:, sel]
out = {}
for mode in (0, 1, 2):
    eng.set_far_field(mode)
    a, e = ls.abscoeff_layers(T, P, tvib=tv)
    out[mode] = e.clone()
for m in (1, 2):
    re = ((out[m] - out[0]).abs() / out[0].abs()).amax(dim=1).cpu().numpy()
    print("pressure x%g, mode %d vs exact, max rel err per layer (emi):" % (scale, m), " ".join(f"{x:.1e}" for x in re))
for mode in (1, 2):
    eng.set_far_field(mode)
    eng.set_overlap(0)
    for _ in range(2): ls.abscoeff_layers(atm["temps"], P_all, tvib=atm["tvib"])
    ms = np.zeros(5)
    for _ in range(3):
        ls.abscoeff_layers(atm["temps"], P_all, tvib=atm["tvib"]); torch.cuda.synchronize(); ms += np.array(ls.last_kernel_ms()) / 3
    print("pressure x%g, mode %d: prep %.3f far %.3f wings %.3f zones %.3f ms" % (scale, mode, ms[0], ms[1], ms[2], ms[3]))
eng.set_far_field(eng.FAR_FIELD_DEFAULT); eng.set_overlap(1)
