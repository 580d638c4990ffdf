"""Box-pair far field against the exact mode with 331 layers (lanes = (box, layer) pairs in sr_m2l_kernel straddle\nboxes when the layer count is not a multiple of 16) on a shard that is not aligned to the box hierarchy."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectrobot_amd import engine as eng, synthetic as syn
grid = syn.make_grid(2990.0, 5e-4, 20000)
L = syn.make_lines(8000, grid, seed=5, n_levels=3)
nl = 331
rng = np.random.default_rng(3)
T = rng.uniform(80, 280, nl); P = 10.0 ** rng.uniform(-6, 2.5, nl); q = rng.uniform(50, 500, nl)
tv = np.array([T + 2.0 * i for i in range(3)])
ls = eng.LineSet(L, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES[:3])
out = {}
for m in (0, 2):
    eng.set_far_field(m)
    a, e = ls.abscoeff_layers(T, P, tvib=tv, q_part=q, g_lo=333, g_hi=19001)
    out[m] = e.cpu().numpy()
eng.set_far_field(eng.FAR_FIELD_DEFAULT)
print("331 layers: max rel diff emi mode 2 vs exact %.2e" % np.max(np.abs(out[2] / out[0] - 1)))
