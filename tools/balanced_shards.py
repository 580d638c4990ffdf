"""The 8 work-balanced shards (distributed.shard_bounds_balanced) of a band-head line list -- 70 % of the lines in
15 % of the grid, the list of tests/test_host_cpu.py::test_shard_bounds_balanced_on_skewed_line_density -- timed one
after the other on one GPU beside the 8 equal-width shards: max / mean of the measured step times."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from spectrobot_amd import engine, synthetic as syn, spect_classes as spcl, distributed as sd
engine.set_device(0)
n = 100000
grid = syn.make_grid(2975.0, 5e-4, n)
L = syn.make_lines(100000, grid, config_id=2, n_levels=12)
rng = np.random.default_rng(3)
L["freq"] = np.sort(np.concatenate([rng.uniform(grid[20000], grid[35000], 70000), rng.uniform(grid[0], grid[-1], 30000)]))
atm = syn.make_atmosphere(80, 12)
ls = engine.LineSet(L, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
los, Lr = B.build_rays(syn, engine, atm, 1)
q = np.atleast_1d(spcl.CalcPartitionSum(6, 1, atm["temps"]))
def timed(lo, hi, steps=60):
    ab = torch.empty((80, hi - lo), dtype=torch.float64, device="cuda"); em = torch.empty_like(ab)
    def step():
        ls.abscoeff_layers(atm["temps"], atm["press"], tvib=atm["tvib"], q_part=q, g_lo=lo, g_hi=hi, out=(ab, em))
        return engine.limb_rays((ab, em), los)
    for _ in range(8): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
import gc; gc.collect(); gc.freeze()
for name, bounds in (("equal width", [sd.shard_bounds(n, 8, r) for r in range(8)]), ("work balanced", sd.shard_bounds_balanced(L["freq"], grid, 8))):
    ms = np.array([timed(lo, hi) for lo, hi in bounds])
    print("band-head line list, %s shards: ms/step %s  max/mean %.3f  (bounds %s)" % (
        name, " ".join("%.3f" % v for v in ms), ms.max() / ms.mean(), " ".join("%d" % b[1] for b in bounds)))
