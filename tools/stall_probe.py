"""Which host call of a step carries the one-time ~40 ms stall seen ~60 steps into a run (tools/step_trace.py)."""
import os, sys, time, gc
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from spectrobot_amd import engine, synthetic as syn, spect_classes as spcl, distributed as sd
engine.set_device(0)
mode = sys.argv[1] if len(sys.argv) > 1 else "both"
if mode == "nogc":
    gc.disable()
grid = syn.make_grid(2975.0, 5e-4, 100000)
L = syn.make_lines(100000, grid, config_id=2, n_levels=12)
atm = syn.make_atmosphere(80, 12)
ls = engine.LineSet(L, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
los, Lr = B.build_rays(syn, engine, atm, 1)
g_lo, g_hi = sd.shard_bounds(100000, 8, 3)
ab = torch.empty((80, g_hi - g_lo), dtype=torch.float64, device="cuda"); em = torch.empty_like(ab)
q = np.atleast_1d(spcl.CalcPartitionSum(6, 1, atm["temps"]))
rows = []
for i in range(200):
    t0 = time.perf_counter()
    if mode != "limb":
        ls.abscoeff_layers(atm["temps"], atm["press"], tvib=atm["tvib"], q_part=q, g_lo=g_lo, g_hi=g_hi, out=(ab, em))
    t1 = time.perf_counter()
    if mode != "coef":
        engine.limb_rays((ab, em), los)
    t2 = time.perf_counter()
    rows.append((t1 - t0, t2 - t1))
torch.cuda.synchronize()
for i, (a, b) in enumerate(rows):
    if a > 5e-3 or b > 5e-3:
        print("mode %s step %d: abscoeff %.1f ms, limb_rays %.1f ms" % (mode, i, a * 1e3, b * 1e3))
print("mode", mode, "median us", np.median([r[0] for r in rows]) * 1e6, np.median([r[1] for r in rows]) * 1e6)
