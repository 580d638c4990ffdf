#!/usr/bin/env python3
"""The same step issued four ways (two library calls / one, timing events on / off), n_rays rays: ms per step."""
import os, sys, time, gc
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench as B
from spectrobot_amd import engine, synthetic as syn, spect_classes as spcl
engine.set_device(0)
n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 64
grid = syn.make_grid(2975.0, 5e-4, 100000)
L = syn.make_lines(100000, grid, config_id=2, n_levels=12)
atm = syn.make_atmosphere(80, 12)
ls = engine.LineSet(L, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
los, Lr = B.build_rays(syn, engine, atm, n_rays)
ab = torch.empty((80, 100000), dtype=torch.float64, device="cuda"); em = torch.empty_like(ab)
q = np.atleast_1d(spcl.CalcPartitionSum(6, 1, atm["temps"]))
def two(): 
    ls.abscoeff_layers(atm["temps"], atm["press"], tvib=atm["tvib"], q_part=q, out=(ab, em))
    return engine.limb_rays((ab, em), los)
def two_staged():
    ls.abscoeff_layers(atm["temps"], atm["press"], tvib=atm["tvib"], q_part=q, out=(ab, em))
    return engine.limb_rays((ab, em), los, resident=False)
def one():
    return ls.limb_step(atm["temps"], atm["press"], los, tvib=atm["tvib"], q_part=q, out=(ab, em))[2]
gc.collect(); gc.freeze()
for rep in range(2):
    for name, fn, timing in (("two calls, timing on", two, 1), ("two calls, LOS staged per call, timing on", two_staged, 1), ("two calls, timing off", two, 0),
                             ("one call, timing on", one, 1), ("one call, timing off", one, 0)):
        engine.set_timing(timing)
        for _ in range(5): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30): fn()
        torch.cuda.synchronize()
        print("%-45s %.3f ms/step" % (name, (time.perf_counter() - t0) / 30 * 1e3), flush=True)
engine.set_timing(1)
