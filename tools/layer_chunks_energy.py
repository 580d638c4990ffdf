#!/usr/bin/env python3
"""Does a layer-chunked coefficient op -- the record tables of a chunk small enough to stay in the 256 MB Infinity Cache
between the kernel that writes them and the four that read them -- save Joules (and, the step being power-bound, time)?
sr_set_table_budget makes the op run in layer batches; ms and J per step for several batch sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
import bench as B
from tools.energy_by_kernel import Smi, Sampler
from spectrobot_amd import engine, synthetic as syn, spect_classes as spcl

engine.set_device(0)
grid = syn.make_grid(2975.0, 5e-4, 100000)
L = syn.make_lines(100000, grid, config_id=2, n_levels=12)
atm = syn.make_atmosphere(80, 12)
ls = engine.LineSet(L, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
los, Lr = B.build_rays(syn, engine, atm, 1)
ab = torch.empty((80, 100000), dtype=torch.float64, device="cuda"); em = torch.empty_like(ab)
q = np.atleast_1d(spcl.CalcPartitionSum(6, 1, atm["temps"]))
engine.set_timing(0)


def step():
    los.refresh_columns()
    ls.limb_step(atm["temps"], atm["press"], los, tvib=atm["tvib"], q_part=q, out=(ab, em))


smi = Smi()
sam = Sampler(smi, 0)
sam.start()
per_layer = 100000 * 112 * 2 + 4 * 1024 * 1024     # (two table sets + far-field scratch + zone sums, roughly)
ref = None
for rep in range(2):
    for nl in (80, 40, 20, 10, 5):
        engine.set_table_budget(int(per_layer * nl * 1.02))
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        time.sleep(0.3)
        n = 500
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        w = sam.window(t0 + 0.4, t1 - 0.02)
        cs = float(ab.sum().item())
        ref = cs if ref is None else ref
        print("layers per batch <= %2d: %.3f ms/step  %6.1f W  %4.0f MHz  %.3f J/step   (checksum rel dev %.1e)"
              % (nl, (t1 - t0) / n * 1e3, np.nanmean(w[:, 1]), np.nanmean(w[:, 2]), np.nanmean(w[:, 1]) * (t1 - t0) / n, abs(cs - ref) / abs(ref)), flush=True)
engine.set_table_budget(48 << 30)
sam.stop_flag = True
