#!/usr/bin/env python3
"""Where the box-pair far field (mode 2) overtakes the per-line expansions at every level (mode 1): the coefficient op
of n lines on a 1e5-point grid x 80 layers, both modes (the threshold of coef_op's sparse-set switch)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_configs as bc
from spectrobot_amd import engine, synthetic as syn
engine.set_device(0)
n_grid = 100000


def timed(fn, n_rep=6):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_rep):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n_rep * 1e3


for n in (5000, 10000, 20000, 30000, 40000, 50000, 70000, 100000):
    grid, L, atm, e_lev = bc.ch4_case(n, n_grid, 80)
    ls = engine.LineSet(L, grid, 6, 1, syn.CH4_MM, e_lev)
    ab = torch.empty((80, n_grid), dtype=torch.float64, device="cuda"); em = torch.empty_like(ab)
    t = {}
    for mode in (2, 1):
        engine.set_far_field(mode)
        t[mode] = timed(lambda: ls.abscoeff_layers(atm["temps"], atm["press"], tvib=atm["tvib"], out=(ab, em)))
    engine.set_far_field(2)
    print("%6d lines (%.2f per grid point): box pairs %.3f ms, per-line expansions %.3f ms" % (n, n / n_grid, t[2], t[1]), flush=True)
    ls.close()
