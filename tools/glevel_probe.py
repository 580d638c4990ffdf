#!/usr/bin/env python3
"""Cost of the level-pair table build on the configs[1] workload (1e5 lines x 1e5 points x 80 rows): per level (sub-lineset
sizes differ 10-fold) in both far-field modes, the whole build, and the folded op beside it."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_configs as bc
from spectrobot_amd import engine, synthetic as syn

engine.set_device(0)
n = int(os.environ.get("N", "100000"))
grid, L, atm, e_lev = bc.ch4_case(n, n, 80)
ls = engine.LineSet(L, grid, 6, 1, syn.CH4_MM, e_lev)
T, P, tv = atm["temps"], atm["press"], atm["tvib"]


def timed(fn, n_rep=5):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_rep):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n_rep * 1e3


ab = torch.empty((80, n), dtype=torch.float64, device="cuda"); em = torch.empty_like(ab)
print("folded op: %.2f ms" % timed(lambda: ls.abscoeff_layers(T, P, tvib=tv, out=(ab, em))))
out = torch.empty((12, 2, 80, n), dtype=torch.float64, device="cuda")
for mode in (2, 1):
    engine.set_far_field(mode)
    print("far-field mode %d: all 12 level pairs %.2f ms" % (mode, timed(lambda: ls.glevel_pairs(T, P, out=out), 3)))
    for lv in range(12):
        nl = int(np.sum((L["lev_up"] == lv) | (L["lev_lo"] == lv)))
        print("   level %2d: %6d lines  %.2f ms" % (lv, nl, timed(lambda: ls.abscoeff_level(T, P, lv, tvib=tv), 3)))
engine.set_far_field(2)
