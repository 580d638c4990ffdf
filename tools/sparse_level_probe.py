#!/usr/bin/env python3
"""Per-kernel cost of ONE sparse level pass (level 5: 9 % of the lines) of the pair-table build, kernels one after the
other -- run under rocprofv3 --kernel-trace --stats.  N = grid = lines (default 1e5), 80 rows."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_configs as bc
from spectrobot_amd import engine, synthetic as syn
engine.set_device(0)
n = int(os.environ.get("N", "100000"))
lev = int(os.environ.get("LEVEL", "5"))
grid, L, atm, e_lev = bc.ch4_case(n, n, 80)
ls = engine.LineSet(L, grid, 6, 1, syn.CH4_MM, e_lev)
engine.set_overlap(0)
for _ in range(12):
    ls.abscoeff_level(atm["temps"], atm["press"], lev, tvib=atm["tvib"])
torch.cuda.synchronize()
