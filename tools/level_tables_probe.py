#!/usr/bin/env python3
"""The level-table build (pair tables / three ctypes) on the configs[1] workload, both routes: ms per build, and -- under
rocprofv3 --kernel-trace --stats -- what its kernels cost.   ROUTE=1|0|both  WHAT=pairs|ctypes  N=100000  REPS=5"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench_configs as bc
from spectrobot_amd import engine, synthetic as syn

engine.set_device(0)
if os.environ.get("OVERLAP") is not None:          # OVERLAP=0: the serial schedule (stand-alone kernel durations in a trace)
    engine.set_overlap(int(os.environ["OVERLAP"]))
n = int(os.environ.get("N", "100000"))
rows = int(os.environ.get("ROWS", "80"))
reps = int(os.environ.get("REPS", "5"))
what = os.environ.get("WHAT", "pairs")
grid, L, atm, e_lev = bc.ch4_case(n, n, rows)
ls = engine.LineSet(L, grid, 6, 1, syn.CH4_MM, e_lev)
T, P, tv = atm["temps"], atm["press"], atm["tvib"]
ab = torch.empty((rows, n), dtype=torch.float64, device="cuda"); em = torch.empty_like(ab)
out = torch.empty((12, 2 if what == "pairs" else 3, rows, n), dtype=torch.float64, device="cuda")


def timed(fn, n_rep):
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.4:    # (two calls from an idle GPU ran at its idle clocks: 12 ms for the 5.7 ms op)
        fn()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_rep):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n_rep * 1e3


t_op = timed(lambda: ls.abscoeff_layers(T, P, tvib=tv, out=(ab, em)), reps)
print("folded op: %.2f ms" % t_op, flush=True)
routes = {"both": (1, 0), "1": (1,), "0": (0,)}[os.environ.get("ROUTE", "both")]
for r in routes:
    engine.set_level_route(r)
    fn = (lambda: ls.glevel_pairs(T, P, out=out)) if what == "pairs" else (lambda: ls.gcoeff_levels(T, P, out=out))
    t = timed(fn, reps)
    print("route %d (%s): all 12 levels %.2f ms = %.2f folded ops" % (r, "multi-channel" if r else "per level", t, t / t_op), flush=True)
engine.set_level_route(1)
