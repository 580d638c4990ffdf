"""Host time per step (enqueue only, no synchronisation) of the two calls of a bench step on a shard so small
that the GPU is never the limit: the floor that multi-GPU strong scaling runs into (one process per GPU, each
issuing every step)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from spectrobot_amd import engine, synthetic as syn, spect_classes as spcl
engine.set_device(0)
grid = syn.make_grid(2975.0, 5e-4, 100000)
L = syn.make_lines(100000, grid, config_id=2, n_levels=12)
atm = syn.make_atmosphere(80, 12)
ls = engine.LineSet(L, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
los, Lr = B.build_rays(syn, engine, atm, 1)
g_lo, g_hi = 0, 100000 // 64
ab = torch.empty((80, g_hi - g_lo), dtype=torch.float64, device="cuda"); em = torch.empty_like(ab)
q = np.atleast_1d(spcl.CalcPartitionSum(6, 1, atm["temps"]))
RES = True
def step(t):
    t0 = time.perf_counter()
    ls.abscoeff_layers(atm["temps"], atm["press"], tvib=atm["tvib"], q_part=q, g_lo=g_lo, g_hi=g_hi, out=(ab, em))
    t1 = time.perf_counter()
    engine.limb_rays((ab, em), los, resident=RES)
    t2 = time.perf_counter()
    t[0] += t1 - t0; t[1] += t2 - t1
def one_call(t):
    t0 = time.perf_counter()
    los.refresh_columns()      # (bench.py's step: the column integration stays in it, two launches)
    ls.limb_step(atm["temps"], atm["press"], los, tvib=atm["tvib"], q_part=q, g_lo=g_lo, g_hi=g_hi, out=(ab, em))
    t[0] += time.perf_counter() - t0
n = 300
for name, fn, res, timing in (("two calls, LOS staged per call (round 4)", step, False, 1), ("two calls, resident LOS", step, True, 1),
                              ("columns + one call (bench.py's step)", one_call, True, 1), ("columns + one call, no timing events", one_call, True, 0)):
    RES = res
    engine.set_timing(timing)
    for _ in range(20): fn([0, 0])
    torch.cuda.synchronize()
    t = [0.0, 0.0]
    w0 = time.perf_counter()
    for _ in range(n): fn(t)
    w1 = time.perf_counter()
    torch.cuda.synchronize()
    w2 = time.perf_counter()
    print("%-42s host per step: coefficient op (or whole step) %.1f us, limb_rays %.1f us, loop %.1f us; drain after the loop %.1f us/step" % (
        name, t[0] / n * 1e6, t[1] / n * 1e6, (w1 - w0) / n * 1e6, (w2 - w1) / n * 1e6))
engine.set_timing(1)
if len(sys.argv) > 1:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): step([0, 0])
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
