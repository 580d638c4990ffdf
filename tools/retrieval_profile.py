"""Where a configs[4] retrieval iteration spends its time: cProfile of the 20-iteration loop (host side) next to the
loop's wall clock, and the kernel timeline's share (forward kernel + instrument step by HIP events)."""
import os, sys, time, copy, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench_configs as BC
from spectrobot_amd import engine, retrieval
engine.set_device(0)
scene = BC.two_gas_scene(40000, 8000, 60000, 60)
bs, pixels, x_true = BC.retrieval_problem(scene)
bs0 = copy.deepcopy(bs)
retrieval.inversion_fast_limb(scene, copy.deepcopy(bs0), pixels, max_it=20)      # warm
torch.cuda.synchronize()
gc.collect(); gc.freeze()
for rep in range(3):
    b = copy.deepcopy(bs0)
    t0 = time.perf_counter()
    retrieval.inversion_fast_limb(scene, b, pixels, max_it=20)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("loop %d: %d iterations, %.3f ms per iteration (%.0f it/s)" % (rep, len(b.history), dt / len(b.history) * 1e3, len(b.history) / dt))
import cProfile, pstats
b = copy.deepcopy(bs0)
pr = cProfile.Profile(); pr.enable()
retrieval.inversion_fast_limb(scene, b, pixels, max_it=20)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
