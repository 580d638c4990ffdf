"""cProfile of the configs[4] retrieval loop (bench_configs.py --config 4): where the host time of an iteration goes."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_configs as bc
from spectrobot_amd import engine, retrieval
engine.set_device(0)
scene = bc.two_gas_scene(40000, 8000, 60000, 60)
bs, pixels, x_true = bc.retrieval_problem(scene)
retrieval.simulate(scene, pixels, bs)       # warm: coefficients cached, buffers allocated
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
chi, obs, sims, bs = retrieval.inversion_fast_limb(scene, bs, pixels, max_it=20)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
pr.disable()
print("iterations %d, %.2f ms each" % (len(bs.history), dt / len(bs.history) * 1e3))
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
