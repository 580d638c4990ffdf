"""Per-step device completion times of the bench step on a shard (events after every step): where a run's time goes
when ms/step of a short run exceeds the steady state (clock ramp, first-use allocations, ...)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from spectrobot_amd import engine, synthetic as syn, spect_classes as spcl, distributed as sd
engine.set_device(0)
shard = sys.argv[1] if len(sys.argv) > 1 else "3/8"
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 5
grid = syn.make_grid(2975.0, 5e-4, 100000)
L = syn.make_lines(100000, grid, config_id=2, n_levels=12)
atm = syn.make_atmosphere(80, 12)
ls = engine.LineSet(L, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
los, Lr = B.build_rays(syn, engine, atm, 1)
r, w = (int(v) for v in shard.split("/"))
g_lo, g_hi = sd.shard_bounds(100000, w, r)
ab = torch.empty((80, g_hi - g_lo), dtype=torch.float64, device="cuda"); em = torch.empty_like(ab)
q = np.atleast_1d(spcl.CalcPartitionSum(6, 1, atm["temps"]))
def step():
    ls.abscoeff_layers(atm["temps"], atm["press"], tvib=atm["tvib"], q_part=q, g_lo=g_lo, g_hi=g_hi, out=(ab, em))
    return engine.limb_rays((ab, em), los)
for _ in range(warm): step()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_steps + 1)]
host = []
t0 = time.perf_counter()
ev[0].record()
for i in range(n_steps):
    step()
    ev[i + 1].record()
    host.append(time.perf_counter() - t0)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
dev = [ev[i].elapsed_time(ev[i + 1]) for i in range(n_steps)]
print("shard %s: %d steps, wall %.3f ms/step" % (shard, n_steps, wall / n_steps * 1e3))
print("device ms between step ends:", " ".join("%.2f" % d for d in dev))
print("host enqueue done at (ms):", " ".join("%.1f" % (h * 1e3) for h in host[:20]), "...")
