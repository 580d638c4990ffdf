#!/usr/bin/env python3
"""Per-kernel times of the coefficient op on config 2 for the library named by SPECTROBOT_HIP_LIB (tuning
aid: compare build variants made with `python spectrobot_amd/build.py --out X -DFLAG`).  Prints the serial
kernel times (5-launch average) and the pipelined ms/step (30 steps)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as B
from spectrobot_amd import engine, synthetic as syn, spect_classes as spcl
engine.set_device(0)
grid = syn.make_grid(2975.0, 5e-4, 100000)
L = syn.make_lines(100000, grid, config_id=2, n_levels=12)
atm = syn.make_atmosphere(80, 12)
ls = engine.LineSet(L, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
los, Lr = B.build_rays(syn, engine, atm, 1)
g_lo, g_hi = 0, 100000
if os.environ.get("SHARD"):     # SHARD=r/W: that spectral shard only
    from spectrobot_amd import distributed as sd
    r_, w_ = (int(v) for v in os.environ["SHARD"].split("/"))
    g_lo, g_hi = sd.shard_bounds(100000, w_, r_)
ab = torch.empty((80, g_hi - g_lo), dtype=torch.float64, device="cuda"); em = torch.empty_like(ab)
q = np.atleast_1d(spcl.CalcPartitionSum(6, 1, atm["temps"]))
def step():
    ls.abscoeff_layers(atm["temps"], atm["press"], tvib=atm["tvib"], q_part=q, g_lo=g_lo, g_hi=g_hi, out=(ab, em))
    return engine.limb_rays((ab, em), los)
engine.set_overlap(0)
for _ in range(3): step()
k = np.zeros(5)
for _ in range(8):
    step(); k += np.array(ls.last_kernel_ms()) / 8
if os.environ.get("SR_SERIAL_ONLY"):   # under rocprofv3 --kernel-trace: stand-alone kernel durations only
    torch.cuda.synchronize()
    print("serial: prep %.3f ff %.3f wings %.3f zones %.3f" % tuple(k[:4]))
    sys.exit(0)
engine.set_overlap(1)
for _ in range(4): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): r = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
print("%s: prep %.3f ff %.3f wings %.3f zones %.3f | %.3f ms/step (%.1f spectra/s) checksum %.15g" % (
    os.environ.get("SPECTROBOT_HIP_LIB", "default"), k[0], k[1], k[2], k[3], dt * 1e3, 1 / dt, float(r.sum())))
