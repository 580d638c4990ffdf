import sys, time, os; sys.path.insert(0,'.')
import numpy as np, torch
import bench as B
from spectrobot_amd import engine, synthetic as syn, spect_classes as spcl, distributed as sd
engine.set_device(0)
grid = syn.make_grid(2975.0, 5e-4, 100000)
L = syn.make_lines(100000, grid, config_id=2, n_levels=12)
atm = syn.make_atmosphere(80, 12)
ls = engine.LineSet(L, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
los, Lr = B.build_rays(syn, engine, atm, 1)
q = np.atleast_1d(spcl.CalcPartitionSum(6, 1, atm["temps"]))
for W, r in ((8, 6), (4, 1), (2, 1)):
    lo, hi = sd.shard_bounds(100000, W, r)
    ab = torch.empty((80, hi-lo), dtype=torch.float64, device="cuda"); em = torch.empty_like(ab)
    def step():
        ls.abscoeff_layers(atm["temps"], atm["press"], tvib=atm["tvib"], q_part=q, g_lo=lo, g_hi=hi, out=(ab, em))
        return engine.limb_rays((ab, em), los)
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(60): rr = step()
    torch.cuda.synchronize(); dt = (time.perf_counter()-t0)/60
    print("%s NW=%s shard %d/%d: %.3f ms/step checksum %.12g" % (os.environ.get("SPECTROBOT_HIP_LIB","default")[-12:], os.environ.get("SR_ZONES_NW","auto"), r, W, dt*1e3, float(rr.sum())))
