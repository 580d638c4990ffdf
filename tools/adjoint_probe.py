#!/usr/bin/env python3
"""The one-pass Jacobian kernels on the configs[3] set (8 rays x 80 layers x 2e5 points, T + VMR Jacobians): path order
with one ray per thread (mode 2), with two rays per thread sharing the loads (3), folded (0, the default); HIP-event
times and the largest deviation of the folded kernel's values from mode 2's, relative to each row's largest."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_configs as bc
from spectrobot_amd import engine, synthetic as syn
engine.set_device(0)
n, nl = 200000, 80
grid, L, atm, e_lev = bc.ch4_case(2000, n, nl, config_id=3, w0=2950.0)   # few lines: only the recursion is timed
ls = engine.LineSet(L, grid, 6, 1, syn.CH4_MM, e_lev)
a = bc.sza_atmosphere(atm, 51.0)
co = ls.abscoeff_layers(a["temps"], a["press"], tvib=a["tvib"])
dco = (co[0] * 0.01, co[1] * 0.01)
vm = np.full(nl, 0.0148)
Lr = syn.limb_los(atm["z"], atm["nd"], [vm], 120.0 + 60.0 * np.arange(8))
los = engine.LimbLOS(Lr["seg_off"], Lr["seg_layer"], Lr["pt_off"], Lr["x"], Lr["nd"], Lr["vmr"], col_scale=[syn.CH4_ISO_RATIO])
W = bc.layer_vmr_weights(atm["z"], Lr["alt"])
pg = np.zeros(nl, np.int32)
res = {}
for mode in (2, 3, 0, 2, 3, 0):
    engine.set_jac_layer_mode(mode)
    f = lambda: engine.limb_rays_jacobians(co, los, dcoeffs=dco, par_gas=pg, par_w=W)
    res[mode] = [x.clone() for x in f()]; torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record(); torch.cuda.synchronize()
    print("mode %d: %.3f ms per set" % (mode, e0.elapsed_time(e1) / 10))
engine.set_jac_layer_mode(0)
for name, x, y in zip(("rad", "jac_layer", "jac_par"), res[0], res[2]):
    sc = y.abs().amax(dim=-1, keepdim=True).clamp_min(1e-300)
    print("folded vs path order, %s: %.2e of the row maximum" % (name, float(((x - y).abs() / sc).max())))
