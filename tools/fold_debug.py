#!/usr/bin/env python3
"""Where the folded one-pass kernel deviates most from the path-order one (configs[3] set), with the forward-sensitivity
kernel as the third opinion."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_configs as bc
from spectrobot_amd import engine, synthetic as syn
engine.set_device(0)
n, nl = 200000, 80
grid, L, atm, e_lev = bc.ch4_case(2000, n, nl, config_id=3, w0=2950.0)
ls = engine.LineSet(L, grid, 6, 1, syn.CH4_MM, e_lev)
a = bc.sza_atmosphere(atm, 51.0)
co = ls.abscoeff_layers(a["temps"], a["press"], tvib=a["tvib"])
dco = (co[0] * 0.01, co[1] * 0.01)
vm = np.full(nl, 0.0148)
Lr = syn.limb_los(atm["z"], atm["nd"], [vm], 120.0 + 60.0 * np.arange(8))
los = engine.LimbLOS(Lr["seg_off"], Lr["seg_layer"], Lr["pt_off"], Lr["x"], Lr["nd"], Lr["vmr"], col_scale=[syn.CH4_ISO_RATIO])
res = {}
for mode in (0, 2, 1):
    engine.set_jac_layer_mode(mode)
    res[mode] = [x.clone() for x in engine.limb_rays_jacobians(co, los, dcoeffs=dco)[:2]]
engine.set_jac_layer_mode(0)
x, y, z = res[0][1], res[2][1], res[1][1]
sc = y.abs().amax(dim=-1, keepdim=True).clamp_min(1e-300)
dev = ((x - y).abs() / sc)
print("fold vs path: %.2e   path vs forward: %.2e   fold vs forward: %.2e" % (float(dev.max()), float(((y - z).abs() / sc).max()), float(((x - z).abs() / sc).max())))
idx = np.unravel_index(int(dev.argmax()), dev.shape)
r, k, j = [int(v) for v in idx]
print("ray %d row %d point %d: fold %.17e path %.17e forward %.17e row max %.3e" % (r, k, j, float(x[r, k, j]), float(y[r, k, j]), float(z[r, k, j]), float(sc[r, k, 0])))
print("abs[k, j] = %.3e, emi %.3e, rad fold %.17e path %.17e" % (float(co[0][k, j]), float(co[1][k, j]), float(res[0][0][r, j]), float(res[2][0][r, j])))
so, sl = Lr["seg_off"], Lr["seg_layer"]
print("segments of the ray:", list(sl[so[r]:so[r + 1]]))
top = torch.topk(dev.flatten(), 8).indices.cpu().numpy()
for t in top:
    rr, kk, jj = [int(v) for v in np.unravel_index(int(t), dev.shape)]
    print("  ray %d row %d point %d dev %.2e  abs %.3e  fold %.6e path %.6e fwd %.6e" % (rr, kk, jj, float(dev[rr, kk, jj]), float(co[0][kk, jj]), float(x[rr, kk, jj]), float(y[rr, kk, jj]), float(z[rr, kk, jj])))
col = np.asarray(los.columns())[0]
ab = co[0][:, j].cpu().numpy(); em = co[1][:, j].cpu().numpy()
s0 = so[r]
tau = np.array([ab[sl[s]] * col[s] for s in range(so[r], so[r + 1])])
print("min abs over layers at the point: %.3e; tau per segment (far -> near):" % ab.min())
print(np.array2string(tau, precision=3, max_line_width=200))
print("source function emi/abs per layer:", np.array2string(em / ab, precision=3, max_line_width=200))
