#!/usr/bin/env python3
"""Where a configs[3] step goes (level-factored route): tables, populations (host), combine, Jacobians -- each phase
synchronised.  usage: tools/config3_probe.py [3d]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_configs as bc
from spectrobot_amd import engine, synthetic as syn

three_d = len(sys.argv) > 1 and sys.argv[1] == "3d"
engine.set_device(0)
n = int(os.environ.get("N", "200000"))
nl, n_rays = 80, 8
szas = [30.0, 37.0, 44.0, 51.0, 58.0, 65.0, 72.0, 80.0]
grid, L, atm, e_lev = bc.ch4_case(n, n, nl, config_id=3, w0=2950.0)
ls = engine.LineSet(L, grid, 6, 1, syn.CH4_MM, e_lev)
vm = np.full(nl, 0.0148)
tz = 120.0 + 60.0 * np.arange(n_rays)
pg = np.zeros(nl, np.int32)
if three_d:
    sets = [bc.los_3d_set(atm, vm, tz, sza, 22.5 * np.arange(n_rays)) for sza in szas]
else:
    sets = []
    for sza in szas:
        Lr = dict(syn.limb_los(atm["z"], atm["nd"], [vm], tz))
        Lr["state"] = bc.sza_atmosphere(atm, sza)
        Lr["seg_alt_layer"] = None
        sets.append(Lr)
for S in sets:
    S["los"] = engine.LimbLOS(S["seg_off"], S["seg_layer"], S["pt_off"], S["x"], S["nd"], S["vmr"], col_scale=[syn.CH4_ISO_RATIO])
    S["W"] = bc.layer_vmr_weights(atm["z"], S["alt"])
allT = np.concatenate([S["state"]["temps"] for S in sets]); allP = np.concatenate([S["state"]["press"] for S in sets])
tv_all = np.concatenate([S["state"]["tvib"] for S in sets], axis=1)
T_rows, P_rows, row = engine.LevelFactored.unique_rows(allT, allP)
print("rows %d steps %d" % (len(T_rows), len(row)))


def tick(msg, t0):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("  %-34s %8.2f ms" % (msg, (t1 - t0) * 1e3))
    return t1


for it in range(3):
    print("step", it)
    t = t00 = time.perf_counter()
    tab = ls.glevel_pairs(T_rows, P_rows); t = tick("tables at T", t)
    ls.set_bounds_temps(T_rows)
    tab_p = ls.glevel_pairs(T_rows + 0.002, P_rows)
    ls.set_bounds_temps(None); t = tick("tables at T + dT", t)
    pop, dpop = ls.level_populations(T_rows[row], tvib=tv_all, derivative=True); t = tick("populations (host)", t)
    (ca, ce), (da, de) = engine.glevel_combine(tab, row, pop, tab_dT=tab_p, dpop=dpop, dT=0.002); t = tick("combine, all sets", t)
    at = 0
    for S in sets:
        m = len(S["seg_layer"])
        kw = dict(seg_jac_row=S["seg_alt_layer"], n_jac_rows=nl) if three_d else {}
        res = engine.limb_rays_jacobians((ca[at:at + m], ce[at:at + m]), S["los"], dcoeffs=(da[at:at + m], de[at:at + m]), par_gas=pg, par_w=S["W"], **kw)
        at += m
    t = tick("Jacobians, %d sets" % len(sets), t)
    del tab, tab_p, ca, ce, da, de, res
    print("  total %.2f ms" % ((time.perf_counter() - t00) * 1e3))
