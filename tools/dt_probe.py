import sys, numpy as np, torch
sys.path.insert(0, '.')
from spectrobot_amd import engine as eng, synthetic as syn
eng.set_device(0)
grid = syn.make_grid(2990.0, 5e-4, 30000)
L = syn.make_lines(20000, grid, seed=5, n_levels=12, config_id=2)
atm = syn.make_atmosphere(12, 12)
ls = eng.LineSet(L, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
T, P, tv = atm["temps"], atm["press"], atm["tvib"]
co = ls.abscoeff_layers(T, P, tvib=tv)
def rel(x, y): return float(((x - y).abs().amax(dim=1) / y.abs().amax(dim=1)).max())
ls.set_bounds_temps(T)
res = {}
for dT in (0.05, 0.01, 0.005, 0.001, 0.0002):
    ap = ls.abscoeff_layers(T + dT, P, tvib=tv); am = ls.abscoeff_layers(T - dT, P, tvib=tv)
    res[dT] = ((ap[0] - co[0]) / dT, (ap[0] - am[0]) / (2 * dT))
ls.set_bounds_temps(None)
ref = res[0.01][1]  # central frozen 0.01
for dT, (f, c) in res.items():
    print("dT %.4f: forward vs ref %.2e   central vs ref %.2e" % (dT, rel(f, ref), rel(c, ref)))
apm = ls.abscoeff_layers(T + 0.05, P, tvib=tv); amm = ls.abscoeff_layers(T - 0.05, P, tvib=tv)
cm = (apm[0] - amm[0]) / 0.1
print("central moving 0.05 vs ref: %.2e" % rel(cm, ref))
d = (res[0.001][0] - ref).abs(); k = int(d.amax(dim=1).argmax()); j = int(d[k].argmax())
print("worst layer", k, "point", j, "values f", float(res[0.001][0][k, j]), "ref", float(ref[k, j]), "max", float(ref[k].abs().max()), "coef", float(co[0][k, j]), "coef max", float(co[0][k].abs().max()))
