import sys; sys.path.insert(0,'/root/repo')
import numpy as np, ctypes as C
from spectrobot_amd import engine as eng, synthetic as syn
from oracle import oracle
from spectrobot_amd._lib import lib, dp
eng.set_device(0)
grid = syn.make_grid(2975.0, 5e-4, 30000)
L = syn.make_lines(3000, grid, config_id=7, n_levels=12)
atm = syn.make_atmosphere(6, 12)
q=np.zeros(6); t=np.ascontiguousarray(atm["temps"]); lib.sr_calc_partition_sum(6,1,t.ctypes.data_as(dp),6,q.ctypes.data_as(dp))
ls = eng.LineSet(L, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
eng.set_far_field(1); a1,e1 = ls.abscoeff_layers(atm["temps"], atm["press"], tvib=atm["tvib"])
eng.set_far_field(0); a0,e0 = ls.abscoeff_layers(atm["temps"], atm["press"], tvib=atm["tvib"])
abo, emo = oracle.abscoeff_layers(L, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES, atm["temps"], atm["press"], q, atm["tvib"], grid, mode=1, n_threads=6)
a1=a1.cpu().numpy(); a0=a0.cpu().numpy()
for name,x,y in (("far vs oracle",a1,abo),("exact vs oracle",a0,abo),("far vs exact",a1,a0)):
    r=np.abs(x-y)/np.abs(y); k,j=np.unravel_index(np.argmax(r),r.shape)
    print(name, r.max(), 'layer',k,'j',j, 'P',atm["press"][k], 'val', y[k,j], 'rowmax', y[k].max())
    print('  per layer max', r.max(axis=1))
r=np.abs(a1-a0)/np.abs(a0)
k=np.argmax(r.max(axis=1)); 
idx=np.argsort(r[k])[-10:]; print(k, idx, r[k][idx])
ic=np.searchsorted(grid, L["freq"]); 
j=idx[-1]; print('nearest lines dist', np.sort(np.abs(ic-j))[:5])
