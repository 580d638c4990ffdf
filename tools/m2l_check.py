"""Box-pair far field (mode 2) against the per-line far field (mode 1) and the exact mode on config 2:
8 layers exact, per-kernel times on all 80."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spectrobot_amd import engine as eng, synthetic as syn

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    grid = syn.make_grid(2975.0, 5e-4, n)
    L = syn.make_lines(n, grid, config_id=2)
    atm = syn.make_atmosphere(80, 12)
    ls = eng.LineSet(L, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
    sel = np.arange(0, 80, 10)
    T, P, tv = atm["temps"][sel], atm["press"][sel], atm["tvib"][:, sel]
    out = {}
    for mode in (0, 1, 2):
        eng.set_far_field(mode)
        a, e = ls.abscoeff_layers(T, P, tvib=tv)
        out[mode] = (a.clone(), e.clone())
    for m in (1, 2):
        ra = ((out[m][0] - out[0][0]).abs() / out[0][0].abs()).amax(dim=1).cpu().numpy()
        re = ((out[m][1] - out[0][1]).abs() / out[0][1].abs()).amax(dim=1).cpu().numpy()
        print("mode %d vs exact, max rel err per layer (abs):" % m, " ".join(f"{x:.1e}" for x in ra))
        print("mode %d vs exact, max rel err per layer (emi):" % m, " ".join(f"{x:.1e}" for x in re))
    T, P, tv = atm["temps"], atm["press"], atm["tvib"]
    for mode in (1, 2):
        eng.set_far_field(mode)
        eng.set_overlap(0)
        for _ in range(2): ls.abscoeff_layers(T, P, tvib=tv)
        ms = np.zeros(5)
        for _ in range(5):
            ls.abscoeff_layers(T, P, tvib=tv); torch.cuda.synchronize(); ms += np.array(ls.last_kernel_ms())
        ms /= 5
        eng.set_overlap(1)
        for _ in range(3): ls.abscoeff_layers(T, P, tvib=tv)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): ls.abscoeff_layers(T, P, tvib=tv)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print("mode %d: prep %.3f far %.3f wings %.3f zones %.3f | pipelined %.3f ms/call" % (mode, ms[0], ms[1], ms[2], ms[3], dt * 1e3))
    eng.set_far_field(1)

if __name__ == "__main__":
    main()
