"""Far-field expansion error and per-kernel times on the BASELINE config-2 workload.

Compares the default mode (far-field expansions + exact near field) with the exact
mode (every (line, point) pair evaluated) on 8 of the 80 layers (every 10th, from the
Lorentz- to the Doppler-dominated end), then times the default mode on all 80.
Used by tools/sweep_farfield.sh to choose (kTheta, kFD).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from spectrobot_amd import engine as eng, synthetic as syn  # noqa: E402


def main():
    n_lines, n_grid = 100000, 100000
    grid = syn.make_grid(2975.0, 5e-4, n_grid)
    L = syn.make_lines(n_lines, grid, config_id=2)
    atm = syn.make_atmosphere(80, 12)
    ls = eng.LineSet(L, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
    sel = np.arange(0, 80, 10)
    T, P, tv = atm["temps"][sel], atm["press"][sel], atm["tvib"][:, sel]
    eng.set_far_field(0)
    a0, e0 = ls.abscoeff_layers(T, P, tvib=tv)
    eng.set_far_field(eng.FAR_FIELD_DEFAULT)
    a1, e1 = ls.abscoeff_layers(T, P, tvib=tv)
    ra = ((a1 - a0).abs() / a0.abs()).amax(dim=1).cpu().numpy()
    re = ((e1 - e0).abs() / e0.abs()).amax(dim=1).cpu().numpy()
    print("far-field vs exact, max rel err per layer (abs):", " ".join(f"{x:.1e}" for x in ra))
    print("far-field vs exact, max rel err per layer (emi):", " ".join(f"{x:.1e}" for x in re))
    del a0, e0, a1, e1
    T, P, tv = atm["temps"], atm["press"], atm["tvib"]
    eng.set_overlap(0)  # per-kernel times
    for _ in range(2):
        ls.abscoeff_layers(T, P, tvib=tv)
    torch.cuda.synchronize()
    ms = np.zeros(5)
    for _ in range(5):
        ls.abscoeff_layers(T, P, tvib=tv)
        torch.cuda.synchronize()
        ms += np.array(ls.last_kernel_ms())
    ms /= 5
    print("kernel ms (prep, far field, near wings, near zones, -): " + " ".join(f"{x:.2f}" for x in ms),
          f"sum {ms.sum():.2f}")
    eng.set_overlap(1)
    ls.abscoeff_layers(T, P, tvib=tv)
    tot = 0.0
    for _ in range(5):
        ls.abscoeff_layers(T, P, tvib=tv)
        tot += sum(ls.last_kernel_ms())
    print(f"with overlap (default): prep + coefficient op {tot / 5:.2f} ms")


if __name__ == "__main__":
    main()
