"""Soak: many back-to-back calls of the whole chain with changing shapes on one LineSet; the device
memory in use must not grow and every 100th result must equal the first of its shape."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from spectrobot_amd import engine as eng, synthetic as syn  # noqa: E402


def main(n_calls=1500):
    grid = syn.make_grid(2980.0, 5e-4, 60000)
    L = syn.make_lines(20000, grid, seed=3, n_levels=12)
    ls = eng.LineSet(L, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
    rng = np.random.default_rng(0)
    shapes = []
    for _ in range(6):
        nl = int(rng.integers(5, 60))
        atm = syn.make_atmosphere(nl, 12)
        lo = int(rng.integers(0, 20000))
        hi = int(rng.integers(40000, 60001))
        sl, ln = syn.limb_path(atm["z"], atm["z"][0] + 1.0)
        col = ln * 1e5 * syn.number_density(atm["press"], atm["temps"])[sl] * 0.0148
        shapes.append((atm, lo, hi, [0, len(sl)], sl, col))
    first = {}
    free0 = None
    t0 = time.time()
    for c in range(n_calls):
        i = int(rng.integers(0, len(shapes)))
        atm, lo, hi, offs, sl, col = shapes[i]
        ab, em = ls.abscoeff_layers(atm["temps"], atm["press"], tvib=atm["tvib"], g_lo=lo, g_hi=hi)
        rad = eng.radiance_rays(ab, em, offs, sl, col)
        if i not in first:
            first[i] = rad.clone()
        elif c % 100 == 0:
            assert bool((rad == first[i]).all()), (c, i)
        if c == 200:
            torch.cuda.synchronize()
            free0 = torch.cuda.mem_get_info()[0]
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    print("%d calls in %.1f s; device memory free after 200 calls %.3f GiB, at the end %.3f GiB" % (
        n_calls, time.time() - t0, free0 / 2 ** 30, free1 / 2 ** 30))
    assert free1 > free0 - (256 << 20), "device memory keeps growing"
    print("soak OK")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1500)
