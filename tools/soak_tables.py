"""Soak of the level-table builds (the multi-channel pass): many back-to-back builds of pair tables and three-ctype tables
with changing row counts and shards, folded ops interleaved on the same handle; the device memory in use must not grow
and every 50th result must equal the first of its shape to 2e-12 of a spectrum's largest value (the shared LDS images
make the tables reproducible to rounding, not bit for bit)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from spectrobot_amd import engine as eng, synthetic as syn  # noqa: E402


def plane_err(a, b):
    s = b.abs().amax(dim=-1, keepdim=True).clamp_min(1e-300)
    return float(((a - b).abs() / s).max())


def main(n_calls=600):
    grid = syn.make_grid(2980.0, 5e-4, 40000)
    L = syn.make_lines(20000, grid, seed=3, n_levels=12)
    ls = eng.LineSet(L, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
    rng = np.random.default_rng(0)
    shapes = []
    for _ in range(5):
        atm = syn.make_atmosphere(int(rng.integers(3, 24)), 12)
        shapes.append((atm, int(rng.integers(0, 12000)), int(rng.integers(28000, 40001))))
    first, free0, worst = {}, None, 0.0
    t0 = time.time()
    for c in range(n_calls):
        i = int(rng.integers(0, len(shapes)))
        atm, lo, hi = shapes[i]
        what = c % 3
        if what == 0:
            r = ls.glevel_pairs(atm["temps"], atm["press"], g_lo=lo, g_hi=hi)
        elif what == 1:
            r = ls.gcoeff_levels(atm["temps"], atm["press"], g_lo=lo, g_hi=hi)
        else:
            r = torch.stack(ls.abscoeff_layers(atm["temps"], atm["press"], tvib=atm["tvib"], g_lo=lo, g_hi=hi))
        key = (i, what)
        if key not in first:
            first[key] = r.clone()
        elif c % 50 < 3:
            worst = max(worst, plane_err(r, first[key]))
        del r
        if c == 100:
            torch.cuda.synchronize()
            free0 = torch.cuda.mem_get_info()[0]
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    print("%d calls in %.1f s; device memory free after 100 calls %.3f GiB, at the end %.3f GiB; worst deviation from the "
          "first result of a shape %.1e" % (n_calls, time.time() - t0, free0 / 2 ** 30, free1 / 2 ** 30, worst))
    assert free1 > free0 - (256 << 20), "device memory keeps growing"
    assert worst < 2e-12
    print("soak OK")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 600)
