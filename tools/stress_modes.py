"""Randomized cross-check of the two evaluation modes (far-field + near kernels vs exact kernels) on
many small configurations: grids, shard offsets, line densities, molar masses, pressures from the
Doppler to the Lorentz regime, with and without non-LTE levels.  GPU only (the oracle is not used:
tests/test_gpu_parity.py pins 16 such configurations to the oracle; this sweeps hundreds).

  python tools/stress_modes.py [first_seed] [n_seeds] [--large]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectrobot_amd import engine as eng, synthetic as syn  # noqa: E402


FAR = eng.FAR_FIELD_DEFAULT  # --far1: the per-line far field (sr_set_far_field(1)) instead of the default box pairs
LARGE = False   # --large: grids up to 1.2e5 points, up to 70 layers, up to one line per point (all kernel paths)


def one(seed):
    rng = np.random.default_rng(50000 + seed)
    step = float(rng.choice([2.5e-4, 5e-4, 1e-3, 2e-3]))
    n_grid = int(rng.integers(300, 40000)) if not LARGE else int(rng.integers(20000, 120000))
    w0 = float(rng.choice([650.0, 2100.0, 2990.0, 4300.0]))
    grid = syn.make_grid(w0, step, n_grid)
    n_lines = int(rng.integers(1, 3000)) if not LARGE else int(rng.integers(2000, n_grid))
    nlev = int(rng.choice([0, 3, 12]))
    L = syn.make_lines(n_lines, grid, seed=60000 + seed, n_levels=nlev)
    if n_lines > 4 and rng.random() < 0.5:   # clustered lines: many share a grid point
        L["freq"][: n_lines // 3] = np.sort(rng.uniform(grid[n_grid // 2], grid[n_grid // 2] + 40 * step, n_lines // 3))
        order = np.argsort(L["freq"], kind="stable")
        L = {k: v[order] for k, v in L.items()}
    mm = float(rng.choice([16.0313, 27.994915, 2.0159, 44.0]))
    nl = int(rng.integers(1, 5)) if not LARGE else int(rng.integers(8, 70))
    T = rng.uniform(70, 300, nl)
    P = 10.0 ** rng.uniform(-7, 3.3, nl)
    tv = None if nlev == 0 else np.array([T + 2.0 * i for i in range(nlev)])
    q = rng.uniform(50, 500, nl)
    ls = eng.LineSet(L, grid, 6, 1, mm, syn.CH4_LEVEL_ENERGIES[:nlev])
    lo = int(rng.integers(0, n_grid // 3))
    hi = int(rng.integers(2 * n_grid // 3, n_grid + 1))
    eng.set_far_field(0)
    a0, e0 = ls.abscoeff_layers(T, P, tvib=tv, q_part=q, g_lo=lo, g_hi=hi)
    eng.set_far_field(FAR)
    a1, e1 = ls.abscoeff_layers(T, P, tvib=tv, q_part=q, g_lo=lo, g_hi=hi)
    a0, e0, a1, e1 = (x.cpu().numpy() for x in (a0, e0, a1, e1))
    nz = a0 != 0
    assert np.array_equal(a1 != 0, nz), "zero pattern differs (seed %d)" % seed
    # abs = sum of (absorption - stimulated emission) terms: under non-LTE populations it crosses zero, and a plain
    # relative difference there measures the cancellation, not the kernels (emi, a sum of positive terms, never
    # shows it).  So: relative to the envelope of |abs| over +-100 points.
    from scipy.ndimage import maximum_filter1d
    env = maximum_filter1d(np.abs(a0), 201, axis=1, mode="nearest")
    da = (np.abs(a1 - a0)[nz] / env[nz]).max() if nz.any() else 0.0
    nze = e0 != 0
    de = np.abs(e1[nze] / e0[nze] - 1).max() if nze.any() else 0.0
    return max(da, de), (n_grid, n_lines, nl, hi - lo, "abs %.1e emi %.1e" % (da, de))


if __name__ == "__main__":
    if "--far1" in sys.argv:
        FAR = 1
        sys.argv.remove("--far1")
    if "--large" in sys.argv:
        LARGE = True
        sys.argv.remove("--large")
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    worst = 0.0
    for s in range(first, first + n):
        d, cfg = one(s)
        worst = max(worst, d)
        if d > 2e-11:
            print("seed", s, cfg, "diff %.2e" % d, flush=True)
        if (s - first) % 50 == 49:
            print("...", s - first + 1, "seeds, worst %.2e" % worst, flush=True)
    print("seeds %d..%d: worst relative difference between the modes %.2e" % (first, first + n - 1, worst))
    sys.exit(0 if worst < 1e-9 else 1)
