"""Far-field modes 1 and 2 against the exact mode on the random configurations of
tests/test_gpu_parity.py::test_randomized_configs_far_vs_exact_vs_oracle: where the largest difference sits.
usage: python tools/m2l_debug.py FIRST_SEED END_SEED"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectrobot_amd import engine as eng, synthetic as syn
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(1000 + seed)
    step = float(rng.choice([2.5e-4, 5e-4, 1e-3, 2e-3]))
    n_grid = int(rng.integers(700, 9000))
    w0 = float(rng.choice([650.0, 2100.0, 2990.0, 4300.0]))
    grid = syn.make_grid(w0, step, n_grid)
    n_lines = int(rng.integers(1, 900))
    nlev = int(rng.choice([0, 3, 12]))
    L = syn.make_lines(n_lines, grid, seed=2000 + seed, n_levels=nlev)
    if n_lines > 4:
        L["freq"][0] = grid[0] - 1500 * step
        L["freq"][-1] = grid[-1] + 2000 * step
    mm = float(rng.choice([16.0313, 27.994915, 2.0159, 44.0]))
    nl = 3
    T = rng.uniform(70, 300, nl)
    P = 10.0 ** rng.uniform(-7, 3.3, nl)
    e_lev = syn.CH4_LEVEL_ENERGIES[:nlev]
    tv = None if nlev == 0 else np.array([T + 2.0 * i for i in range(nlev)])
    q = rng.uniform(50, 500, nl)
    ls = eng.LineSet(L, grid, 6, 1, mm, e_lev)
    lo = int(rng.integers(0, n_grid // 3))
    hi = int(rng.integers(2 * n_grid // 3, n_grid + 1))
    res = {}
    for far in (0, 1, 2):
        eng.set_far_field(far)
        a, e = ls.abscoeff_layers(T, P, tvib=tv, q_part=q, g_lo=lo, g_hi=hi)
        res[far] = a.cpu().numpy()
    eng.set_far_field(eng.FAR_FIELD_DEFAULT)
    nz = res[0] != 0
    for far in (1, 2):
        d = np.abs(res[far] / np.where(nz, res[0], 1) - 1) * nz
        k, j = np.unravel_index(np.argmax(d), d.shape)
        print("seed %d far %d: step %g n_grid %d lines %d mm %g P %s T %s lo %d hi %d | max rel %.2e at layer %d point %d (of %d)" % (
            seed, far, step, n_grid, n_lines, mm, np.array2string(P, precision=2), np.array2string(T, precision=0), lo, hi, d.max(), k, j, hi - lo), flush=True)
