#!/usr/bin/env python3
"""One seed of tools/stress_fold.py in detail: where the folded kernels deviate."""
import os, sys, runpy
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
seed = int(sys.argv[1])
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "stress_fold.py")).read()
# run the body for one seed and keep the variables
src = src.replace("for seed in range(s0, s0 + ns):", "for seed in range(%d, %d):" % (seed, seed + 1))
g = {"__name__": "__main__", "__file__": os.path.join(os.path.dirname(os.path.abspath(__file__)), "stress_fold.py")}
sys.argv = [sys.argv[0], str(seed), "1"]
exec(compile(src, "stress_fold.py", "exec"), g)
out, L, torch = g["out"], g["L"], g["torch"]
names = ("rad", "rad(jac)", "jac", "rad(jacs)", "jl", "jp")
for i, nm in enumerate(names):
    x, y, zf = out[0][i], out[2][i], out[1][i]
    sc = y.abs().amax(dim=-1, keepdim=True).clamp_min(1e-10 * float(y.abs().max())).clamp_min(1e-250)
    dv = (x - y).abs() / sc
    idx = np.unravel_index(int(dv.argmax()), dv.shape)
    print("%-10s fold vs path %.2e  fold vs fwd %.2e  path vs fwd %.2e  at %s: fold %.6e path %.6e fwd %.6e rowmax %.3e" % (
        nm, float(dv.max()), float(((x - zf).abs() / sc).max()), float(((y - zf).abs() / sc).max()), idx,
        float(x[idx]), float(y[idx]), float(zf[idx]), float(sc[idx[:-1]][0])))
so, sl = L["seg_off"], L["seg_layer"]
print("ray 0 layers:", list(sl[so[0]:so[1]]))
