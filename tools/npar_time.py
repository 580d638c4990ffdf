import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench_configs as BC
from spectrobot_amd import engine
engine.set_device(0)
scene = BC.two_gas_scene(40000, 8000, 60000, 60)
bs, pixels, x_true = BC.retrieval_problem(scene)
alts = [a for pix in pixels for a in pix.los_alts()]
los, alt = scene.los(alts)
coeffs = scene.coefficient_stack()
z = scene.z
zz = np.append(z, z[-1] + (z[-1] - z[-2]))
for n_par in (4, 7, 8, 9, 12, 16, 24, 32):
    nodes = np.linspace(z[0], z[-1], n_par)
    W = np.array([np.interp(alt, zz, np.clip(1.0 - np.abs(zz - c) / (nodes[1] - nodes[0]), 0.0, None)) for c in nodes])
    pg = (np.arange(n_par) % 2).astype(np.int32)
    for _ in range(2): engine.limb_rays_jacobian(coeffs, los, pg, W)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(5): engine.limb_rays_jacobian(coeffs, los, pg, W)
    ev[1].record(); torch.cuda.synchronize()
    print("n_par %2d: %.3f ms per call (18 LOS x 60000 points, per-call staging)" % (n_par, ev[0].elapsed_time(ev[1]) / 5))
