"""The forward model of one configs[4] retrieval iteration (18 LOS x (radiance + 7 parameter Jacobians), two gases) a
few times: the target of tools/pmc_cmd.sh for the counters of sr_limb_fold_sens_lds_kernel, and its HIP-event time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench_configs as BC
from spectrobot_amd import engine
engine.set_device(0)
scene = BC.two_gas_scene(40000, 8000, 60000, 60)
bs, pixels, x_true = BC.retrieval_problem(scene)
alts = [a for pix in pixels for a in pix.los_alts()]
los, alt = scene.los(alts)
coeffs = scene.coefficient_stack()
par_gas, par_w = scene.profile_weights(bs, alt)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for _ in range(2):
    out = engine.limb_rays_jacobian(coeffs, los, par_gas, par_w, joint=True, resident=True)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(n):
    out = engine.limb_rays_jacobian(coeffs, los, par_gas, par_w, joint=True, resident=True)
ev[1].record(); torch.cuda.synchronize()
buf = out[2]
print("forward model: %.3f ms per call (%d calls), checksum %.17g" % (ev[0].elapsed_time(ev[1]) / n, n, float(buf.double().sum().item())))
