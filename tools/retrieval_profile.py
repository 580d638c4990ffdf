"""Where a configs[4] retrieval iteration spends its time: cProfile of the 20-iteration loop (host side) next to the
loop's wall clock, and the kernel timeline's share (forward kernel + instrument step by HIP events)."""
import os, sys, time, copy, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench_configs as BC
from spectrobot_amd import engine, retrieval
engine.set_device(0)
scene = BC.two_gas_scene(40000, 8000, 60000, 60)
bs, pixels, x_true = BC.retrieval_problem(scene)
bs0 = copy.deepcopy(bs)
retrieval.inversion_fast_limb(scene, copy.deepcopy(bs0), pixels, max_it=20)      # warm
torch.cuda.synchronize()
gc.collect(); gc.freeze()
for rep in range(3):
    b = copy.deepcopy(bs0)
    t0 = time.perf_counter()
    retrieval.inversion_fast_limb(scene, b, pixels, max_it=20)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("loop %d: %d iterations, %.3f ms per iteration (%.0f it/s)" % (rep, len(b.history), dt / len(b.history) * 1e3, len(b.history) / dt))
# wall-clock share of the iteration's phases (wrappers around the calls of retrieval.simulate / inversion_fast_limb)
from spectrobot_amd import spect_main_module as smm
acc = {}
def timed(obj, name, label):
    f = getattr(obj, name)
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t
    setattr(obj, name, w)
    return f
saved = [(type(scene), "los", timed(type(scene), "los", "scene.los (VMRs at the sample points, set_vmr)")),
         (type(scene), "profile_weights", timed(type(scene), "profile_weights", "profile_weights")),
         (type(scene), "coefficient_stack", timed(type(scene), "coefficient_stack", "coefficient_stack")),
         (engine, "retrieval_forward", timed(engine, "retrieval_forward", "retrieval_forward (one call: enqueue + wait for the GPU + copy + FOV)")),
         (engine, "limb_rays_jacobian", timed(engine, "limb_rays_jacobian", "limb_rays_jacobian (enqueue)")),
         (engine, "hires_to_lowres", timed(engine, "hires_to_lowres", "hires_to_lowres (enqueue + wait for the GPU + copy)")),
         (smm, "fov_closed_form", timed(smm, "fov_closed_form", "fov_closed_form")),
         (smm, "chicalc", timed(smm, "chicalc", "chicalc")),
         (smm, "retrieval_converged", timed(smm, "retrieval_converged", "retrieval_converged")),
         (smm, "inversion_algebra", timed(smm, "inversion_algebra", "inversion_algebra")),
         (retrieval, "simulate", timed(retrieval, "simulate", "simulate (total)"))]
b = copy.deepcopy(bs0)
t0 = time.perf_counter()
retrieval.inversion_fast_limb(scene, b, pixels, max_it=20)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
n = len(b.history)
print("phases of one iteration (%d iterations, %.3f ms each with the wrappers):" % (n, dt / n * 1e3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-74s %7.1f us" % (k, v / n * 1e6))
for obj, name, f in saved:
    setattr(obj, name, f)
import cProfile, pstats
b = copy.deepcopy(bs0)
pr = cProfile.Profile(); pr.enable()
retrieval.inversion_fast_limb(scene, b, pixels, max_it=20)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
