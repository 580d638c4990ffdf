#!/usr/bin/env python3
"""Joules per flop of a pure v_fma_f64 stream (tools/microbench/fma_energy, run as a child process) beside the power sampler
of tools/energy_by_kernel.py: the floor the zones kernel's 36 pJ per flop is set against."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools.energy_by_kernel import Smi, Sampler

here = os.path.dirname(os.path.abspath(__file__))
exe = os.path.join(here, "microbench", "fma_energy")
if not os.path.exists(exe):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", exe + ".hip", "-o", exe])
smi = Smi()
sam = Sampler(smi, 0)
sam.start()
time.sleep(1.0)
t_idle = time.perf_counter()
time.sleep(1.0)
w = sam.window(t_idle, time.perf_counter())
print("idle: %.0f W, %.0f MHz" % (np.nanmean(w[:, 1]), np.nanmean(w[:, 2])))
t0 = time.perf_counter()
out = subprocess.run([exe, "4"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120).stdout.decode().strip()
t1 = time.perf_counter()
w = sam.window(t0 + 1.2, t1 - 0.3)
sam.stop_flag = True
print(out)
tf = float(out.split()[-2])
p, f = np.nanmean(w[:, 1]), np.nanmean(w[:, 2])
print("under the stream: %.0f W, sclk %.0f MHz -> %.1f pJ per flop (%.1f pJ above idle)  [%d samples]"
      % (p, f, p / (tf * 1e12) * 1e12, (p - 290.0) / (tf * 1e12) * 1e12, len(w)))
