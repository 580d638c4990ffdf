#!/usr/bin/env python3
"""What bounds the headline step: Joules, clock and time per kernel ("Next round" 2).

Each kernel of the coefficient op ALONE, looped for seconds (sr_set_kernel_repeat on the serial schedule), and the whole
step (pipelined and serial), beside a sampler thread that reads socket power, the shader clock and the energy
accumulator through librocm_smi64 at ~50 Hz.  Per row: launches, ms per launch (wall over the loop), mean W, mean sclk,
J per launch (mean W x time, and from the energy counter where the box has one).

    python3 tools/energy_by_kernel.py [--seconds 3] [--tag NAME]      (SPECTROBOT_HIP_LIB=<variant.so> for a variant build)
"""
import argparse
import ctypes as C
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np  # noqa: E402


class Smi(object):
    """librocm_smi64 by ctypes: power [W], sclk [MHz], energy [J] of every device it lists."""

    class Freqs(C.Structure):
        _fields_ = [("has_deep_sleep", C.c_bool), ("num_supported", C.c_uint32), ("current", C.c_uint32),
                    ("frequency", C.c_uint64 * 33)]

    def __init__(self):
        self.L = C.CDLL("/opt/rocm/lib/librocm_smi64.so")
        rc = self.L.rsmi_init(C.c_uint64(0))
        if rc != 0:
            raise RuntimeError("rsmi_init -> %d" % rc)
        n = C.c_uint32(0)
        self.L.rsmi_num_monitor_devices(C.byref(n))
        self.n = n.value

    def power(self, d):
        p = C.c_uint64(0)
        if self.L.rsmi_dev_current_socket_power_get(C.c_uint32(d), C.byref(p)) == 0:
            return p.value * 1e-6
        t = C.c_int(0)
        if self.L.rsmi_dev_power_get(C.c_uint32(d), C.byref(p), C.byref(t)) == 0:
            return p.value * 1e-6
        if self.L.rsmi_dev_power_ave_get(C.c_uint32(d), C.c_uint32(0), C.byref(p)) == 0:
            return p.value * 1e-6
        return float("nan")

    def sclk(self, d):
        f = Smi.Freqs()
        if self.L.rsmi_dev_gpu_clk_freq_get(C.c_uint32(d), C.c_int(0), C.byref(f)) == 0 and f.current < 33:
            return f.frequency[f.current] * 1e-6
        return float("nan")

    def energy(self, d):
        e, res, ts = C.c_uint64(0), C.c_float(0), C.c_uint64(0)
        if self.L.rsmi_dev_energy_count_get(C.c_uint32(d), C.byref(e), C.byref(res), C.byref(ts)) == 0:
            return e.value * float(res.value) * 1e-6     # micro Joules -> J
        return float("nan")


class Sampler(threading.Thread):
    def __init__(self, smi, dev, period=0.02):
        threading.Thread.__init__(self, daemon=True)
        self.smi, self.dev, self.period = smi, dev, period
        self.rows, self.stop_flag = [], False

    def run(self):
        while not self.stop_flag:
            t = time.perf_counter()
            self.rows.append((t, self.smi.power(self.dev), self.smi.sclk(self.dev), self.smi.energy(self.dev)))
            time.sleep(max(0.0, self.period - (time.perf_counter() - t)))

    def window(self, t0, t1):
        r = np.array([x for x in self.rows if t0 <= x[0] <= t1])
        return r if len(r) else np.zeros((0, 4))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--tag", default=os.path.basename(os.environ.get("SPECTROBOT_HIP_LIB", "")) or "in-tree")
    ap.add_argument("--lines", type=int, default=100000)
    ap.add_argument("--grid", type=int, default=100000)
    ap.add_argument("--layers", type=int, default=80)
    args = ap.parse_args()
    import torch
    import bench as B
    from spectrobot_amd import engine, synthetic as syn, spect_classes as spcl
    from spectrobot_amd._lib import lib
    engine.set_device(0)
    smi = Smi()
    grid = syn.make_grid(2975.0, 5e-4, args.grid)
    L = syn.make_lines(args.lines, grid, config_id=2, n_levels=12)
    atm = syn.make_atmosphere(args.layers, 12)
    ls = engine.LineSet(L, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
    los, Lr = B.build_rays(syn, engine, atm, 1)
    ab = torch.empty((args.layers, args.grid), dtype=torch.float64, device="cuda")
    em = torch.empty_like(ab)
    q = np.atleast_1d(spcl.CalcPartitionSum(6, 1, atm["temps"]))

    def op():
        ls.abscoeff_layers(atm["temps"], atm["press"], tvib=atm["tvib"], q_part=q, out=(ab, em))

    def step():
        los.refresh_columns()
        ls.limb_step(atm["temps"], atm["press"], los, tvib=atm["tvib"], q_part=q, out=(ab, em))

    # which rsmi device is ours: the one whose power rises under load
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    idle = [smi.power(d) for d in range(smi.n)]
    t_end = time.perf_counter() + 1.0
    while time.perf_counter() < t_end:
        step()
    busy = [smi.power(d) for d in range(smi.n)]
    torch.cuda.synchronize()
    dev = int(np.nanargmax(np.array(busy) - np.array(idle)))
    print("# %s  rsmi devices %d, ours %d (idle %s W, busy %s W)" % (args.tag, smi.n, dev, [round(v) for v in idle], [round(v) for v in busy]), flush=True)
    sam = Sampler(smi, dev)
    sam.start()
    time.sleep(1.0)
    t0 = time.perf_counter()
    time.sleep(1.0)
    w = sam.window(t0, time.perf_counter())
    print("# idle: %.0f W, sclk %.0f MHz" % (np.nanmean(w[:, 1]), np.nanmean(w[:, 2])), flush=True)

    # stand-alone kernel times (serial schedule, HIP events)
    engine.set_overlap(0)
    for _ in range(3):
        op()
    torch.cuda.synchronize()
    kms = np.array(ls.last_kernel_ms())
    print("# serial HIP-event ms (prep, far field, wings, zones, -): %s" % np.round(kms, 3), flush=True)
    names = ["sr_prep_kernel", "sr_farfield_kernel (level-0 pass)", "sr_s2m_kernel + sr_m2m_kernel x4", "sr_m2l_kernel",
             "sr_abscoeff_near_zones_kernel", "sr_abscoeff_near_wings_kernel"]
    guess_ms = [0.27, 0.45, 0.40, 0.20, 3.6, 1.05]
    rows = []

    def measure(name, fn, n_launch, reps):
        """fn() enqueues `n_launch` launches of the thing measured (plus, for the kernel loops, one serial op around them);
        repeated `reps` times; the window excludes the first 0.4 s (clock / power ramp)."""
        torch.cuda.synchronize()
        time.sleep(0.3)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        w = sam.window(t0 + 0.4, t1 - 0.02)
        dt = t1 - t0
        p, f = (np.nanmean(w[:, 1]), np.nanmean(w[:, 2])) if len(w) else (float("nan"), float("nan"))
        e_cnt = float("nan")
        wa = sam.window(t0, t1)
        if len(wa) > 2 and np.isfinite(wa[:, 3]).all():
            e_cnt = (wa[-1, 3] - wa[0, 3]) / max(wa[-1, 0] - wa[0, 0], 1e-9) * dt     # J over the loop (counter slope x time)
        n = n_launch * reps
        rows.append((name, n, dt / n * 1e3, p, f, p * dt / n, e_cnt / n, len(w)))
        print("%-40s %7d launches  %8.4f ms  %7.1f W  %6.0f MHz  %8.4f J/launch (P x t)  %8.4f J/launch (counter)  [%d samples]"
              % rows[-1], flush=True)

    for k in range(6):
        n = int(min(20000, max(50, args.seconds / (guess_ms[k] * 1e-3))))
        assert lib.sr_set_kernel_repeat(k, n) == 0
        # the one serial op around the loop (5.3 ms) is part of the window: < 1 % for the long loops
        measure(names[k] + " alone", op, n, 1)
    assert lib.sr_set_kernel_repeat(-1, 1) == 0
    n_steps = int(args.seconds / 5.5e-3)
    measure("serial op (sr_set_overlap(0))", op, 1, n_steps)
    engine.set_overlap(1)
    engine.set_timing(0)
    for _ in range(5):
        step()
    measure("pipelined step (columns + op + recursion)", step, 1, n_steps)
    measure("pipelined op only", op, 1, n_steps)
    engine.set_timing(1)
    sam.stop_flag = True
    # the sum of the kernels' Joules against the step's
    kj = sum(r[5] for r in rows[:6])
    print("# sum of the six kernels alone: %.3f J, %.3f ms;  serial op %.3f J, %.3f ms;  pipelined step %.3f J, %.3f ms"
          % (kj, sum(r[2] for r in rows[:6]), rows[6][5], rows[6][2], rows[7][5], rows[7][2]), flush=True)


if __name__ == "__main__":
    main()
