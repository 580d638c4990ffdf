#!/usr/bin/env python3
"""bench.py -- limb spectra/s of the SpectRobot spectral hot path on MI355X.

A "step" is one pass of the hot path over one synthetic limb case: per-layer
abs/emi coefficient spectra (prep + gather kernels), the LOS columns and radiance
recursion of the ray batch and, for N > 1, the all-gather of the spectral shards.
Default workload = BASELINE.json configs[1]: CH4 Titan limb, 1e5 lines x 1e5 nu-grid
x 80 layers, 1 ray, fp64.  Inputs are uploaded before the timed region (HBM resident).

  python bench.py --gpus N --steps K --warmup W     N > 1 without WORLD_SIZE: starts its own N ranks (launch_ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (the same ranks, from torchrun)
  python bench.py --config 2|3|4|lut  extra lines for BASELINE configs[2..4] and the look-up-table build (bench_configs.py;
                                      not the headline)

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline`
(dominant kernel: EXECUTED flops, counted on the device, over its HIP-event time)
and `cpu_baseline` (the oracle, i.e. the CPU restatement of the reference, on the
host cores; N = 1 only).
"""
import argparse
import gc
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# six streams per coefficient op: give them hardware queues of their own (see spectrobot_amd/__init__.py; set before
# anything initialises HIP)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X fp64 vector (= fp64 matrix) dense peak, SURVEY 8-d / AMD datasheet
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)
FLOP_PER_EVAL = 16             # SURVEY 8-d flop model per (line, layer, grid point), brute force
# Executed-work flop model.  Per evaluation, SURVEY 8-d: region 1 = 9, region 2 = 17,
# core = 140, + 6 for the weighted accumulations; region 3 (no exp / cos, degree 4 / 5 instead of 6 / 7: counted from
# core_region3: rx 8, two polynomials 9 x 2 + 11 x 2, quotient 12, weights 6) = 70, not the core's 146.  Per far-field (line, box) expansion: counted from
# sr_farfield_kernel's source (setup 42, reciprocal 9, f0..f3 14, C = degree + 1 coefficients x 2 outputs).  Per
# (point, level) polynomial: 2 outputs x Horner of the degree.  Per window-end expansion: series + its share
# of the lane scan.
# Box-pair far field (default): per (line, side) multipole expansion -- since round 4 ONE convolution per line serves
# both sides (per line, at degree 22 = 21 moment orders: Laurent series 52, two anchors 28, powers 40, convolution 121 fma,
# two weighted accumulations 42 fma, first-order anchor corrections 8 fma + 4 = 466; 430 per side before); per (source box,
# target box, layer) translation Q x C x 2 outputs fma (the MFMA tiles execute padded ones: padding not counted); the short
# series of the window-band lines in the level-0 pass are not counted at all.
# The far field's numbers follow the expansion degree the library was built with (flop_model; degree 22 until the end of
# round 6: 281 / 93 / 233 / 1932; degree 19 since: 253 / 81 / 188 / 1440).
ASYNC_GATHER = os.environ.get("SR_GATHER_ASYNC", "1") != "0"   # SR_GATHER_ASYNC=0: every step waits for its all-gather


def flop_model(degree=22):
    C, Q = degree + 1, degree - 1
    per_line = 52.0 * Q / 21 + 28 + 40.0 * Q / 21 + 242.0 * (Q / 21.0) ** 2 + 84.0 * Q / 21 + 20
    return {"region1_evals": 15, "region2_evals": 23, "region3_evals": 70, "region4_evals": 146,
            "farfield_expansions": int(round(65 + 216.0 * C / 23)), "poly_point_levels": 5 + 4 * degree, "window_end_expansions": 187,
            "multipole_line_sides": int(round(per_line / 2)), "box_pair_translations": 4 * Q * C}


FLOP = flop_model(22)          # (set from the loaded library in main(): engine.far_field_degree())
KERNEL_COUNTERS = {
    "sr_farfield_kernel": ("farfield_expansions", "multipole_line_sides", "box_pair_translations"),
    "sr_abscoeff_near_wings_kernel": ("region1_evals", "window_end_expansions", "poly_point_levels"),
    "sr_abscoeff_near_zones_kernel": ("region2_evals", "region3_evals", "region4_evals"),
}


def kernel_sources_sha256():
    """Hash of the kernel sources: a PMC traffic figure is only reported for the build it was measured on."""
    h = hashlib.sha256()
    for f in ("sr_kernels.hip", "sr_device.hpp", "sr_kernels.hpp"):     # the kernels; the host API does not change their traffic
        h.update(open(os.path.join(ROOT, "spectrobot_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def host_cores():
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # the GPU box gives each job a CPU share through the cgroup quota (16 cores for one GPU)
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(round(int(quota) / int(period)))))
    except Exception:
        pass
    return cores


def reference_kernel_timing():
    """The reference's own compiled humliv_bb (oracle/_ref, built from the reference's Fortran by
    oracle/Makefile), ONE pinned core, 2000 calls after 200 warm-up calls, median and min.  Run before the
    threaded oracle leg so that its threads do not disturb it."""
    try:
        from oracle import ref_fortran as RF
        if not RF.available():
            return {"available": False}
        aff = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
        try:
            if aff:
                os.sched_setaffinity(0, {sorted(aff)[len(aff) // 2]})
            out = RF.time_humliv_bb()
        finally:
            if aff:
                os.sched_setaffinity(0, aff)
        out["available"] = True
        out["cores"] = 1
        return out
    except Exception as e:  # the reference build is optional on the box
        return {"available": False, "note": str(e)[:100]}


def serial_kernel_times_and_counts(engine, ls, step, n=5):
    """Stand-alone HIP-event durations of the coefficient kernels (sr_set_overlap(0): one after the other on the
    caller's stream, average of n launches of `step`) and the work they execute (one pass of the counting
    instantiations).  `step` must run exactly one coefficient op on `ls` last."""
    engine.set_overlap(0)
    step()
    serial_kms = np.zeros(5)
    for _ in range(n):
        step()
        serial_kms += np.array(ls.last_kernel_ms()) / n
    engine.set_counting(1)
    step()
    counts = ls.last_eval_counts()
    engine.set_counting(0)
    engine.set_overlap(1)
    return serial_kms, counts


def coefficient_roofline(serial_kms, counts, far_field=3):
    """roofline object of the dominant coefficient kernel from its stand-alone duration and executed work."""
    far_name = ("sr_farfield_kernel" if far_field == 1 else
                "sr_farfield_kernel (level 0) + sr_s2m_kernel + sr_m2m_kernel + sr_m2l_kernel")
    names = [far_name, "sr_abscoeff_near_wings_kernel", "sr_abscoeff_near_zones_kernel"]
    per_kernel = {}
    for nm, ms in zip(names, serial_kms[1:4]):
        fl = float(sum(FLOP[c] * counts[c] for c in KERNEL_COUNTERS[nm.split(" ")[0]]))
        per_kernel[nm] = {"ms": float(ms), "executed_flops": fl,
                          "achieved_tflops": fl / (ms * 1e-3) / 1e12 if ms > 0 else None,
                          "frac": fl / (ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS if ms > 0 else None}
    dom = max(names, key=lambda n: per_kernel[n]["ms"])
    d = per_kernel[dom]
    return {"bound": "fp64-valu", "achieved": d["achieved_tflops"], "peak": FP64_VALU_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": d["frac"], "traffic": None,
            "kernel": dom, "kernel_ms": d["ms"], "flops_per_launch": d["executed_flops"],
            "executed_counts": counts, "flop_model": FLOP, "kernels": per_kernel,
            "sr_prep_kernel_ms": float(serial_kms[0]),
            "mode": "far-field (box pairs)" if far_field in (2, 3) else "far-field (per line)"}


def level_tables_roofline(engine, ls, build, folded_op, n=3):
    """roofline object of the dominant kernel of a level-table build (the multi-channel pass): sr_zones_mc_kernel's
    executed flops / its stand-alone HIP-event duration.  Its evaluations are the folded zones kernel's of the FULL line
    list on the same rows (same ownership rules: counted by the folded op's counting instantiation, `folded_op`), each
    with one more weighted accumulation (three planes instead of two: + 2 flop); durations from a build under the serial
    schedule (sr_set_overlap(0): the far passes queue behind the zones kernel on the caller's stream)."""
    engine.set_overlap(0)
    folded_op()
    engine.set_counting(1)
    folded_op()
    counts = ls.last_eval_counts()
    engine.set_counting(0)
    build()
    ms = np.zeros(4)
    for _ in range(n):
        build()
        ms += np.array(ls.last_level_tables_ms()) / n
    engine.set_overlap(1)
    fz = float(sum((FLOP[c] + 2) * counts[c] for c in ("region2_evals", "region3_evals", "region4_evals")))
    fw = float((FLOP["region1_evals"] + 2) * counts["region1_evals"])
    return {"bound": "fp64-valu", "achieved": fz / (ms[1] * 1e-3) / 1e12, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": fz / (ms[1] * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS, "traffic": None, "kernel": "sr_zones_mc_kernel<256, 8>",
            "kernel_ms": float(ms[1]), "flops_per_launch": fz,
            "kernels": {"sr_prep_kernel (full list, three weights)": {"ms": float(ms[0])},
                        "sr_zones_mc_kernel": {"ms": float(ms[1]), "executed_flops": fz},
                        "far-only passes left after the zones kernel (serial schedule: all of them)": {"ms": float(ms[2])},
                        "sr_wings_mc_kernel": {"ms": float(ms[3]), "executed_flops_rows_only": fw,
                                               "note": "window ends run in rows here (the folded kernel scans them): their "
                                                       "evaluations and the far passes' polynomials are not in the count"}},
            "executed_counts": counts,
            "note": "one multi-channel build of all levels on these rows: every line evaluated once, three LDS adds per evaluation"}


def cpu_baseline(L, atm, grid, mm, e_lev, q_part, seconds_hint, n_layers_total, rays, gpu=None):
    """Oracle (kind 'port', mode 1 = direct accumulate) on all host cores over a bounded sample of the
    SAME workload: all lines, full grid, `ns` evenly spaced layers -- coefficients AND the radiance
    recursion of the ray batch over those layers."""
    from oracle import oracle as O
    ref = reference_kernel_timing()
    cores = host_cores()
    per_layer_core_s = 55e-6 * len(L["freq"])   # ~50 us per (line, layer) per core
    ns = int(max(1, min(n_layers_total, round(seconds_hint * cores / max(per_layer_core_s, 1e-9)))))
    ns = max(min(ns, n_layers_total), min(cores, n_layers_total))
    sel = np.linspace(0, n_layers_total - 1, ns).round().astype(int)
    tv = None if atm["tvib"] is None else atm["tvib"][:, sel]
    offs, lays, cols = rays
    t0 = time.time()
    # the checker's Q(T) is the oracle's own (fixture table of the reference's Fortran), not the product's: q_part is
    # what the product was given and is ignored here
    q_or = O.partition_sums(6, 1, atm["temps"][sel])
    abo, emo = O.abscoeff_layers(L, mm, e_lev, atm["temps"][sel], atm["press"][sel], q_or, tv, grid, mode=1,
                                 n_threads=cores)
    t_coef = time.time() - t0
    # the recursion over the sampled layers only (segments of the other layers skipped): same share of the work
    pos = {int(k): i for i, k in enumerate(sel)}
    t0 = time.time()
    for r in range(len(offs) - 1):
        sl = [s for s in range(offs[r], offs[r + 1]) if int(lays[s]) in pos]
        O.radiance_ray(abo, emo, [pos[int(lays[s])] for s in sl], cols[sl])
    t_rad = time.time() - t0
    dt = t_coef + t_rad
    out = {"value": (ns / n_layers_total) / dt * (len(offs) - 1), "unit": "spectra/s", "cores": cores, "kind": "port",
           "sample": "%d of %d layers (evenly spaced), all %d lines, full %d-point grid: coefficients %.1f s on %d "
                     "threads + radiance recursion of %d ray(s) over those layers %.2f s (1 thread); extrapolated "
                     "x%d/%d" % (ns, n_layers_total, len(L["freq"]), len(grid), t_coef, cores, len(offs) - 1, t_rad,
                                 n_layers_total, ns),
           "reference_humliv_bb": ref}
    if ref.get("available"):
        # the reference's own kernel beside the port: n_lines * n_layers calls of humliv_bb, nothing else
        pairs = float(len(L["freq"])) * n_layers_total
        out["reference_humliv_bb"]["kernel_only_spectra_per_s_all_cores_ideal"] = cores / (pairs * ref["median_us"] * 1e-6)
    if gpu is not None:  # the checker: the timed GPU result against the oracle at the BASELINE size
        ab, em = (t[sel].cpu().numpy() for t in gpu)
        out["parity_vs_oracle"] = {
            "layers": int(ns), "points": int(abo.size),
            "max_rel_err_abs": float(np.max(np.abs(ab - abo) / np.abs(abo))),
            "max_rel_err_emi": float(np.max(np.abs(em - emo) / np.abs(emo))),
            "note": "coefficient spectra of the timed far-field run vs the CPU oracle on the sampled layers, "
                    "full grid, all lines; requirement 1e-6 (the pytest of the same comparison: "
                    "tests/test_gpu_configs.py::test_config1_full_size_vs_oracle)"}
    return out


def build_rays(syn, engine, atm, n_rays, vmr=0.0148):
    """Tangent heights z_t = 100 + 12.5 r km: the LOS batch of the device pipeline -- per
    segment the layer and the LOS sample points (x, n, vmr) over which curgod_fort_2 gives the column."""
    nd = syn.number_density(atm["press"], atm["temps"])
    L = syn.limb_los(atm["z"], nd, [np.full(len(atm["z"]), vmr)], 100.0 + 12.5 * np.arange(n_rays) + 1e-3)
    los = engine.LimbLOS(L["seg_off"], L["seg_layer"], L["pt_off"], L["x"], L["nd"], L["vmr"], col_scale=[syn.CH4_ISO_RATIO])
    return los, L


def cpu_columns(L, scale):
    """The same columns on the host for the CPU leg (oracle's curgod_fort_2)."""
    from oracle import oracle as O
    return np.array([scale * O.curgod(2, L["nd"][a:b], L["x"][a:b], vmr=L["vmr"][0][a:b])
                     for a, b in zip(L["pt_off"][:-1], L["pt_off"][1:])])


def launch_ranks(n, argv=None, build=True):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start the N ranks as fresh child processes of
    this script (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torchrun would) and return the job's exit
    code.  The reference fans out in-process too (spect_main_module.py:2746-2767: one Process per LOS; :1619-1630,
    2814-2818: spectral splits) -- its callers never wrap it in a launcher.  This process has not imported torch nor
    touched the GPU and never does: it only builds the library (hipcc), waits, forwards rank 0's JSON line (the children
    inherit stdout; only rank 0 prints) and, when a rank dies, ends the others -- by their own PIDs.
    argv: the child's command line after the interpreter (default: this script with this process's arguments)."""
    import signal
    import socket
    import subprocess
    import __graft_entry__
    if build:
        __graft_entry__.ensure_built(builder=True)       # once, before any rank starts
    argv = [os.path.abspath(__file__)] + sys.argv[1:] if argv is None else list(argv)
    with socket.socket() as s:                           # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SR_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable] + argv, env=env))
    rc = 0
    try:
        live = list(procs)
        while live:
            time.sleep(0.05)
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:                # one rank failed: the others would wait in a collective for ever
                    rc = code if code > 0 else 1
                    for q in live:
                        q.terminate()
                    t_end = time.time() + 10.0
                    while any(q.poll() is None for q in live) and time.time() < t_end:
                        time.sleep(0.05)
                    for q in live:
                        if q.poll() is None:
                            q.kill()
    except BaseException:
        for p in procs:
            if p.poll() is None:
                p.send_signal(signal.SIGTERM)
        raise
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="1", help="1 (default): BASELINE configs[1], the headline; "
                    "2, 3, 4: extra lines for configs[2..4]; lut: the look-up-table / G-coefficient build, the one thing "
                    "the reference publishes a figure for (bench_configs.py)")
    ap.add_argument("--lines", type=int, default=100000)
    ap.add_argument("--grid", type=int, default=100000)
    ap.add_argument("--layers", type=int, default=80)
    ap.add_argument("--rays", type=int, default=1)
    ap.add_argument("--ppl", type=int, default=8, help="grid points per lane in the coefficient kernels")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--shard", default="", help="R/W: time only the spectral shard of rank R of W on this GPU "
                    "(tuning aid for the multi-GPU shard size; not a bench line of the metric)")
    ap.add_argument("--3d", dest="three_d", action="store_true", help="--config 3 in its 3-D form: a coefficient row per LOS "
                    "step (P, T, T_vib at the local SZA along the path), Jacobians per altitude layer")
    ap.add_argument("--level-route", type=int, default=1, choices=(0, 1), help="--config 3 / lut: level tables by the multi-channel "
                    "pass (1, default: every line once) or by one coefficient op per level (0: the route of rounds 4-5)")
    ap.add_argument("--balanced", action="store_true", help="N > 1: shards of equal line-window WORK (distributed.shard_bounds_balanced) "
                    "instead of equal width; the gather then pads to the widest shard")
    ap.add_argument("--exact", action="store_true", help="evaluate every (line, point) exactly (no far-field expansions)")
    ap.add_argument("--far-field", type=int, default=3, choices=(1, 2, 3), help="3 (default): far-field expansions from box "
                    "pairs (multipole -> local), sparse line sets per line and box; 2: box pairs always; 1: per line and box")
    args = ap.parse_args()
    args.config = args.config if args.config == "lut" else int(args.config)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:   # the driver's form: python bench.py --gpus N
        sys.exit(launch_ranks(args.gpus))

    import __graft_entry__
    __graft_entry__.ensure_built(builder=int(os.environ.get("LOCAL_RANK", "0")) == 0)  # fresh checkouts carry no library
    if args.config != 1:
        import bench_configs
        return bench_configs.main(args)
    import torch
    from spectrobot_amd import engine, synthetic as syn, distributed as sd
    from spectrobot_amd._lib import lib, dp

    rank, local, world = sd.init_from_env()
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)
    engine.set_device(local % max(torch.cuda.device_count(), 1))
    FLOP.update(flop_model(engine.far_field_degree()))
    engine.set_points_per_lane(args.ppl)
    engine.set_far_field(0 if args.exact else args.far_field)
    info = engine.device_info()

    # ---- synthetic workload, identical on every rank ----
    n_lev = 12
    grid = syn.make_grid(2975.0, 5e-4, args.grid)
    L = syn.make_lines(args.lines, grid, config_id=2, n_levels=n_lev)
    atm = syn.make_atmosphere(args.layers, n_lev)
    e_lev = syn.CH4_LEVEL_ENERGIES
    q_part = np.zeros(args.layers)
    tt = np.ascontiguousarray(atm["temps"])
    assert lib.sr_calc_partition_sum(6, 1, tt.ctypes.data_as(dp), args.layers, q_part.ctypes.data_as(dp)) == 0
    los, Lr = build_rays(syn, engine, atm, args.rays)

    ls = engine.LineSet(L, grid, 6, 1, syn.CH4_MM, e_lev)     # lines -> HBM (outside the timed region)
    # the ranks' spectral shards: equal width (default: one in-place all-gather) or equal modelled work (--balanced)
    bounds = [sd.shard_bounds(args.grid, world, r) for r in range(world)]
    if args.balanced and world > 1:
        bounds = sd.shard_bounds_balanced(L["freq"], grid, world)
    g_lo, g_hi = bounds[rank]
    if args.shard:
        assert world == 1
        g_lo, g_hi = sd.shard_bounds(args.grid, int(args.shard.split("/")[1]), int(args.shard.split("/")[0]))
    npts = g_hi - g_lo
    ab = torch.empty((args.layers, npts), dtype=torch.float64, device="cuda")
    em = torch.empty_like(ab)
    full = torch.empty((args.rays, args.grid), dtype=torch.float64, device="cuda") if world > 1 else None

    def step():
        # The LOS columns (curgod_fort_2 per segment) integrated on the device from the resident batch's sample points --
        # part of the step as in every round; what the resident batch saves is re-staging an unchanged batch from the
        # host --, then coefficient op + recursion in one library call (sr_limb_step_dev).
        los.refresh_columns()
        _, _, rad = ls.limb_step(atm["temps"], atm["press"], los, tvib=atm["tvib"], q_part=q_part, g_lo=g_lo, g_hi=g_hi,
                                 out=(ab, em))
        if args.shard:
            return rad
        # async: the next step's kernels need not wait for this step's (latency-bound) gather; barrier() below
        # synchronises the device, collectives included, before any time is taken or `full` is read
        return sd.all_gather_spectrum(rad, args.grid, world, rank, out=full, bounds=bounds, async_op=ASYNC_GATHER)

    wait_s = [0.0]   # time this rank's host spent in wait_gathers() (the asynchronous gathers still outstanding at a barrier)

    def barrier():
        t_w = time.perf_counter()
        sd.wait_gathers()
        wait_s[0] += time.perf_counter() - t_w
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    los.handle(args.layers)       # the batch resident on the device (outside the timed region, like the line list)
    for _ in range(args.warmup):
        step()
    barrier()
    # A full collection of the Python garbage collector walks every object torch / numpy imported (~40 ms here):
    # one of those inside the timed region of a 1 ms step is a 40-step hiccup (tools/stall_probe.py found it at the
    # 60th step of every run).  Collect now and move what exists into the permanent generation.
    gc.collect()
    gc.freeze()
    kms = np.zeros(5)
    engine.set_timing(0)          # no per-kernel timing events in the timed steps (diagnostics: the pass below has them)
    step()
    barrier()
    wait_s[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):   # no host synchronisation inside: consecutive steps pipeline on the GPU
        spec = step()
    t_enq = time.perf_counter() - t0                       # host time to enqueue the timed steps
    barrier()
    elapsed = time.perf_counter() - t0
    wait_timed = wait_s[0]
    engine.set_timing(1)
    for _ in range(args.steps):   # HIP-event times of this rank's kernels (each query synchronises that step)
        step()
        kms += np.array(ls.last_kernel_ms())
    barrier()
    dist_rec = None
    if world > 1:
        dev_t = "cuda" if torch.distributed.get_backend() == "nccl" else "cpu"
        # every rank's own numbers, for rank 0's line (the first real SCALE run must be diagnosable from
        # its one line): elapsed, host enqueue time, host time in wait_gathers(), all over the timed steps
        mine = torch.tensor([elapsed, t_enq, wait_timed], dtype=torch.float64, device=dev_t)
        every = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(every, mine)
        per_rank = np.array([e.cpu().numpy() for e in every])
        elapsed = float(per_rank[:, 0].max())              # the bench contract: MAX over ranks
        costs, n_lines_rank = sd.shard_costs(L["freq"], grid, bounds)
        # what the collective really was, and the asynchronous gather of the last timed step against a blocking one
        # of the same shard (outside the timed region)
        dist_rec = dict(sd.dist_info(), async_gather=bool(ASYNC_GATHER), gathers=dict(sd.stats),
                        shards={"bounds": [list(b) for b in bounds], "balanced": bool(args.balanced),
                                "points": [hi - lo for lo, hi in bounds], "lines_prepared": n_lines_rank,
                                "model_cost_max_over_mean": float(max(costs) / (sum(costs) / len(costs)))},
                        per_rank={"ms_per_step": [float(v) / args.steps * 1e3 for v in per_rank[:, 0]],
                                  "ms_per_step_min": float(per_rank[:, 0].min()) / args.steps * 1e3,
                                  "ms_per_step_max": float(per_rank[:, 0].max()) / args.steps * 1e3,
                                  "host_enqueue_ms_per_step": [float(v) / args.steps * 1e3 for v in per_rank[:, 1]],
                                  "host_ms_in_wait_gathers": [float(v) * 1e3 for v in per_rank[:, 2]]},
                        rank0_ms_in_wait_gathers=float(per_rank[0, 2]) * 1e3)
        rad_chk = engine.limb_rays((ab, em), los, resident=False)   # (the per-call staging route: a second opinion)
        blocking = sd.all_gather_spectrum(rad_chk, args.grid, world, rank, bounds=bounds, async_op=False)
        torch.cuda.synchronize()
        dist_rec["async_equals_blocking"] = bool(torch.equal(blocking, spec))
    kms /= args.steps
    prep_ms = float(kms[0])
    main_ms = float(kms[1:].sum())   # far-field mode: the coefficient op as a whole (its kernels overlap)
    checksum = float(spec.sum().item())
    # outside the timed region: the kernels of the coefficient op one after the other (HIP events on the
    # stream they run on), and the work each one executes (counting instantiations, device counters)
    serial_kms = counts = None
    if not args.exact:
        serial_kms, counts = serial_kernel_times_and_counts(engine, ls, step)
    # outside the timed region: the brute-force kernels (every evaluation exact) for reference
    exact_kms = None
    if not args.exact and world == 1:
        engine.set_far_field(0)
        step()
        exact_kms = np.zeros(5)
        for _ in range(2):
            spec_x = step()
            exact_kms += np.array(ls.last_kernel_ms()) / 2
        exact_checksum = float(spec_x.sum().item())
        engine.set_far_field(args.far_field)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = args.steps / elapsed  # one limb spectrum (x rays) per step, whole job
        evals = float(args.lines) * 13010.0 * args.layers / world           # SURVEY 8-d, per launch
        flops_bf = FLOP_PER_EVAL * evals
        alg_bytes = (8.0 * args.grid * args.layers * 2 + 80.0 * args.lines) / world
        gbs = alg_bytes / (main_ms * 1e-3) / 1e9
        # HBM traffic from the PMC counters: only for the build it was measured on (sources hash)
        traffic, traffic_src = None, None
        pmc_name = next((f for f in ("r06_pmc_hbm.json", "r05_pmc_hbm.json", "r04_pmc_hbm.json", "r03_pmc_hbm.json")
                         if os.path.exists(os.path.join(ROOT, "profiles", f))), "r06_pmc_hbm.json")
        pmc = os.path.join(ROOT, "profiles", pmc_name)
        if os.path.exists(pmc) and world == 1 and args.lines == 100000 and args.grid == 100000 and not args.shard:
            try:
                pj = json.load(open(pmc))
                if pj.get("kernel_sources_sha256") == kernel_sources_sha256():
                    traffic = pj.get("coefficient_kernels_hbm_bytes_per_step")
                    traffic_src = {"file": "profiles/" + pmc_name, "profile_tag": pj.get("profile_tag"),
                                   "kernel_sources_sha256": pj.get("kernel_sources_sha256")[:16],
                                   "raw_counter_bytes": pj.get("coefficient_kernels_hbm_bytes_per_step_raw"),
                                   "note": "FETCH_SIZE doubled for vector reads (calibration: profiles/"
                                           "r03_fetch_calibration.txt), WRITE_SIZE as reported; per step of the "
                                           "coefficient kernels"}
                else:
                    traffic_src = {"file": "profiles/" + pmc_name, "stale": True,
                                   "note": "measured on other kernel sources than this build: not reported"}
            except Exception:
                traffic = None
        if args.exact:
            xm = float(kms[1] + kms[2])
            roofline = {"bound": "fp64-valu", "achieved": flops_bf / (xm * 1e-3) / 1e12, "peak": FP64_VALU_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": flops_bf / (xm * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                        "traffic": None, "kernel": "sr_abscoeff_wings_kernel + sr_abscoeff_cores_kernel",
                        "kernels_ms": dict(zip(["sr_prep_kernel", "sr_abscoeff_wings_kernel",
                                                "sr_abscoeff_cores_kernel"], [float(v) for v in kms[:3]])),
                        "flops_per_launch": flops_bf, "mode": "exact"}
        else:
            roofline = coefficient_roofline(serial_kms, counts, args.far_field)
            roofline.update({
                "traffic": traffic, "traffic_source": traffic_src,
                "coefficient_op_ms_in_timed_steps": main_ms,
                "note": "dominant kernel of the step; achieved = flops it EXECUTES (evaluations counted on the device "
                        "by the counting instantiations of the same kernels x the per-region flops of SURVEY 8-d; region 3 "
                        "at its own 70) / its average HIP-event duration over 5 launches with the kernels one after the "
                        "other on the caller's stream (in the timed steps the zones kernel overlaps the far-field kernel "
                        "on a second stream); peak 78.6 TFLOP/s = fp64 vector = fp64 MFMA dense peak of MI355X (no MFMA "
                        "use: not a contraction).  HBM is not the bound: arithmetic intensity ~1e3 flop/B"})
        out = {
            "metric": "limb spectra/sec (1e5 lines x 1e5 nu-grid, 80 layers)",
            "value": value * args.rays, "unit": "spectra/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "CH4 Titan limb (BASELINE configs[1]): %d lines x %d-pt grid x %d layers, "
                                   "%d ray(s), 12 non-LTE levels" % (args.lines, args.grid, args.layers, args.rays),
                       "n_lines": args.lines, "n_grid": args.grid, "n_layers": args.layers, "n_rays": args.rays,
                       "sharding": ("spectral window / %d, one all-gather per step (backend %s: nccl = RCCL over xGMI)"
                                    % (world, dist_rec["backend"]) if world > 1 else
                                    ("ONLY shard %s timed (tuning aid)" % args.shard if args.shard else "none")),
                       "mode": "exact" if args.exact else ("far-field, box pairs" if args.far_field in (2, 3) else "far-field, per line"),
                       "far_field": {"theta": 4, "degree": engine.far_field_degree(),
                                     "truncation_bound_rel_to_a_lines_own_contribution": engine.far_field_truncation_bound(),
                                     "note": "the degree follows the accuracy budget (required 1e-6; parity tests 1e-10); "
                                             "degree 22 = bound 2.6e-13 until the end of round 6 (-DSR_KFD=22)"},
                       "device": info["name"],
                       "cu_count": info["cu_count"]},
            "roofline": roofline,
            "roofline_hbm": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "bytes_per_launch": alg_bytes,
                             "note": "the BASELINE-named figure: algorithmic bytes 8*n_grid*n_layers*2 + 80*n_lines "
                                     "(SURVEY 8-d) / coefficient-op time; <<1 by construction (compute bound)"},
            "algorithmic_speedup_vs_brute_force": {
                "brute_force_flops": flops_bf, "equivalent_tflops": flops_bf / (main_ms * 1e-3) / 1e12,
                "ratio_to_fp64_peak": flops_bf / (main_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                "note": "16 flop x n_lines*13010*n_layers evaluations of the reference formulation / coefficient-op "
                        "time: NOT a roofline fraction -- in far-field mode ~94 % of those evaluations are replaced "
                        "by per-box Taylor expansions, so this exceeds what brute force could reach"},
            "checksum": checksum,
        }
        if dist_rec is not None:
            out["dist"] = dist_rec
        if exact_kms is not None:
            xm = float(exact_kms[1] + exact_kms[2])
            out["roofline_exact_mode"] = {
                "bound": "fp64-valu", "achieved": flops_bf / (xm * 1e-3) / 1e12, "peak": FP64_VALU_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": flops_bf / (xm * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                "kernel": "sr_abscoeff_wings_kernel + sr_abscoeff_cores_kernel (every (line, layer, point) "
                          "evaluated; same results to ~1e-13)",
                "kernels_ms": {"sr_prep_kernel": float(exact_kms[0]), "sr_abscoeff_wings_kernel": float(exact_kms[1]),
                               "sr_abscoeff_cores_kernel": float(exact_kms[2])},
                "checksum": exact_checksum}
        if world == 1 and args.cpu_seconds > 0:
            gpu = None
            if not args.shard:  # ab / em hold the last exact-mode step: recompute the timed mode's result
                step()
                torch.cuda.synchronize()
                gpu = (ab, em)
            out["cpu_baseline"] = cpu_baseline(L, atm, grid, syn.CH4_MM, e_lev, q_part, args.cpu_seconds, args.layers,
                                               (Lr["seg_off"], Lr["seg_layer"], cpu_columns(Lr, syn.CH4_ISO_RATIO)),
                                               gpu=gpu)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
            # the shipped Python path, quoted separately.  2.3 ms per (line, layer)
            # per core was measured in the survey container, not on this box; ideal
            # scaling over this box's cores is assumed.
            pairs = float(args.lines) * args.layers
            py_sps = out["cpu_baseline"]["cores"] / (pairs * 2.3e-3)
            out["cpu_baseline"]["reference_python_path_extrapolated"] = {
                "value": py_sps, "unit": "spectra/s", "speedup": out["value"] / py_sps,
                "note": "shape + pickle + accumulate = 2.3 ms per (line, layer) per core (BASELINE.md 2, survey "
                        "container), x ideal scaling over the cores above"}
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
