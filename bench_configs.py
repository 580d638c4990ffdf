#!/usr/bin/env python3
"""Workloads of BASELINE.json configs[2..4] (synthetic inputs of SURVEY 8-d) -- shared by
tests/test_gpu_configs.py and `bench.py --config N`, which prints ONE extra JSON line per config (not the
headline: that is configs[1], bench.py's default).

  config 2  CH4 Titan limb, 1e5 lines x 1e5 grid x 80 layers, 64 tangent-height rays (z_t = 100 + 12.5 r km)
            batched on one coefficient op, spectral window sharded over the ranks, one all-gather
  config 3  radtran_3Dvs2D_sza30-80 shape: 8 solar zenith angles = 8 independent atmospheres (vibrational
            temperatures follow the illumination) x a set of 8 rays, 2e5-point grid, 2e5 lines, per-layer
            Jacobians w.r.t. T (finite differences of the coefficient op, DT_SCHEME below, + the recursion's
            sensitivity) and w.r.t. the VMR of every layer (analytic)
  config 4  retrieval loop: HCN (mol 23) + CH4 on one grid, VIMS-like bands, Gauss-Newton /
            Levenberg-Marquardt over up to 20 iterations with the reference's stopping rule; a step is one
            iteration (forward model + Jacobians of all LOS + algebra)
"""
import json
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
HCN_MM = 27.010899          # molparam.txt HCN 124
HCN_ISO_RATIO = 0.985114
HCN_LEVEL_ENERGIES = np.array([0., 711.98, 1411.41, 2096.85, 3311.48, 4004.17])   # HCN-like level ladder, cm^-1


# d/dT of the coefficients in configs[3]: engine.coefficients_dT.  "forward": two coefficient ops per set -- the
# quotient (c(T + 0.002 K) - c(T)) / 0.002 K with the region boundaries of the perturbed op frozen at T, 3e-4..5e-4 of a
# layer's largest derivative from the frozen central reference, i.e. the accuracy class of rounds 1-3's three-op central
# difference with moving boundaries (2e-4..3e-4: tests/test_gpu_configs.py::test_temperature_derivative_schemes);
# SR_DT_SCHEME=central: three ops, frozen, 3e-5..5e-5.  The line reports the other scheme's time beside its own.
import os as _os
DT_SCHEME = _os.environ.get("SR_DT_SCHEME", "forward")
DT_SCHEME_NOTE = ("forward difference of the coefficient op, 0.002 K, region boundaries frozen at T: two ops per set"
                  if DT_SCHEME == "forward" else
                  "central differences of the coefficient op, +-0.05 K, region boundaries frozen at T: three ops per set")

def ch4_case(n_lines, n_grid, n_layers=80, n_levels=12, config_id=2, w0=2975.0):
    """SURVEY 8-d CH4 case: grid, lines, atmosphere, level energies."""
    from spectrobot_amd import synthetic as syn
    grid = syn.make_grid(w0, 5e-4, n_grid)
    L = syn.make_lines(n_lines, grid, config_id=config_id, n_levels=n_levels)
    atm = syn.make_atmosphere(n_layers, n_levels)
    atm["nd"] = syn.number_density(atm["press"], atm["temps"])
    return grid, L, atm, syn.CH4_LEVEL_ENERGIES[:n_levels]


def tangent_heights(n_rays):
    return 100.0 + 12.5 * np.arange(n_rays) + 1e-3      # z_t = 100 + 12.5 r km


# 3-D atmosphere of configs[3], modelled as the reference's (radtran_3Dvs2D_sza30-80_test.py:66-116): kinetic
# temperature and pressure on (latitude box, altitude) -- AtmProfile(grid, TT, 'temp', ['box', 'lin']) on the seven
# boxes of lat_ext -- and only the vibrational temperatures by the local illumination.  (Rounds 1-3 made the kinetic
# temperature a continuous function of the SZA too, which no two LOS steps share and the reference does not do.)
LAT_BOX_DT = np.array([-6.0, -4.0, -2.0, 0.0, 1.5, -3.0, -7.0])      # K, offset of the box's kinetic profile
TANGENT_LAT_DEG = 5.0        # the pixels' tangent points: one latitude band, the SZA varies with local time
SUBSOLAR_LAT_DEG = -12.0     # Titan, 2006-07


def vib_temps(atm, temps, alt_layer, mu):
    """T_vib [n_levels, n] at altitude shells alt_layer for cos SZA mu: the excess over the kinetic temperature scales
    with the illumination (clipped at the terminator)."""
    mu = np.clip(np.asarray(mu, float), 0.0, 1.0)
    exc = (atm["tvib"] - atm["temps"][None, :])[:, alt_layer]
    return np.asarray(temps, float)[None, :] + exc * (0.4 + 1.2 * mu)[None, :]


def sza_atmosphere(atm, sza_deg, n_levels=12, box=3):
    """The 1-D column of one pixel (use_tangent_sza = True): the kinetic profile of its latitude box (all eight SZA
    pixels sit in the equatorial box: they SHARE (P, T)), vibrational temperatures at the tangent point's SZA."""
    mu = np.cos(np.deg2rad(sza_deg))
    out = dict(atm)
    out["temps"] = atm["temps"] + LAT_BOX_DT[box]
    k = np.arange(len(atm["temps"]))
    out["tvib"] = vib_temps(atm, out["temps"], k, np.full(len(k), mu))
    return out


def step_atmosphere(atm, seg_alt_layer, seg_mu, seg_box=None):
    """(T, P, T_vib) of every LOS step of a 3-D path: the kinetic state of the step's (latitude box, altitude shell),
    the vibrational temperatures of the step's own illumination."""
    k = np.asarray(seg_alt_layer)
    box = np.full(len(k), 3) if seg_box is None else np.asarray(seg_box)
    temps = atm["temps"][k] + LAT_BOX_DT[box]
    return dict(temps=temps, press=atm["press"][k], tvib=vib_temps(atm, temps, k, seg_mu))


def los_3d_set(atm, vm, tz, sza, az):
    """One set of 3-D limb rays (tangent heights tz, headings az east of north, tangent points at TANGENT_LAT_DEG seeing
    the sun at `sza`) with the state of every step."""
    from spectrobot_amd import geometry as geo
    Lr = geo.limb_los_3d(atm["z"], atm["nd"], [vm], tz, sza, az, tangent_lat_deg=TANGENT_LAT_DEG, subsolar_lat_deg=SUBSOLAR_LAT_DEG)
    Lr["seg_box"] = geo.lat_box_index(Lr["seg_lat"])
    Lr["state"] = step_atmosphere(atm, Lr["seg_alt_layer"], Lr["seg_mu"], Lr["seg_box"])
    return Lr


ROUTE = _os.environ.get("SR_CONFIG3_ROUTE", "factored")     # "direct": a folded coefficient op per set (rounds 1-3)
DT_FACTORED = float(os.environ.get("SR_DT_FACTORED", "0.02"))   # K: the T + dT tables of the level-factored route (linearised weights): error of the radiance Jacobian against central differences 5.8e-4 / 3.5e-4 / 8.0e-4 / 1.5e-3 at 0.01 / 0.02 / 0.05 / 0.1 K


def layer_vmr_weights(z, alt):
    """par_w [n_layers, n_pt]: triangular weight of every altitude level at the LOS sample altitudes (the VMR
    profile is piecewise linear on the levels: vmr(alt) = sum_k w_k(alt) vmr_k)."""
    top = z[-1] + (z[-1] - z[-2])
    zz = np.append(z, top)
    W = np.zeros((len(z), len(alt)))
    for k in range(len(z)):
        m = np.zeros(len(zz))
        m[k] = 1.0
        if k == len(z) - 1:
            m[-1] = 1.0          # the profile is continued with its last value above the top level
        W[k] = np.interp(alt, zz, m)
    return W


def two_gas_scene(n_lines_ch4, n_lines_hcn, n_grid, n_layers, n_bands=14, seed=0, w0=3290.0):
    """HCN + CH4 on one grid around the HCN nu3 / CH4 band overlap, VIMS-like Gaussian bands."""
    from spectrobot_amd import engine, synthetic as syn, retrieval
    grid = syn.make_grid(w0, 5e-4, n_grid)
    Lc = syn.make_lines(n_lines_ch4, grid, config_id=4, n_levels=12)
    Lh = syn.make_lines(n_lines_hcn, grid, config_id=5, n_levels=6)
    Lh["a_coeff"] = Lh["a_coeff"] * 30.0
    atm = syn.make_atmosphere(n_layers, 12)
    z = atm["z"]
    tv_h = np.array([atm["temps"] + (0.0 if L == 0 else 25.0 * (1.0 - np.exp(-(z - 100.0) / 250.0))) for L in range(6)])
    ch4 = retrieval.Gas("CH4", engine.LineSet(Lc, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES),
                        np.full(n_layers, 1.48e-4), syn.CH4_ISO_RATIO, tvib=atm["tvib"])
    hcn = retrieval.Gas("HCN", engine.LineSet(Lh, grid, 23, 1, HCN_MM, HCN_LEVEL_ENERGIES),
                        np.full(n_layers, 2e-6), HCN_ISO_RATIO, tvib=tv_h)
    lam = np.linspace(1e7 / grid[-1] + 1.2, 1e7 / grid[0] - 1.2, n_bands)
    scene = retrieval.LimbScene(grid, z, atm["temps"], atm["press"], [ch4, hcn], lam, np.full(n_bands, 1.1))
    return scene


def retrieval_problem(scene, n_pix=6, seed=1, noise_frac=0.004):
    """Truth, a priori, BayesSet and noisy synthetic observations of `scene` (two gases, 4 + 3 altitude nodes)."""
    from spectrobot_amd import spect_main_module as smm, retrieval
    rng = np.random.default_rng(seed)
    z = scene.z
    span = z[-1] - z[0]
    nodes = {"CH4": [z[0] + f * span for f in (0.06, 0.3, 0.55, 0.85)], "HCN": [z[0] + f * span for f in (0.1, 0.45, 0.8)]}
    truth = {"CH4": np.array([1.7e-4, 1.5e-4, 1.25e-4, 1.0e-4]), "HCN": np.array([1.2e-6, 2.6e-6, 4.0e-6])}
    apr = {"CH4": np.full(4, 1.3e-4), "HCN": np.full(3, 2.2e-6)}
    pixels = [retrieval.LimbPixel(z[0] + (0.08 + 0.13 * i) * span, fov_half=0.02 * span, pixel_rot=10.0 * (i % 3))
              for i in range(n_pix)]

    def bayes(values):
        bs = smm.BayesSet(tag="HCN+CH4 limb")
        for name in ("CH4", "HCN"):
            bs.add_set(smm.LinearProfile_1D_new(name, z, nodes[name], apr[name], 0.5 * apr[name], first_guess_prof=values[name]))
        return bs

    bs_true = bayes(truth)
    for name in ("CH4", "HCN"):
        scene.gas(name).add_clim(bs_true.sets[name].profile())
    y_true = retrieval.radtrans(scene, pixels)
    for pix, y in zip(pixels, y_true):
        sig = noise_frac * np.abs(y.spectrum).max() * np.ones_like(y.spectrum)
        pix.observation = retrieval.Spectrum(y.spectrum + sig * rng.standard_normal(sig.size), scene.bands_nm)
        pix.noise = retrieval.Spectrum(sig, scene.bands_nm)
    x_true = np.concatenate([truth["CH4"], truth["HCN"]])
    return bayes(apr), pixels, x_true


def _sync_time(fn, steps, warmup):
    import torch
    import gc
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    # as bench.py: a full collection of Python's garbage collector walks ~1e6 objects (40 ms) -- inside 20 timed steps
    # that is 2 ms per step (it hit --config 2 this round: 8.1 instead of 5.9 ms); collect now, freeze what exists
    gc.collect()
    gc.freeze()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, out


def config3_3d(args, rank, world, info, base):
    """configs[3] in its 3-D form (radtran_3Dvs2D_sza30-80_test.py:353-379: use_tangent_sza = False,
    invert_LOS_direction = True): per ray and per LOS STEP its own state.  Level-factored route (default): the pair
    tables once per step of 8 sets on the distinct (latitude box, altitude) rows the rays touch, at T and T + dT, then
    one combine per set for the ~900 steps' coefficient rows and their T derivatives, then the one-pass Jacobians
    (temperature per altitude layer through seg_jac_row, VMR per level).  SR_CONFIG3_ROUTE=direct: a folded
    coefficient op over the steps themselves, twice per set (rounds 1-3)."""
    import torch
    from spectrobot_amd import engine, synthetic as syn
    n_grid, n_lines, n_rays, szas = 200000, 200000, 8, [30.0, 37.0, 44.0, 51.0, 58.0, 65.0, 72.0, 80.0]
    if args.grid != 100000:
        n_grid = n_lines = args.grid      # reduced sizes for tests
    grid, L, atm, e_lev = ch4_case(n_lines, n_grid, args.layers, config_id=3, w0=2950.0)
    ls = engine.LineSet(L, grid, 6, 1, syn.CH4_MM, e_lev)
    vm = np.full(args.layers, 0.0148)
    tz = 120.0 + 60.0 * np.arange(n_rays)
    az = 22.5 * np.arange(n_rays)                       # the ray set fans out in azimuth
    my = szas[rank::world]
    pg = np.zeros(args.layers, np.int32)
    sets = [los_3d_set(atm, vm, tz, sza, az) for sza in my]
    for Lr in sets:
        Lr["los"] = engine.LimbLOS(Lr["seg_off"], Lr["seg_layer"], Lr["pt_off"], Lr["x"], Lr["nd"], Lr["vmr"], col_scale=[syn.CH4_ISO_RATIO])
        Lr["W"] = layer_vmr_weights(atm["z"], Lr["alt"])
    n_steps = len(sets[0]["seg_layer"])
    # the distinct (P, T) rows of all sets
    allT = np.concatenate([Lr["state"]["temps"] for Lr in sets])
    allP = np.concatenate([Lr["state"]["press"] for Lr in sets])
    T_rows, P_rows, step_row = engine.LevelFactored.unique_rows(allT, allP)
    at = 0
    for Lr in sets:
        Lr["row"] = step_row[at:at + len(Lr["seg_layer"])]
        at += len(Lr["seg_layer"])
    tv_all = np.concatenate([Lr["state"]["tvib"] for Lr in sets], axis=1)
    timing = {}

    def jacobians(Lr, co, dco):
        return engine.limb_rays_jacobians(co, Lr["los"], dcoeffs=dco, par_gas=pg, par_w=Lr["W"], seg_jac_row=Lr["seg_alt_layer"],
                                          n_jac_rows=args.layers)

    state = {}

    def step_factored():
        # the tables are REBUILT every step (a retrieval iteration moves the temperatures), in place
        lf = state["lf"].rebuild(T_rows, P_rows) if "lf" in state else state.setdefault("lf", engine.LevelFactored(ls, T_rows, P_rows, dT=DT_FACTORED))
        # ONE combine for the steps of all sets: the tables are read once (per set: once per set)
        (ca, ce), (da, de) = lf.steps(step_row, tvib=tv_all, derivative=True)
        res, at = None, 0
        for Lr in sets:
            n = len(Lr["seg_layer"])
            res = jacobians(Lr, (ca[at:at + n], ce[at:at + n]), (da[at:at + n], de[at:at + n]))
            at += n
        return res

    def step_direct():
        res = None
        for Lr in sets:
            a = Lr["state"]
            co, dco = engine.coefficients_dT(ls, a["temps"], a["press"], tvib=a["tvib"], scheme=DT_SCHEME)
            res = jacobians(Lr, co, dco)
        return res

    step = step_direct if ROUTE == "direct" else step_factored
    dt, res = _sync_time(step, max(1, args.steps // 10), min(args.warmup, 1))
    extra = {}
    if rank == 0 and ROUTE != "direct":
        # roofline of the route's own kernel, the combine: HBM-bound (tables read once per call + outputs written)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lf = engine.LevelFactored(ls, T_rows, P_rows, dT=DT_FACTORED)
        torch.cuda.synchronize()
        timing["tables_ms"] = (time.perf_counter() - t0) * 1e3
        Lr = sets[0]
        pop, dpop = ls.level_populations(T_rows[Lr["row"]], tvib=Lr["state"]["tvib"], derivative=True)
        outs = [torch.empty((len(Lr["row"]), n_grid), dtype=torch.float64, device="cuda") for _ in range(4)]
        used = len(np.unique(Lr["row"]))
        bytes_alg = 8.0 * n_grid * (2 * 2 * lf.tab.shape[0] * used + 4 * len(Lr["row"]))
        extra["roofline"] = _event_time(
            lambda: engine.glevel_combine(lf.tab, Lr["row"], pop, tab_dT=lf.tab_dT, dpop=dpop, dT=DT_FACTORED, out=outs),
            "sr_glevel_combine_kernel<12, true>", bytes_alg,
            "one set: the %d distinct (P, T) rows of its %d steps x 2 x 12 pair spectra x 2 tables read once + abs, emi, "
            "d abs / d T, d emi / d T of every step written" % (used, len(Lr["row"])))
        extra["roofline"]["traffic"] = None
        del lf, outs
        extra["tables"] = {"rows": int(len(T_rows)), "steps_all_sets": int(len(step_row)), "ms_per_build_T_and_T_plus_dT": timing.get("tables_ms"),
                           "note": "distinct (latitude box, altitude) rows of the %d sets' rays; each row's 24 pair spectra are "
                                   "built once at T and once at T + %.3f K and serve every step on it" % (len(sets), DT_FACTORED)}
        if world == 1 and args.cpu_seconds > 0:
            import bench as B
            from spectrobot_amd._lib import lib, dp
            a0 = sets[0]["state"]
            sel = np.unique(np.linspace(0, n_steps - 1, 80).round().astype(int))   # 80 of the set's steps
            a80 = dict(temps=a0["temps"][sel], press=a0["press"][sel], tvib=a0["tvib"][:, sel])
            q_part = np.zeros(len(sel))
            tt = np.ascontiguousarray(a80["temps"])
            assert lib.sr_calc_partition_sum(6, 1, tt.ctypes.data_as(dp), len(sel), q_part.ctypes.data_as(dp)) == 0
            one = np.array([0, 1], np.int32)
            cb = B.cpu_baseline(L, a80, grid, syn.CH4_MM, e_lev, q_part, args.cpu_seconds, len(sel),
                                (one, np.zeros(1, np.int32), np.ones(1)))
            # the oracle leg: folded coefficient rows of `ns` steps (value = share of 80 steps per second); a set has
            # n_steps step rows at T and at T + dT and 8 rays: spectra/s = 8 / (2 n_steps / 80 / value)
            per80 = cb["value"]
            cb["value"] = n_rays / (2.0 * n_steps / 80.0 / per80)
            cb["sample"] += "; scaled to a set of %d step rows at T and T + dT (the CPU leg runs the folded per-step op, no recursion of the 8 rays' Jacobians)" % n_steps
            extra["cpu_baseline"] = cb
            extra["speedup_vs_cpu_baseline"] = (len(szas) * n_rays / dt) / cb["value"]
    out = dict(base, **extra)
    out = dict(out, metric="limb spectra/sec with per-layer T and VMR Jacobians, 3-D path (BASELINE configs[3])",
               value=len(szas) * n_rays / dt if world == len(szas) or world == 1 else len(my) * n_rays * world / dt,
               ms_per_step=dt * 1e3, scaling="weak" if world > 1 else "n/a",
               config={"workload": "3-D atmosphere (BASELINE configs[3], use_tangent_sza = False): %d tangent SZA x %d rays fanned in "
                                   "azimuth, %d lines x %d-pt grid, kinetic T on (latitude box, altitude), T_vib by the local SZA: a "
                                   "coefficient row per LOS step (%d steps per set), d/dT_k per altitude layer and d/dVMR_k per level"
                                   % (len(szas), n_rays, n_lines, n_grid, n_steps),
                       "route": ("level-factored: pair tables on the distinct (P, T) rows at T and T + %.3f K (boundaries frozen), one "
                                 "combine per set with analytic d pop / d T" % DT_FACTORED) if ROUTE != "direct" else
                                ("direct: " + DT_SCHEME_NOTE + ", over the steps themselves"),
                       "device": info["name"]},
               checksum=float(res[0].sum().item()), los_steps_per_set=n_steps)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def _event_time(fn, kernel, bytes_alg, note, n=5):
    """HIP-event time of `fn` on the stream it launches on (torch's current stream) and the HBM roofline of its
    algorithmic bytes."""
    import torch
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    gbs = bytes_alg / (ms * 1e-3) / 1e9
    return {"kernel": kernel, "ms": ms, "bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0,
            "bytes_per_launch": bytes_alg, "note": note}


def main(args):
    import torch
    from spectrobot_amd import engine, synthetic as syn, distributed as sd, retrieval
    rank, local, world = sd.init_from_env()
    assert world == args.gpus
    engine.set_device(local % max(torch.cuda.device_count(), 1))
    import bench as _B
    _B.FLOP.update(_B.flop_model(engine.far_field_degree()))
    engine.set_level_route(getattr(args, "level_route", 1))
    info = engine.device_info()
    base = {"unit": "spectra/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "higher_is_better": True, "vs_baseline": None, "dtype": "f64", "data": "synthetic", "headline": False}
    if args.config == 2:
        n_rays = 64 if args.rays == 1 else args.rays
        grid, L, atm, e_lev = ch4_case(args.lines, args.grid, args.layers)
        ls = engine.LineSet(L, grid, 6, 1, syn.CH4_MM, e_lev)
        g_lo, g_hi = sd.shard_bounds(args.grid, world, rank)
        Lr = syn.limb_los(atm["z"], atm["nd"], [np.full(args.layers, 0.0148)], tangent_heights(n_rays))
        los = engine.LimbLOS(Lr["seg_off"], Lr["seg_layer"], Lr["pt_off"], Lr["x"], Lr["nd"], Lr["vmr"],
                             col_scale=[syn.CH4_ISO_RATIO])
        ab = torch.empty((args.layers, g_hi - g_lo), dtype=torch.float64, device="cuda")
        em = torch.empty_like(ab)
        full = torch.empty((n_rays, args.grid), dtype=torch.float64, device="cuda") if world > 1 else None

        def step():
            # as the headline's step: the LOS columns are integrated again on the device from the resident batch's sample
            # points (two launches), then coefficient op + folded recursion of the 64 rays
            los.refresh_columns()
            ls.abscoeff_layers(atm["temps"], atm["press"], tvib=atm["tvib"], g_lo=g_lo, g_hi=g_hi, out=(ab, em))
            rad = engine.limb_rays((ab, em), los)
            return sd.all_gather_spectrum(rad, args.grid, world, rank, out=full)

        dt, spec = _sync_time(step, args.steps, args.warmup)
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda" if torch.distributed.get_backend() == "nccl" else "cpu")
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        extra = {}
        if rank == 0:
            import bench as B
            kms, counts = B.serial_kernel_times_and_counts(engine, ls, lambda: ls.abscoeff_layers(
                atm["temps"], atm["press"], tvib=atm["tvib"], g_lo=g_lo, g_hi=g_hi, out=(ab, em)))
            extra["roofline"] = B.coefficient_roofline(kms, counts)
            extra["recursion"] = _event_time(lambda: (los.refresh_columns(), engine.limb_rays((ab, em), los))[1],
                                             "sr_los_columns_kernel + sr_fold_dense_pack_kernel + sr_limb_fold_fwd_kernel",
                                             bytes_alg=8.0 * (2 * ab.numel() + n_rays * ab.shape[1]),
                                             note="64 rays x 160 segments x %d points: algorithmic bytes = the two coefficient "
                                                  "tables once + the radiances; VALU-bound (one fused exp / expm1 per segment "
                                                  "and point), the 64 rays re-read the tables from L2 / MALL" % ab.shape[1])
            if world == 1 and args.cpu_seconds > 0:
                q_part = np.zeros(args.layers)
                from spectrobot_amd._lib import lib, dp
                tt = np.ascontiguousarray(atm["temps"])
                assert lib.sr_calc_partition_sum(6, 1, tt.ctypes.data_as(dp), args.layers, q_part.ctypes.data_as(dp)) == 0
                extra["cpu_baseline"] = B.cpu_baseline(L, atm, grid, syn.CH4_MM, e_lev, q_part, args.cpu_seconds, args.layers,
                                                       (Lr["seg_off"], Lr["seg_layer"], B.cpu_columns(Lr, syn.CH4_ISO_RATIO)))
                extra["speedup_vs_cpu_baseline"] = n_rays / dt / extra["cpu_baseline"]["value"]
        if world > 1:
            # what the collective really was (as the headline's line): torch.distributed's own view of the group, which
            # branch of all_gather_spectrum ran, and this rank's own shard found again in the gathered spectrum
            rad_chk = engine.limb_rays((ab, em), los)
            gathered = sd.all_gather_spectrum(rad_chk, args.grid, world, rank)
            torch.cuda.synchronize()
            extra["dist"] = dict(sd.dist_info(), async_gather=False, gathers=dict(sd.stats),
                                 gathered_holds_own_shard=bool(torch.equal(gathered[:, g_lo:g_hi], rad_chk)),
                                 note="n_rays > 1: one blocking all_gather_into_tensor of [n_rays, n_grid / W] per rank + one "
                                      "permuting copy (distributed.all_gather_spectrum)")
        out = dict(base, **extra)
        out = dict(out, metric="limb spectra/sec, 64 rays batched (BASELINE configs[2])", value=n_rays / dt,
                   ms_per_step=dt * 1e3, scaling="strong",
                   config={"workload": "CH4 Titan limb (BASELINE configs[2]): %d lines x %d-pt grid x %d layers, %d rays on one "
                                       "coefficient op, device LOS pipeline, spectral window / %d + one all-gather"
                                       % (args.lines, args.grid, args.layers, n_rays, world), "device": info["name"]},
                   checksum=float(spec.sum().item()))
    elif args.config == 3 and getattr(args, "three_d", False):
        return config3_3d(args, rank, world, info, base)
    elif args.config == 3:
        n_grid, n_lines, n_rays, szas = 200000, 200000, 8, [30.0, 37.0, 44.0, 51.0, 58.0, 65.0, 72.0, 80.0]
        grid, L, atm, e_lev = ch4_case(n_lines, n_grid, args.layers, config_id=3, w0=2950.0)
        ls = engine.LineSet(L, grid, 6, 1, syn.CH4_MM, e_lev)
        vm = np.full(args.layers, 0.0148)
        Lr = syn.limb_los(atm["z"], atm["nd"], [vm], 120.0 + 60.0 * np.arange(n_rays))
        los = engine.LimbLOS(Lr["seg_off"], Lr["seg_layer"], Lr["pt_off"], Lr["x"], Lr["nd"], Lr["vmr"], col_scale=[syn.CH4_ISO_RATIO])
        W = layer_vmr_weights(atm["z"], Lr["alt"])
        my = szas[rank::world]                       # independent ray batches: SZA sets split over the ranks, no collective
        atms = [sza_atmosphere(atm, sza) for sza in my]
        nl = args.layers

        def coef3(a, scheme=DT_SCHEME):
            return engine.coefficients_dT(ls, a["temps"], a["press"], tvib=a["tvib"], scheme=scheme)

        pg = np.zeros(args.layers, np.int32)
        # level-factored route: the sets share their (P, T) rows (one latitude box); only T_vib differs
        T_rows, P_rows, row0 = engine.LevelFactored.unique_rows(np.concatenate([a["temps"] for a in atms]),
                                                                 np.concatenate([a["press"] for a in atms]))
        tv_all = np.concatenate([a["tvib"] for a in atms], axis=1)
        timing = {}

        state = {}

        def step_factored():
            # the tables are REBUILT every step (a retrieval iteration moves the temperatures), in place
            lf = state["lf"].rebuild(T_rows, P_rows) if "lf" in state else state.setdefault("lf", engine.LevelFactored(ls, T_rows, P_rows, dT=DT_FACTORED))
            (ca, ce), (da, de) = lf.steps(row0, tvib=tv_all, derivative=True)     # all sets' layers in one combine
            res = None
            for i in range(len(my)):
                sl = slice(i * nl, (i + 1) * nl)
                # radiances + d/dT_k + d/dVMR_k of the 8 rays in one pass per ray
                res = engine.limb_rays_jacobians((ca[sl], ce[sl]), los, dcoeffs=(da[sl], de[sl]), par_gas=pg, par_w=W)
            return res

        def step_direct(scheme=DT_SCHEME):
            res = None
            for a in atms:
                co, dco = coef3(a, scheme)
                res = engine.limb_rays_jacobians(co, los, dcoeffs=dco, par_gas=pg, par_w=W)
            return res

        step = step_direct if ROUTE == "direct" else step_factored
        dt, res = _sync_time(step, max(1, args.steps // 4), min(args.warmup, 1))
        extra = {}
        dt_c, res_c = _sync_time(lambda: step_direct("central"), max(1, args.steps // 8), 1)
        dt_d = None
        if ROUTE != "direct":
            dt_d, _ = _sync_time(step_direct, max(1, args.steps // 8), 1)
        if rank == 0:
            jt, jt_c = res[1], res_c[1]      # temperature Jacobians of the last set: this route / three folded ops, central
            extra["temperature_derivative"] = {
                "scheme": ("pair tables at T and T + %.3f K (boundaries frozen at T, line weights linearised about T), d pop / d T analytic" % DT_FACTORED) if ROUTE != "direct" else DT_SCHEME_NOTE,
                "ms_per_step_direct_central_differences": dt_c * 1e3,
                "spectra_per_s_direct_central_differences": len(szas) * n_rays / dt_c if world in (1, len(szas)) else None,
                "ms_per_step_direct_forward_difference": None if dt_d is None else dt_d * 1e3,
                "max_rel_dev_of_T_jacobian_from_central": float(((jt - jt_c).abs().amax() / jt_c.abs().amax()).item())}
            del jt, jt_c
        del res_c
        if rank == 0:
            import bench as B
            a0 = atms[0]
            kms, counts = B.serial_kernel_times_and_counts(engine, ls, lambda: ls.abscoeff_layers(a0["temps"], a0["press"], tvib=a0["tvib"]), n=3)
            extra["roofline"] = B.coefficient_roofline(kms, counts)
            if ROUTE != "direct" and getattr(args, "level_route", 1) == 1:
                # the step's dominant kernel is the table builds' sr_zones_mc_kernel (two builds per step): its roofline is the
                # line's; the folded op's kernels (the direct route's) beside it
                tab_tmp = torch.empty((12, 2, len(T_rows), n_grid), dtype=torch.float64, device="cuda")
                extra["roofline_folded_op"] = extra["roofline"]
                extra["roofline"] = B.level_tables_roofline(engine, ls, lambda: ls.glevel_pairs(T_rows, P_rows, out=tab_tmp),
                                                            lambda: ls.abscoeff_layers(T_rows, P_rows))
                del tab_tmp
            co, dco = coef3(a0)
            jac_bytes = 8.0 * (res[1].numel() + res[2].numel())
            extra["jacobian_kernel"] = _event_time(
                lambda: engine.limb_rays_jacobians(co, los, dcoeffs=dco, par_gas=pg, par_w=W), "sr_limb_adjoint_fold_kernel<1, true, true, 1>",
                bytes_alg=jac_bytes + 8.0 * (4 * co[0].numel() + res[0].numel()),
                note="per set of 8 rays: algorithmic bytes = the two Jacobians written once (%.2f GB) + the four coefficient "
                     "tables read once + the radiances; the folded kernel takes a ray's two segments of a shell together "
                     "and stores every value once (round 4; the path-order kernels stored and read-add-stored)" % (jac_bytes / 1e9))
            del co, dco
            if ROUTE != "direct":
                for _ in range(2):  # the second build: the first one after the counting passes re-creates scratch
                    lf = None
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    lf = engine.LevelFactored(ls, T_rows, P_rows, dT=DT_FACTORED)
                    torch.cuda.synchronize()
                    timing["tables_ms"] = (time.perf_counter() - t0) * 1e3
                pop, dpop = ls.level_populations(T_rows[row0], tvib=tv_all, derivative=True)
                outs = [torch.empty((len(row0), n_grid), dtype=torch.float64, device="cuda") for _ in range(4)]
                extra["combine_kernel"] = _event_time(
                    lambda: engine.glevel_combine(lf.tab, row0, pop, tab_dT=lf.tab_dT, dpop=dpop, dT=DT_FACTORED, out=outs),
                    "sr_glevel_combine_kernel<12, true>", 8.0 * n_grid * (2 * 2 * lf.tab.shape[0] * len(T_rows) + 4 * len(row0)),
                    "all %d sets at once: %d (P, T) rows x 2 x 12 pair spectra x 2 tables read once + abs, emi, d abs / d T, "
                    "d emi / d T of %d (set, layer) steps written" % (len(my), len(T_rows), len(row0)))
                del lf, outs
                extra["tables"] = {"rows": int(len(T_rows)), "steps_all_sets": int(len(row0)),
                                   "ms_per_build_T_and_T_plus_dT": timing.get("tables_ms"),
                                   "note": "the %d SZA pixels lie in one latitude box and share its kinetic profile: the 24 pair "
                                           "spectra of a layer are built once (at T and at T + %.3f K) for all of them"
                                           % (len(my), DT_FACTORED)}
            if world == 1 and args.cpu_seconds > 0:
                from spectrobot_amd._lib import lib, dp
                q_part = np.zeros(args.layers)
                tt = np.ascontiguousarray(a0["temps"])
                assert lib.sr_calc_partition_sum(6, 1, tt.ctypes.data_as(dp), args.layers, q_part.ctypes.data_as(dp)) == 0
                cb = B.cpu_baseline(L, a0, grid, syn.CH4_MM, e_lev, q_part, args.cpu_seconds, args.layers,
                                    (Lr["seg_off"], Lr["seg_layer"], B.cpu_columns(Lr, syn.CH4_ISO_RATIO)))
                # the oracle leg is ONE folded coefficient op + the radiances of the set's 8 rays; the reference-shaped
                # CPU path needs it at T and at T + dT per set (it has no level-factored shortcut across sets unless it
                # keeps 24 spectra per layer in memory): the Jacobian recursions themselves are not in the CPU figure
                cb["value"] = cb["value"] / 2.0
                cb["sample"] += "; /2: a set needs the coefficient op at T and at T + dT (the CPU leg runs one and no Jacobian recursion)"
                extra["cpu_baseline"] = cb
                extra["speedup_vs_cpu_baseline"] = (len(szas) * n_rays / dt) / cb["value"]
        out = dict(base, **extra)
        out = dict(out, metric="limb spectra/sec with per-layer T and VMR Jacobians (BASELINE configs[3])",
                   value=len(szas) * n_rays / dt if world == len(szas) or world == 1 else len(my) * n_rays * world / dt,
                   ms_per_step=dt * 1e3, scaling="weak" if world > 1 else "n/a",
                   config={"workload": "3D-atmosphere ray sets (BASELINE configs[3]): %d SZA x %d rays, %d lines x %d-pt grid x %d "
                                       "layers, d/dT_k and d/dVMR_k (analytic) for every layer; kinetic T on (latitude box, altitude) "
                                       "-- the pixels share one box --, T_vib by the SZA; SZA sets are independent batches (split "
                                       "over ranks, no collective)" % (len(szas), n_rays, n_lines, n_grid, args.layers),
                           "route": ("level-factored: pair tables of the %d shared (P, T) rows at T and T + %.3f K, one combine for "
                                     "all sets, analytic d pop / d T" % (len(T_rows), DT_FACTORED)) if ROUTE != "direct" else
                                    ("direct: " + DT_SCHEME_NOTE + ", a folded op per set"),
                           "device": info["name"]},
                   checksum=float(res[0].sum().item()), jacobian_gb_per_sza=(res[1].numel() + res[2].numel()) * 8 / 1e9)
    elif args.config == 4:
        scene = two_gas_scene(40000, 8000, 60000, 60)
        bs, pixels, x_true = retrieval_problem(scene)
        # N > 1: every rank its spectral shard of radiances, Jacobians and partial band integrals, one all-reduce per
        # iteration, the algebra replicated (retrieval.simulate; spect_main_module.py:2814-2853)
        shard = sd.shard_bounds(len(scene.grid), world, rank) if world > 1 else None
        import copy
        import gc
        bs0 = copy.deepcopy(bs)
        # warm-up (the bench contract's untimed steps): one whole retrieval from the first guess -- geometry, resident LOS
        # batch, band weight table and buffers exist afterwards, as they do for every retrieval after the first of a run.
        # Timed: `--steps`-many iterations at least, in whole retrievals from the same first guess (the loop converges in
        # 8 iterations: one retrieval alone is a 5 ms sample)
        retrieval.inversion_fast_limb(scene, copy.deepcopy(bs0), pixels, max_it=20, shard=shard)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        gc.collect()
        gc.freeze()      # (see _sync_time)
        # (at least 200 iterations: three retrievals -- `--steps 20` -- are a 10 ms sample that starts on an idle GPU's
        # clocks and read 2240-2280 it/s where 30 retrievals in a row read 2575, same box, round 6)
        n_min = max(200, args.steps)
        n_it, n_runs, starts = 0, 0, [copy.deepcopy(bs0) for _ in range(n_min)]
        t0 = time.perf_counter()
        while n_it < n_min:
            chi, obs, sims, bs = retrieval.inversion_fast_limb(scene, starts[n_runs], pixels, max_it=20, shard=shard)
            n_it += len(bs.history)
            n_runs += 1
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        dt = time.perf_counter() - t0
        # the same loop with the coefficient op of both gases inside every iteration (what a retrieval of temperatures
        # or vibrational temperatures costs; in a VMR retrieval like this one the coefficients do not change)
        t0 = time.perf_counter()
        _, _, _, bs_r = retrieval.inversion_fast_limb(scene, bs0, pixels, max_it=20, shard=shard, refresh=True)
        torch.cuda.synchronize()
        dt_refresh = (time.perf_counter() - t0) / len(bs_r.history)
        out = dict(base, metric="retrieval iterations/sec, HCN + CH4 (BASELINE configs[4])", unit="iterations/s",
                   value=n_it / dt, ms_per_step=dt / n_it * 1e3, steps=n_it, scaling="strong" if world > 1 else "n/a",
                   config={"workload": "Gauss-Newton / LM retrieval loop (BASELINE configs[4]): HCN + CH4, 40000 + 8000 lines x "
                                       "60000-pt grid x 60 layers, 6 pixels x 3 LOS, 7 profile parameters, %d VIMS-like bands, "
                                       "max 20 iterations, reference stopping rule" % len(scene.bands_nm),
                           "coefficients": "computed once before the loop and cached: only VMRs are retrieved, the coefficient "
                                           "spectra do not depend on them (the reference's drivers do the same through their "
                                           "LUTs); ms_per_iteration_with_coefficient_refresh recomputes both gases' coefficient "
                                           "op in every iteration",
                           "sharding": ("spectral window / %d: radiances, Jacobians and partial band integrals per rank, one "
                                        "all-reduce of [n_los x (1 + n_par) x n_bands] per iteration, algebra replicated" % world
                                        if world > 1 else "none"),
                           "timed": "%d whole retrievals from the same first guess after one untimed retrieval" % n_runs,
                           "device": info["name"]},
                   ms_per_iteration_with_coefficient_refresh=dt_refresh * 1e3,
                   chi_history=[float(c) for c in bs.history], stop=bs.stop,
                   max_rel_dev_from_truth=float(np.max(np.abs(bs.param_vector() - x_true) / x_true)))
        if world > 1:
            out["dist"] = sd.dist_info()
        if rank == 0:
            # roofline of the iteration's dominant kernel: radiances + column-parameter Jacobians of the 18 LOS
            alts = [a for pix in pixels for a in pix.los_alts()]
            los, alt = scene.los(alts)
            g_lo, g_hi = (0, len(scene.grid)) if shard is None else retrieval.shard_with_halo(len(scene.grid), *shard)
            coeffs = scene.coefficients(g_lo=g_lo, g_hi=g_hi)
            par_gas, par_w = scene.profile_weights(bs, alt)
            n_sh = g_hi - g_lo
            # ... measured on the launch the timed iterations make (round 6: the one-sweep kernel with the instrument bands in
            # its epilogue, behind engine.retrieval_forward): HIP events of the library's own around record packing + kernel
            with_fov = sum(pix.fov_half > 0 for pix in pixels)
            los_b, pg_b, pw_b = retrieval._one_call_batch(scene, pixels, bs, alts, with_fov)
            cstack = scene.coefficient_stack(g_lo=g_lo, g_hi=g_hi)
            engine.set_timing(2)
            k_ms = []
            for _ in range(6):
                engine.retrieval_forward(cstack, los_b, pg_b, pw_b, bs.param_vector(), scene.grid, scene.bands_nm, scene.widths_nm,
                                         out_units=scene.out_units, g_lo=g_lo, fov=scene._fov_fac if with_fov else None,
                                         buf=getattr(scene, "_fwd_buf", None))
                k_ms.append(los_b.last_forward_kernel_ms(len(scene.z), pg_b, pw_b, scene.grid))
            engine.set_timing(1)
            ms = float(np.mean(k_ms[1:]))
            bytes_alg = 8.0 * n_sh * (2 * 2 * len(scene.z)) + 8.0 * len(alts) * (1 + len(par_gas)) * len(scene.bands_nm) * ((n_sh + 63) // 64)
            gbs = bytes_alg / (ms * 1e-3) / 1e9
            out["roofline"] = {
                "kernel": "sr_limb_fold_sens_lds_kernel<2, true> (+ sr_fold_dense_pack_kernel, ~4 us, inside the same two events)",
                "ms": ms, "bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0, "bytes_per_launch": bytes_alg,
                "note": "one iteration's forward model: %d LOS x (radiance + %d parameter Jacobians) on %d points, the %d instrument "
                        "bands integrated in the kernel's epilogue (an MFMA product per wave; no spectra are written: algorithmic "
                        "bytes = the two gases' coefficient tables once + a partial band sum per wave); the rays re-read the tables "
                        "from L2 / MALL.  The launch is small (4230 blocks of 60 dependent shell visits) and bound by that chain and "
                        "by its VALU work (63 %% issue-busy), not by HBM (round 5: one sweep of forward sensitivities in fold order "
                        "with the ray's records in LDS, 0.48 -> 0.26 ms; round 6: + the band integrals, 0.26 + 0.057 (instrument "
                        "kernel) -> 0.28)" % (len(alts), len(par_gas), n_sh, len(scene.bands_nm))}
            out["roofline"]["traffic"] = None
            # the kernel's own counters say 63 % VALU-issue-busy and 87 % L2 hits -- its bound is the fp64
            # vector unit (and the latency of a ray's 60 dependent shell visits), not HBM: the line's fraction is flop based.
            # Flop model per (point, ray, shell visit) of the one-sweep kernel (sr_limb_fold_sens_lds_kernel): the
            # far and the near segment's attenuation (one fused exp / expm1 / quotient each, ~34 flop), tau and the source of
            # the gases (6 flop per gas and segment), the recursion itself (12), five fma per parameter.
            n_seg_all = int(los.seg_off[-1])                   # both segments of every shell a ray crosses
            n_par_, n_gas_ = len(par_gas), len(coeffs)
            flops = float(n_sh) * (n_seg_all * (34.0 + 6.0 * n_gas_ + 6.0) + 0.5 * n_seg_all * 10.0 * n_par_)
            flops += 2.0 * n_sh * len(alts) * (1 + n_par_) * len(scene.bands_nm)     # the band integrals (useful flops of the MFMA tiles)
            hb = dict(out["roofline"])
            out["roofline"].update({"bound": "fp64-valu", "achieved": flops / (hb["ms"] * 1e-3) / 1e12, "peak": 78.6, "unit": "TFLOP/s",
                                    "frac": flops / (hb["ms"] * 1e-3) / 1e12 / 78.6, "flops_per_launch": flops,
                                    "hbm_view": {"achieved_gbs": hb["achieved"], "frac_of_8_tb_s": hb["frac"],
                                                 "bytes_per_launch": hb["bytes_per_launch"]}})
            if world == 1 and args.cpu_seconds > 0:
                # CPU leg: the oracle's recursion (two gases mixed on the host) for the 18 LOS, once for the radiances and
                # once per parameter -- what a CPU port without the sensitivity recursion does per iteration (finite
                # differences); coefficients cached as on the GPU.  One thread (the oracle's recursion is scalar).
                from oracle import oracle as O
                import time as _t
                a_h = [c[0].cpu().numpy() for c in coeffs]
                e_h = [c[1].cpu().numpy() for c in coeffs]
                col = los.columns()
                import bench as B
                from concurrent.futures import ThreadPoolExecutor
                cores = B.host_cores()
                n_do = len(alts) * max(1, int(np.ceil(cores / len(alts))))     # whole passes over the LOS, at least one per core

                def one(i):
                    r = i % len(alts)
                    sl = slice(los.seg_off[r], los.seg_off[r + 1])
                    lay = los.seg_layer[sl]
                    O.radiance_ray(a_h[0], e_h[0], lay, col[0][sl])      # (the C recursion releases the GIL)
                    O.radiance_ray(a_h[1], e_h[1], lay, col[1][sl])
                t0 = _t.time()
                with ThreadPoolExecutor(cores) as ex:
                    list(ex.map(one, range(n_do)))
                t_wall = _t.time() - t0
                t_iter = t_wall / n_do * len(alts) * (1 + len(par_gas))
                out["cpu_baseline"] = {"value": 1.0 / t_iter, "unit": "iterations/s", "cores": cores, "kind": "port",
                                       "sample": "the oracle's recursion of %d LOS passes (both gases' tables, %d points) on %d "
                                                 "threads: %.3f s, x %d LOS x (1 + %d parameters by finite differences) per "
                                                 "iteration; coefficients cached as on the GPU; instrument step and algebra "
                                                 "not included" % (n_do, n_sh, cores, t_wall, len(alts), len(par_gas))}
                out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    elif args.config == "lut":
        # The one thing the reference publishes a figure for: the look-up-table / G-coefficient build,
        # "n_lines x 3 / 30000 x n_PT minutes" = 6 ms per (line, P-T couple) with its n_threads worker processes
        # (spect_main_module.py:791-801; :1863 "like 3 x len(PTcouples) minutes").  LookUpTable.make on the configs[1]
        # case: 1e5 lines, 12 levels, 3 ctypes, the calc_PT_couples_atmosphere lattice of its 80-layer atmosphere with the
        # LUTopt of the reference's 3-D driver (temp_step 5 K, pres_step_log 1.0: radtran_3Dvs2D_sza30-80_test.py:313-316).
        from spectrobot_amd import spect_main_module as smm, spect_classes as spcl, spect_base_module as sbm
        grid, L, atm, e_lev = ch4_case(args.lines, args.grid, args.layers)
        iso = sbm.IsoMolec(6, 1, syn.CH4_MM, mol_name="CH4")
        for i, e in enumerate(e_lev):
            iso.add_level("L%02d" % i, e, local_vibtemp=atm["tvib"][i])
        iw = int(np.argmax(L["air_broad"]))
        widest = spcl.SpectLine([6, 1, L["freq"][iw], 0.0, L["a_coeff"][iw], L["air_broad"][iw], 0.0, L["e_lower"][iw],
                                 L["t_dep_broad"][iw], 0.0, "L%02d" % L["lev_up"][iw], "L%02d" % L["lev_lo"][iw], "", "", "",
                                 L["g_up"][iw], L["g_lo"][iw]], nomi=spcl.cose_hit)

        class Atm(object):
            pres, temp = atm["press"], atm["temps"]
        PT = smm.calc_PT_couples_atmosphere([widest], iso, Atm, pres_step_log=1.0, temp_step=5.0)
        ls = engine.LineSet(L, grid, 6, 1, syn.CH4_MM, e_lev)
        sg = spcl.SpectralGrid(grid, units="cm_1")
        lut = smm.LookUpTable(iso, [grid[0], grid[-1]], LTE=False)

        def step():
            lut.make(sg, ls, PT, pt_batch=len(PT))
            return lut

        dt, _ = _sync_time(step, max(1, args.steps // 10), 1)
        n_pairs = float(ls.n_kept) * len(PT)
        value = n_pairs / dt
        gb = sum(st.device.numel() for st in lut.sets.values()) * 8 / 1e9
        extra = {}
        if rank == 0:
            import bench as B
            # roofline: the dominant kernel of the build is the coefficient op's zones kernel, on the level sub-linesets
            kms, counts = B.serial_kernel_times_and_counts(engine, ls, lambda: ls.abscoeff_level(
                [pt[1] for pt in PT], [pt[0] for pt in PT], 0), n=3)
            extra["roofline"] = B.coefficient_roofline(kms, counts)
            extra["roofline"]["note"] = ("the largest of the per-level route's 24 coefficient ops (the pass over the lines whose lower or "
                                         "upper level is level 0: 80 %% of the list, %d (P, T) rows; timed with the tracked-level "
                                         "weights, the same kernels): executed flops of its dominant kernel / its stand-alone "
                                         "HIP-event duration" % len(PT))
            if getattr(args, "level_route", 1) == 1:
                Tl, Pl = np.array([pt[1] for pt in PT]), np.array([pt[0] for pt in PT])
                g_tmp = torch.empty((12, 3, len(PT), len(grid)), dtype=torch.float64, device="cuda")
                extra["roofline_per_level_op"] = extra["roofline"]
                extra["roofline"] = B.level_tables_roofline(engine, ls, lambda: ls.gcoeff_levels(Tl, Pl, out=g_tmp),
                                                            lambda: ls.abscoeff_layers(Tl, Pl))
                extra["roofline"]["kernel"] = "sr_zones_mc_kernel<128, 8>"
                del g_tmp
            if world == 1 and args.cpu_seconds > 0:
                from oracle import oracle as O
                import time as _t
                # the oracle's G-coefficient spectra (its restatement of add_PT -> BuildCoeff for every level and ctype) on a
                # bounded sample of the couples, all host cores
                cores = B.host_cores()
                ns = int(max(1, min(len(PT), round(args.cpu_seconds * cores / (3 * 55e-6 * ls.n_kept)))))
                sel = np.linspace(0, len(PT) - 1, ns).round().astype(int)
                t0 = _t.time()
                O.gcoeff_layers(L, syn.CH4_MM, e_lev, np.array([PT[i][1] for i in sel]), np.array([PT[i][0] for i in sel]),
                                np.ones(len(sel)), None, grid, n_threads=cores)
                t_cpu = _t.time() - t0
                extra["cpu_baseline"] = {"value": float(ls.n_kept) * ns / t_cpu, "unit": "(line, PT couple) pairs/s", "cores": cores,
                                         "kind": "port",
                                         "sample": "%d of the %d couples, all %d lines, all 12 levels x 3 ctypes, full %d-point grid: "
                                                   "%.1f s on %d threads" % (ns, len(PT), ls.n_kept, len(grid), t_cpu, cores)}
                extra["speedup_vs_cpu_baseline"] = value / extra["cpu_baseline"]["value"]
        out = dict(base, **extra)
        ref_pairs_per_s = 1.0 / 6e-3
        out = dict(out, metric="look-up-table build: (line, P-T couple) pairs per second (LookUpTable.make)",
                   unit="(line, PT couple) pairs/s", value=value, ms_per_step=dt * 1e3, scaling="n/a",
                   vs_baseline=None,   # (BASELINE.md holds no published number for this metric; the reference's inline estimate is beside it)
                   reference_estimate={"value": ref_pairs_per_s, "unit": "(line, PT couple) pairs/s", "ratio": value / ref_pairs_per_s,
                                       "source": "spect_main_module.py:791-801: n_lines x 3 / 30000 x n_PT minutes = 6 ms per "
                                                 "(line, PT couple) with n_threads worker processes, hardware not stated "
                                                 "(BASELINE.md 1); for this table: %.0f minutes" % (n_pairs * 6e-3 / 60.0)},
                   config={"route": "multi-channel pass: all levels and ctypes of a batch of couples from one walk of the line list "
                                    "(sr_gcoeff_levels_dev)" if getattr(args, "level_route", 1) == 1 else "one coefficient op per level and ctype pair (sr_gcoeff_layers_dev)",
                           "workload": "LookUpTable.make (spect_main_module.py:718-788): %d lines x 12 levels x 3 ctypes x %d (P, T) "
                                       "couples (calc_PT_couples_atmosphere of the configs[1] atmosphere, temp_step 5 K, pres_step_log "
                                       "1.0) x %d-pt grid = %.1f GB of G spectra, resident in HBM" % (ls.n_kept, len(PT), len(grid), gb),
                           "line_evaluations_per_line": "1 (multi-channel pass; the per-level route: 3 -- absorption + sp_emission in one pass "
                                                        "over the lines whose lower or upper level is L, ind_emission in a second pass "
                                                        "over the upper-level lines only)",
                           "device": info["name"]})
    else:
        raise SystemExit("--config must be 1..4 or lut")
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
