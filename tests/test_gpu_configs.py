"""BASELINE.json configs[1..4] AT THEIR WORKLOAD on the GPU (bench_configs.py builds the synthetic inputs of
SURVEY 8-d): coefficient parity against the oracle on sampled layers at full lines x grid, radiances against the
oracle's recursion, shard-vs-whole, Jacobians against finite differences, the 20-iteration two-gas retrieval.
Needs a real MI355X."""
import os

import numpy as np
import pytest

from conftest import relerr, far_tol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    from spectrobot_amd import engine
    engine.set_device(0)
    return engine


def _q_of(mol, iso, temps):
    """Q(T) for the ORACLE's side of a comparison from the oracle itself: its Lagrange restatement over the table the
    reference's Fortran returned (tests/golden/tips2003.npz) -- nothing of the product."""
    from conftest import GOLDEN
    from oracle import oracle as O
    g = np.load(os.path.join(GOLDEN, "tips2003.npz"), allow_pickle=False)
    keys = [tuple(int(v) for v in k) for k in g["keys"]]
    tab = g["q_tab"][keys.index((int(mol), int(iso)))]
    return np.array([O.calc_partition_sum(g["t_grid"], tab, float(t)) for t in np.atleast_1d(np.asarray(temps, float))])


def _q(temps):
    return _q_of(6, 1, temps)


@pytest.fixture(scope="module")
def cfg1(eng, oracle):
    """configs[1] / [2] inputs, the GPU coefficients of all 80 layers and the oracle's on 8 of them."""
    import bench_configs as bc
    from spectrobot_amd import synthetic as syn
    grid, L, atm, e_lev = bc.ch4_case(100000, 100000, 80)
    ls = eng.LineSet(L, grid, 6, 1, syn.CH4_MM, e_lev)
    ab, em = ls.abscoeff_layers(atm["temps"], atm["press"], tvib=atm["tvib"])
    sel = np.array([0, 9, 21, 33, 44, 56, 68, 79])
    abo, emo = oracle.abscoeff_layers(L, syn.CH4_MM, e_lev, atm["temps"][sel], atm["press"][sel], _q(atm["temps"][sel]),
                                      atm["tvib"][:, sel], grid, mode=1, n_threads=min(8, os.cpu_count() or 1))
    return dict(grid=grid, L=L, atm=atm, e_lev=e_lev, ls=ls, ab=ab, em=em, sel=sel, abo=abo, emo=emo)


def test_config1_full_size_vs_oracle(eng, cfg1):
    """configs[1]: 1e5 lines x 1e5 grid x 80 layers -- 8 layers from the Lorentz- to the Doppler-dominated end
    against the oracle at full lines x grid (north_star bound 1e-6; observed ~4e-12), both evaluation modes."""
    ab, em = cfg1["ab"][cfg1["sel"]].cpu().numpy(), cfg1["em"][cfg1["sel"]].cpu().numpy()
    assert relerr(ab, cfg1["abo"]) < 1e-10 and relerr(em, cfg1["emo"]) < 1e-10
    atm, sel = cfg1["atm"], cfg1["sel"][[0, 4, 7]]
    eng.set_far_field(0)
    try:
        a0, e0 = cfg1["ls"].abscoeff_layers(atm["temps"][sel], atm["press"][sel], tvib=atm["tvib"][:, sel])
    finally:
        eng.set_far_field(eng.FAR_FIELD_DEFAULT)
    assert relerr(a0.cpu().numpy(), cfg1["abo"][[0, 4, 7]]) < 1e-10 and relerr(e0.cpu().numpy(), cfg1["emo"][[0, 4, 7]]) < 1e-10


def test_config2_64_rays_and_eight_shards(eng, oracle, cfg1):
    """configs[2]: 64 tangent-height rays (z_t = 100 + 12.5 r km) batched on the coefficients of configs[1]:
    radiances through the device LOS pipeline against the oracle's recursion (all rays, a 1e4-point window of the
    grid); the 8 spectral shards of shard_bounds(1e5, 8, r) computed one after the other and concatenated equal
    the unsharded coefficients and radiances."""
    import torch
    import bench_configs as bc
    from spectrobot_amd import synthetic as syn, distributed as sd
    atm, ls, grid = cfg1["atm"], cfg1["ls"], cfg1["grid"]
    Lr = syn.limb_los(atm["z"], atm["nd"], [np.full(80, 0.0148)], bc.tangent_heights(64))
    los = eng.LimbLOS(Lr["seg_off"], Lr["seg_layer"], Lr["pt_off"], Lr["x"], Lr["nd"], Lr["vmr"], col_scale=[syn.CH4_ISO_RATIO])
    rad = eng.limb_rays((cfg1["ab"], cfg1["em"]), los)
    assert tuple(rad.shape) == (64, 100000) and bool(torch.isfinite(rad).all()) and float(rad.min()) >= 0.0
    col = los.columns()[0]
    lo, hi = 45000, 55000
    a_h, e_h = cfg1["ab"][:, lo:hi].cpu().numpy(), cfg1["em"][:, lo:hi].cpu().numpy()
    rad_h = rad[:, lo:hi].cpu().numpy()
    for r in range(64):
        s = slice(Lr["seg_off"][r], Lr["seg_off"][r + 1])
        want = oracle.radiance_ray(a_h, e_h, Lr["seg_layer"][s], col[s])
        assert relerr(rad_h[r], want) < 1e-12, r
    # the band-integrated limb radiance rises with tangent height while the lower atmosphere is opaque
    # (non-LTE emission from above), peaks, and falls monotonically above the peak
    tot = rad.sum(dim=1).cpu().numpy()
    pk = int(np.argmax(tot))
    assert 0 < pk < 60 and np.all(np.diff(tot[pk:]) < 0)
    parts_a, parts_r = [], []
    for r in range(8):
        g_lo, g_hi = sd.shard_bounds(100000, 8, r)
        a_s, e_s = ls.abscoeff_layers(atm["temps"], atm["press"], tvib=atm["tvib"], g_lo=g_lo, g_hi=g_hi)
        parts_a.append(a_s)
        parts_r.append(eng.limb_rays((a_s, e_s), los))
    a_cat, r_cat = torch.cat(parts_a, dim=1), torch.cat(parts_r, dim=1)
    assert float(((a_cat - cfg1["ab"]).abs() / cfg1["ab"].abs()).max()) < far_tol(1e-12)   # (a shard's boxes start at its first point)
    assert float(((r_cat - rad).abs() / rad.abs().clamp_min(1e-300)).max()) < far_tol(1e-12)


def test_config3_2e5_grid_T_and_vmr_jacobians(eng, oracle):
    """configs[3]: 2e5-point grid, 2e5 lines, 80 layers, a set of 8 rays at one of the 8 SZA atmospheres (the
    bench loops over all 8): coefficients against the oracle on sampled layers; per-layer temperature Jacobian
    and per-layer VMR Jacobian against finite differences of the whole chain at sampled layers."""
    import torch
    import bench_configs as bc
    from spectrobot_amd import synthetic as syn
    n = 200000
    grid, L, atm0, e_lev = bc.ch4_case(n, n, 80, config_id=3, w0=2950.0)
    atm = bc.sza_atmosphere(atm0, 51.0)
    ls = eng.LineSet(L, grid, 6, 1, syn.CH4_MM, e_lev)
    T, P, tv = atm["temps"], atm["press"], atm["tvib"]
    ab, em = ls.abscoeff_layers(T, P, tvib=tv)
    sel = np.array([2, 30, 61])
    abo, emo = oracle.abscoeff_layers(L, syn.CH4_MM, e_lev, T[sel], P[sel], _q(T[sel]), tv[:, sel], grid, mode=1, n_threads=3)
    assert relerr(ab[sel].cpu().numpy(), abo) < 1e-10 and relerr(em[sel].cpu().numpy(), emo) < 1e-10
    vmr = np.full(80, 0.0148)
    tz = 120.0 + 60.0 * np.arange(8)
    Lr = syn.limb_los(atm["z"], atm["nd"], [vmr], tz)
    W = bc.layer_vmr_weights(atm["z"], Lr["alt"])
    assert np.allclose(W.sum(axis=0), 1.0)

    def los_for(v):
        return eng.LimbLOS(Lr["seg_off"], Lr["seg_layer"], Lr["pt_off"], Lr["x"], Lr["nd"], (W.T @ v)[None, :],
                           col_scale=[syn.CH4_ISO_RATIO])

    los = los_for(vmr)
    dT = 0.05
    ap = ls.abscoeff_layers(T + dT, P, tvib=tv)
    am = ls.abscoeff_layers(T - dT, P, tvib=tv)
    dco = ((ap[0] - am[0]) / (2 * dT), (ap[1] - am[1]) / (2 * dT))
    del ap, am
    jt = eng.limb_rays_layer_jacobian((ab, em), dco, los)
    rad, jv = eng.limb_rays_jacobian((ab, em), los, np.zeros(80, np.int32), W)
    assert tuple(jt.shape) == (8, 80, n) and tuple(jv.shape) == (8, 80, n)
    assert relerr(rad.cpu().numpy(), eng.limb_rays((ab, em), los).cpu().numpy()) < 1e-13
    # finite differences of the chain: only the perturbed layer's coefficients change
    for k in (1, 25, 60):
        def run(sign):
            a2, e2 = ab.clone(), em.clone()
            Tk = T[k:k + 1] + sign * dT
            ak, ek = ls.abscoeff_layers(Tk, P[k:k + 1], tvib=tv[:, k:k + 1])
            a2[k], e2[k] = ak[0], ek[0]
            return eng.limb_rays((a2, e2), los)
        fd = (run(+1) - run(-1)) / (2 * dT)
        scale = fd.abs().amax(dim=1, keepdim=True).clamp_min(1e-300)
        assert float(((jt[:, k] - fd).abs() / scale).max()) < 1e-5, k
        h = 1e-3 * vmr[k]
        vp, vm_ = vmr.copy(), vmr.copy()
        vp[k] += h
        vm_[k] -= h
        fdv = (eng.limb_rays((ab, em), los_for(vp)) - eng.limb_rays((ab, em), los_for(vm_))) / (2 * h)
        scale = fdv.abs().amax(dim=1, keepdim=True).clamp_min(1e-300)
        assert float(((jv[:, k] - fdv).abs() / scale).max()) < 1e-5, k
    # a ray does not see the layers wholly below its tangent height
    assert float(jt[7, :40].abs().max()) == 0.0 and float(jv[7, :40].abs().max()) == 0.0


def test_config4_two_gas_retrieval_20_iterations(eng):
    """configs[4]: HCN (mol 23) + CH4 on one grid, forward model + analytic Jacobians of both gases' profile
    parameters for all LOS in one launch, Gauss-Newton / LM loop with the reference's stopping rule and at most
    20 iterations (spect_main_module.py:2725-2987).  The Jacobian inside the loop is checked against finite
    differences of the low-resolution forward model first.
    History of the acceptance criterion (an unpinned component: say so): the first version demanded that at least
    four parameters end within 5 % of the a priori of the truth; it failed once on the GPU box in round 2 (two did:
    the lowest nodes sit under an opaque path and are not constrained by the measurement) and was replaced (commit
    0785807) by the statistical statement below -- nodes with averaging kernel > 0.5 end within 4 sigma of the
    truth and closer to it than the a priori -- which is what optimal estimation promises."""
    import bench_configs as bc
    from spectrobot_amd import retrieval
    scene = bc.two_gas_scene(12000, 2500, 24000, 40)
    bs, pixels, x_true = bc.retrieval_problem(scene)
    x0 = bs.param_vector().copy()
    for name in bs.sets:                       # the scene still holds the truth the observations were made from
        scene.gas(name).add_clim(bs.sets[name].profile())
    sims, derivs = retrieval.simulate(scene, pixels, bs)
    for p in (1, 5):
        h = 1e-3 * x0[p]
        outs = []
        for sign in (+1, -1):
            par = bs.params()[p]
            par.value = x0[p] + sign * h
            for name in bs.sets:
                scene.gas(name).add_clim(bs.sets[name].profile())
            outs.append(np.array([s.spectrum for s in retrieval.simulate(scene, pixels, None)[0]]))
            par.value = x0[p]
        fd = (outs[0] - outs[1]) / (2 * h)
        an = np.array([row[p].spectrum for row in derivs])
        assert np.max(np.abs(an - fd)) < 2e-5 * np.max(np.abs(fd)), p
    chi, obs, sims, bs = retrieval.inversion_fast_limb(scene, bs, pixels, max_it=20)
    h = bs.history
    assert bs.stop in ("converged", "raised") and 3 <= len(h) <= 20
    assert h[-1] < 0.2 * h[0] and h[-1] < 2.0               # reduced chi square from >> 1 down to ~1
    bs.update_parerror()
    x, err = bs.param_vector(), np.array([p.ret_error for p in bs.params()])
    well = np.diag(bs.av_kernel) > 0.5        # the nodes the measurement constrains (the lowest ones sit under an opaque path)
    assert well.sum() >= 4
    assert (np.abs(x - x_true)[well] < 4.0 * err[well]).all()
    assert (np.abs(x - x_true)[well] < np.abs(x0 - x_true)[well]).all()


def test_two_ranks_on_one_gpu_gathered_spectrum(eng, tmp_path):
    """The sharded path end to end with TWO fresh processes sharing the one GPU (SR_DIST_BACKEND=gloo: RCCL needs
    one device per rank): each rank computes its spectral shard (halo lines replicated, far-field hierarchy
    anchored at the shard start) and the shards are gathered; rank 0 compares with the shards computed one after
    the other in a single process (bit for bit: the path is deterministic) and with the unsharded run (to
    rounding: a shard's tiles start at its own g_lo).  Equal-width and work-balanced bounds."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "w.py"
    script.write_text("""
import os, sys
import numpy as np
sys.path.insert(0, %r)
import torch
from spectrobot_amd import engine, synthetic as syn, distributed as sd
rank, local, world = sd.init_from_env()
engine.set_device(0)
n = 60000
grid = syn.make_grid(2980.0, 5e-4, n)
rng = np.random.default_rng(1)
L = syn.make_lines(12000, grid, seed=44, n_levels=12)
L["freq"] = np.sort(np.concatenate([rng.uniform(grid[9000], grid[20000], 8000), rng.uniform(grid[0], grid[-1], 4000)]))
atm = syn.make_atmosphere(20, 12)
nd = syn.number_density(atm["press"], atm["temps"])
Lr = syn.limb_los(atm["z"], nd, [np.full(20, 0.0148)], [atm["z"][0] + 3.0, atm["z"][7] + 2.0, atm["z"][15] + 1.0])
los = engine.LimbLOS(Lr["seg_off"], Lr["seg_layer"], Lr["pt_off"], Lr["x"], Lr["nd"], Lr["vmr"], col_scale=[syn.CH4_ISO_RATIO])
ls = engine.LineSet(L, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
def shard(lo, hi):
    co = ls.abscoeff_layers(atm["temps"], atm["press"], tvib=atm["tvib"], g_lo=lo, g_hi=hi)
    return engine.limb_rays(co, los)
for bounds in ([sd.shard_bounds(n, world, r) for r in range(world)], sd.shard_bounds_balanced(L["freq"], grid, world)):
    lo, hi = bounds[rank]
    full = sd.all_gather_spectrum(shard(lo, hi), n, world, rank, bounds=bounds)
    if rank == 0:
        seq = torch.cat([shard(a, b) for a, b in bounds], dim=1)
        assert torch.equal(full, seq), "gathered != shards computed one after the other"
        whole = shard(0, n)
        rel = float(((full - whole).abs() / whole.abs().clamp_min(1e-300)).max())
        assert rel < max(1e-12, engine.far_field_truncation_bound()), rel   # (shards anchor their far-field boxes themselves)
        assert bounds[0][1] != n // 2 or bounds is not None
torch.distributed.barrier()
print("rank", rank, "ok")
""" % root)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2", SR_DIST_BACKEND="gloo",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)


def test_bench_gpus2_launches_itself(eng):
    """The driver's own form, `python bench.py --gpus 2 ...` with no WORLD_SIZE: bench.py starts its two ranks as fresh
    child processes before it touches the GPU (launch_ranks), here both on the one GPU (SR_DIST_BACKEND=gloo: RCCL
    needs a device per rank); exactly one JSON line comes back, with the `dist` record of a two-rank group, a gathered
    spectrum equal to a blocking gather, and exit code 0."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--lines", "20000", "--grid", "40000", "--layers", "16"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["dist"]["world_size"] == 2 and rec["dist"]["backend"] == "gloo"
    assert rec["dist"]["async_equals_blocking"] is True
    assert rec["value"] > 0 and rec["steps"] == 5 and rec["scaling"] == "strong"
    # round 6: the N > 1 line carries what a first real SCALE run needs to be read -- per-rank step times (the value is
    # their MAX), host enqueue time, host time in wait_gathers(), the shards with their line counts and modelled balance
    d = rec["dist"]
    assert len(d["per_rank"]["ms_per_step"]) == 2 and d["per_rank"]["ms_per_step_min"] <= d["per_rank"]["ms_per_step_max"]
    assert abs(d["per_rank"]["ms_per_step_max"] - rec["ms_per_step"]) < 1e-9
    assert len(d["per_rank"]["host_enqueue_ms_per_step"]) == 2 and len(d["per_rank"]["host_ms_in_wait_gathers"]) == 2
    assert d["rank0_ms_in_wait_gathers"] >= 0.0
    sh = d["shards"]
    assert sh["balanced"] is False and sh["bounds"] == [[0, 20000], [20000, 40000]] and sh["points"] == [20000, 20000]
    assert len(sh["lines_prepared"]) == 2 and min(sh["lines_prepared"]) > 10000 and sh["model_cost_max_over_mean"] >= 1.0
    # --balanced: work-balanced bounds (multiples of 64 points), the gather pads to the widest shard
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--balanced",
                        "--lines", "20000", "--grid", "40000", "--layers", "8", "--cpu-seconds", "0"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    rec = json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][-1])
    sh = rec["dist"]["shards"]
    assert sh["balanced"] is True and sh["bounds"][0][0] == 0 and sh["bounds"][1][1] == 40000 and sh["bounds"][0][1] % 64 == 0
    assert rec["dist"]["async_equals_blocking"] is True


def test_bench_gpus2_under_torchrun(eng):
    """The driver's launch form for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr
    127.0.0.1 --master-port P bench.py --gpus 2 ...` -- WORLD_SIZE is set by the launcher, so bench.py does NOT start
    ranks of its own; both ranks on the one GPU (SR_DIST_BACKEND=gloo), one JSON line from rank 0, exit code 0."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(SR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5",
                        "--warmup", "2", "--lines", "20000", "--grid", "40000", "--layers", "16"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=root)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["dist"]["world_size"] == 2 and rec["dist"]["async_equals_blocking"] is True
    assert rec["value"] > 0 and rec["steps"] == 5


def test_rccl_async_gather_branch_one_rank(eng, tmp_path):
    """The RCCL branch of the bench's step on hardware: backend "nccl" with a ONE-rank process group on the one GPU
    (RCCL needs a device per rank, so more ranks cannot be rehearsed here).  Nine steps of coefficient op + limb
    recursion + asynchronous all_gather_spectrum(force_collective=True) into one `out`, the shard tensor freshly
    allocated each step as in bench.py: RCCL initialisation, the lifetime of the Work handles and of their input
    shards (four in flight, older ones waited on), the ordering of consecutive gathers into the same buffer, and
    wait_gathers(); the result equals a blocking gather and the shard itself bit for bit.  Round 4: also the 64-ray
    branch (one collective + the permuting copy), the padded branch of unequal / strided shards and the all-reduce of
    the sharded retrieval, all through RCCL with the one-rank group."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "w.py"
    script.write_text("""
import os, sys
import numpy as np
sys.path.insert(0, %r)
import torch
from spectrobot_amd import engine, synthetic as syn, distributed as sd
rank, local, world = sd.init_from_env(backend="nccl", single_rank_group=True)
assert (rank, world) == (0, 1) and sd.dist_info() == {"backend": "nccl", "world_size": 1, "rank": 0}
engine.set_device(0)
n = 20000
grid = syn.make_grid(2980.0, 5e-4, n)
L = syn.make_lines(3000, grid, seed=45, n_levels=12)
atm = syn.make_atmosphere(12, 12)
nd = syn.number_density(atm["press"], atm["temps"])
Lr = syn.limb_los(atm["z"], nd, [np.full(12, 0.0148)], [atm["z"][0] + 3.0])
los = engine.LimbLOS(Lr["seg_off"], Lr["seg_layer"], Lr["pt_off"], Lr["x"], Lr["nd"], Lr["vmr"], col_scale=[syn.CH4_ISO_RATIO])
ls = engine.LineSet(L, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
full = torch.zeros((1, n), dtype=torch.float64, device="cuda")
ab = torch.empty((12, n), dtype=torch.float64, device="cuda"); em = torch.empty_like(ab)
def step(scale, **kw):
    # the atmosphere changes from step to step, so a gather that read a recycled shard would show
    ls.abscoeff_layers(atm["temps"] + scale, atm["press"], tvib=atm["tvib"] + scale, out=(ab, em))
    rad = engine.limb_rays((ab, em), los)
    return rad, sd.all_gather_spectrum(rad, n, 1, 0, out=full, force_collective=True, **kw)
for i in range(9):
    rad, out = step(0.5 * i, async_op=True)
    assert out is full
    del rad, out
    junk = torch.full((1, n), -1.0, dtype=torch.float64, device="cuda")   # would take over a freed shard's block
    del junk
assert sd.stats["async_gathers"] == 9 and sd.stats["evicted_waits"] == 5 and len(sd._pending) == 4, sd.stats
sd.wait_gathers()
torch.cuda.synchronize()
assert not sd._pending
got = full.clone()
rad, blocking = step(0.5 * 8, async_op=False)
torch.cuda.synchronize()
assert sd.stats["blocking_gathers"] == 1
assert torch.equal(blocking, rad) and torch.equal(got, rad), float((got - rad).abs().max())
assert float(rad.abs().max()) > 0
# round 4: the other branches of all_gather_spectrum and the retrieval's all-reduce, on RCCL too
# (a) a ray batch (configs[2]: 64 rays): equal shards, one collective into a [W, n_rays, q] buffer + the permuting copy
Lr64 = syn.limb_los(atm["z"], nd, [np.full(12, 0.0148)], atm["z"][0] + 1.0 + 1.7 * np.arange(64))
los64 = engine.LimbLOS(Lr64["seg_off"], Lr64["seg_layer"], Lr64["pt_off"], Lr64["x"], Lr64["nd"], Lr64["vmr"], col_scale=[syn.CH4_ISO_RATIO])
rad64 = engine.limb_rays((ab, em), los64)
before = dict(sd.stats)
g64 = sd.all_gather_spectrum(rad64, n, 1, 0, force_collective=True, async_op=True)     # async is ignored for n_rays > 1
torch.cuda.synchronize()
assert g64 is not rad64 and torch.equal(g64, rad64) and sd.stats["blocking_gathers"] == before["blocking_gathers"] + 1
# (b) the padded path (unequal shards of shard_bounds_balanced; here forced by a shard that is a strided view)
wide = torch.zeros((64, n + 7), dtype=torch.float64, device="cuda")
wide[:, :n] = rad64
view = wide[:, :n]
assert not view.is_contiguous()
gp = sd.all_gather_spectrum(view, n, 1, 0, bounds=[(0, n)], force_collective=True)
torch.cuda.synchronize()
assert torch.equal(gp, rad64) and sd.stats["blocking_gathers"] == before["blocking_gathers"] + 2
# wrong bounds are refused before any collective (ADVICE round 3)
for bad in ([(5, n)], [(0, n - 1)], [(0, n), (n, n)]):
    try:
        sd.all_gather_spectrum(rad64, n, 1, 0, bounds=bad, force_collective=True)
        raise SystemExit("bounds {} were accepted".format(bad))
    except ValueError:
        pass
# (c) the all-reduce of a sharded retrieval iteration: [n_los, 1 + n_par, n_bands] partial band integrals
part = torch.arange(18 * 8 * 14, dtype=torch.float64, device="cuda").reshape(18, 8, 14) * 1e-9
want = part.clone()
assert sd.all_reduce_sum(part, force_collective=True) is part
torch.cuda.synchronize()
assert torch.equal(part, want)          # one rank: the sum over the ranks is the rank's own part
torch.distributed.barrier()
torch.distributed.destroy_process_group()
print("rccl one-rank ok", sd.stats)
""" % root)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("SR_DIST_BACKEND", None)
    p = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0 and "rccl one-rank ok" in out, out


def test_retrieval_forward_in_one_call(eng):
    """The iteration's forward model in ONE library call (sr_retrieval_forward_dev: parameter vector -> VMRs on the device
    -> columns -> radiances + Jacobians -> instrument bands -> closed-form FOV per pixel) against the separate calls of
    retrieval.simulate (VMR profiles interpolated on the host, LimbLOS.set_vmr, limb_rays_jacobian, hires_to_lowres,
    smm.fov_closed_form): the same low-resolution spectra and derivatives to rounding (the device sums x_p w_p where the
    host interpolates sum_p mask_p x_p), at the first guess and after a parameter update; the two loops then walk the
    same chi-square history.  Pixels without a field of view (the centre rays), and a spectral shard's partial sums.

    PARITY UNPINNED: checked against other product kernels (the separate calls) only -- the reference's
    radtran_fast is in the absent spect_base_module; the instrument step and FOV closed form inside the call are pinned
    to the reference separately (test_hires_to_lowres_golden, the FOV_integr_1D fixture)."""
    import copy
    import bench_configs as bc
    from spectrobot_amd import retrieval
    scene = bc.two_gas_scene(6000, 1500, 16000, 30)
    bs, pixels, x_true = bc.retrieval_problem(scene)
    bs0 = copy.deepcopy(bs)

    def both_ways(pix, **kw):
        out = []
        for one in (False, True):
            retrieval.ONE_CALL = one
            try:
                out.append(retrieval.simulate(scene, pix, bs, arrays=True, **kw))
            finally:
                retrieval.ONE_CALL = True
        (l0, d0), (l1, d1) = out
        assert l0.shape == l1.shape and d0.shape == d1.shape and d0.shape[1] == len(bs.params())
        assert np.max(np.abs(l1 - l0)) <= 1e-12 * np.max(np.abs(l0))
        assert np.max(np.abs(d1 - d0)) <= 1e-11 * np.max(np.abs(d0))
        return l1, d1

    for name in bs.sets:
        scene.gas(name).add_clim(bs.sets[name].profile())
    both_ways(pixels)
    for k, par in enumerate(bs.params()):          # another point of the parameter space
        par.value *= 1.0 + 0.07 * (k % 3 - 1)
    for name in bs.sets:
        scene.gas(name).add_clim(bs.sets[name].profile())
    whole, _ = both_ways(pixels)
    flat = [retrieval.LimbPixel(p.limb_tg_alt, fov_half=0.0, observation=p.observation, noise=p.noise) for p in pixels]
    both_ways(flat)
    n = len(scene.grid)
    parts = [both_ways(pixels, shard=(lo, hi))[0] for lo, hi in ((0, n // 3), (n // 3, n))]
    assert np.max(np.abs(parts[0] + parts[1] - whole)) <= far_tol(1e-12) * np.max(np.abs(whole))
    hist = []
    for one in (False, True):
        retrieval.ONE_CALL = one
        try:
            _, _, _, b = retrieval.inversion_fast_limb(scene, copy.deepcopy(bs0), pixels, max_it=20)
        finally:
            retrieval.ONE_CALL = True
        hist.append((list(b.history), b.param_vector(), b.stop))
    assert len(hist[0][0]) == len(hist[1][0]) and hist[0][2] == hist[1][2]
    assert np.allclose(hist[0][0], hist[1][0], rtol=1e-9) and np.allclose(hist[0][1], hist[1][1], rtol=1e-8)


def test_config4_retrieval_sharded_over_two_ranks(eng, tmp_path):
    """configs[4] on N GPUs (Jacobian spectra follow the shard pattern, the algebra stays small): two
    fresh processes share the one GPU (gloo), each runs retrieval.inversion_fast_limb on its spectral shard --
    radiances, Jacobians and PARTIAL band integrals, one all-reduce per iteration -- and both must reproduce the
    chi-square history and the retrieved parameters of the unsharded loop (to rounding: the band integrals are summed
    in another order), iteration for iteration."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "w.py"
    script.write_text("""
import os, sys, json
import numpy as np
sys.path.insert(0, %r)
import torch
import bench_configs as bc
from spectrobot_amd import engine, retrieval, distributed as sd
rank, local, world = sd.init_from_env()
engine.set_device(0)
scene = bc.two_gas_scene(6000, 1500, 16000, 30)
bs, pixels, x_true = bc.retrieval_problem(scene)
shard = sd.shard_bounds(len(scene.grid), world, rank) if world > 1 else None
# one simulation first: the sharded forward model against ... itself unsharded (same process, whole grid)
sims_s, der_s = retrieval.simulate(scene, pixels, bs, shard=shard)
chi, obs, sims, bs = retrieval.inversion_fast_limb(scene, bs, pixels, max_it=20, shard=shard)
print("RESULT " + json.dumps({"rank": rank, "hist": [float(c) for c in bs.history], "x": bs.param_vector().tolist(),
                              "stop": bs.stop, "sim0": np.array([s.spectrum for s in sims_s]).tolist(),
                              "der0": np.array([[d.spectrum for d in row] for row in der_s]).tolist()}))
if world > 1:
    torch.distributed.barrier()
""" % root)

    def run(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29551", WORLD_SIZE=str(world), SR_DIST_BACKEND="gloo",
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
        outs = [p.communicate(timeout=900)[0].decode() for p in procs]
        assert all(p.returncode == 0 for p in procs), outs
        import json
        return [json.loads([l for l in o.splitlines() if l.startswith("RESULT ")][-1][7:]) for o in outs]

    whole = run(1)[0]
    two = run(2)
    assert len(whole["hist"]) >= 3
    for r in two:
        assert r["stop"] == whole["stop"] and len(r["hist"]) == len(whole["hist"])
        assert np.allclose(r["sim0"], whole["sim0"], rtol=1e-12, atol=0) and np.allclose(r["der0"], whole["der0"], rtol=1e-9, atol=1e-300)
        assert np.allclose(r["hist"], whole["hist"], rtol=1e-8)
        assert np.allclose(r["x"], whole["x"], rtol=1e-8)
    assert two[0]["hist"] == two[1]["hist"] and two[0]["x"] == two[1]["x"]      # the ranks agree bit for bit


def test_temperature_derivative_schemes(eng):
    """engine.coefficients_dT: finite differences of the coefficient op in T with the region boundaries of the
    perturbed ops frozen at T (sr_lineset_set_bounds_temps).  Reference value: the frozen central difference of
    +-0.01 K.  (i) The default (frozen central, +-0.05 K) agrees with it to ~3e-5 of a layer's largest derivative,
    (ii) rounds 1-3's central difference with boundaries that move with T to ~2e-4 (the seams: 1e-5..1e-4 steps of a
    line's value over 0.1 K), (iii) the two-op forward difference (0.002 K) to ~5e-4 (its truncation against the
    1e-7 staircase of the reference's single-precision cmplx(ry, -rx)).  (iv) Frozen at its own temperatures a call
    reproduces the unfrozen call bit for bit; another layer count than the frozen one is refused."""
    import torch
    from spectrobot_amd import synthetic as syn
    grid = syn.make_grid(2990.0, 5e-4, 30000)
    L = syn.make_lines(20000, grid, seed=5, n_levels=12, config_id=2)
    atm = syn.make_atmosphere(12, 12)
    ls = eng.LineSet(L, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
    T, P, tv = atm["temps"], atm["press"], atm["tvib"]
    co, (da_ref, de_ref) = eng.coefficients_dT(ls, T, P, tvib=tv, scheme="central", dT=0.01)
    _, (da_c, de_c) = eng.coefficients_dT(ls, T, P, tvib=tv, coeffs=co)
    _, (da_m, de_m) = eng.coefficients_dT(ls, T, P, tvib=tv, coeffs=co, frozen=False)
    _, (da_f, de_f) = eng.coefficients_dT(ls, T, P, tvib=tv, coeffs=co, scheme="forward")
    ls.set_bounds_temps(T)
    try:
        a_0, e_0 = ls.abscoeff_layers(T, P, tvib=tv)
        with pytest.raises(RuntimeError):
            ls.abscoeff_layers(T[:5], P[:5], tvib=tv[:, :5])
    finally:
        ls.set_bounds_temps(None)
    assert torch.equal(a_0, co[0]) and torch.equal(e_0, co[1])
    # layer batches (a long LOS: the 3-D path of configs[3]) see their own slice of the boundary temperatures
    ls.set_bounds_temps(T)
    eng.set_table_budget(3 * ls.n_kept * 208)
    try:
        a_b, e_b = ls.abscoeff_layers(T + 0.002, P, tvib=tv)
    finally:
        eng.set_table_budget(48 << 30)
    a_u, e_u = ls.abscoeff_layers(T + 0.002, P, tvib=tv)
    ls.set_bounds_temps(None)
    assert torch.equal(a_b, a_u) and torch.equal(e_b, e_u)

    def rel(x, y):  # per layer, relative to the layer's largest derivative
        return float(((x - y).abs().amax(dim=1) / y.abs().amax(dim=1)).max())
    print("against the frozen central difference of 0.01 K: frozen central 0.05 K %.1e %.1e, moving central %.1e %.1e, "
          "frozen forward 0.002 K %.1e %.1e" % (rel(da_c, da_ref), rel(de_c, de_ref), rel(da_m, da_ref), rel(de_m, de_ref),
                                               rel(da_f, da_ref), rel(de_f, de_ref)))
    assert rel(da_c, da_ref) < 1e-4 and rel(de_c, de_ref) < 1e-4
    assert rel(da_m, da_ref) < 1e-3 and rel(de_m, de_ref) < 1e-3
    assert rel(da_f, da_ref) < 2e-3 and rel(de_f, de_ref) < 2e-3
    assert rel(da_c, da_ref) < rel(da_m, da_ref)      # freezing the seams is what makes the quotient smooth


def test_frozen_boundaries_at_other_temperatures_lose_nothing(eng):
    """sr_lineset_set_bounds_temps with boundary temperatures several K away from the call's own (ADVICE rounds 3, 4):
    the zone of a line is then placed with the widths of Tb while the kernels' candidate bound (widest zone of the
    layer) came from T alone -- for Tb > T a line's outermost region-2 points were never visited.
    (a) With the SAME boundary temperatures every far-field mode (3 = the default, 2, 1) must reproduce the exact mode
    (0: every (line, point) evaluated, the regions placed by the same frozen boundaries) at working precision: a
    dropped outer zone is a 1e-5..1 error there, the far-field truncation 1e-12.  A sparse set (one line per 200
    points: per-line expansions in mode 3) and a dense one (0.4 lines per point: box pairs), the folded op, the
    per-level pair tables (sub-linesets with their own boundary slices) and linearised weights.
    (b) Sanity against the UNFROZEN call of the same mode: what freezing legitimately changes is where the Humlicek
    regions hand over, up to ~1e-2 (region 2 has no Gaussian part and reaches inside |x| + y = 5.5 at Tb = T - 8 K
    and y -> 0); a dropped contribution of a sparse line is O(1)."""
    import torch
    import torch.nn.functional as F
    from spectrobot_amd import synthetic as syn
    grid = syn.make_grid(2990.0, 5e-4, 30000)
    atm = syn.make_atmosphere(6, 12)
    T, P, tv = atm["temps"], atm["press"], atm["tvib"]

    def env_rel(x, y):
        # relative to the local envelope of |y| (+-64 points): under non-LTE populations abs crosses zero
        sc = F.max_pool1d(y.abs().reshape(1, -1, y.shape[-1]), 129, stride=1, padding=64).reshape(y.shape)
        return float(((x - y).abs() / sc.clamp_min(1e-300)).max())

    try:
        for n_lines, seed in ((150, 11), (12000, 12)):
            L = syn.make_lines(n_lines, grid, seed=seed, n_levels=12, config_id=2)
            ls = eng.LineSet(L, grid, 6, 1, syn.CH4_MM, syn.CH4_LEVEL_ENERGIES)
            for dTb in (8.0, -8.0, 25.0):
                for linear in (False, True):
                    if linear and dTb != 8.0:
                        continue
                    res = {}
                    for mode in (0, 3, 2, 1):
                        eng.set_far_field(mode)
                        ls.set_bounds_temps(T + dTb, linear_weights=linear)
                        try:
                            res[mode] = ls.abscoeff_layers(T, P, tvib=tv) + (ls.glevel_pairs(T, P),)
                        finally:
                            ls.set_bounds_temps(None)
                    for mode in (3, 2, 1):
                        ra, re = env_rel(res[mode][0], res[0][0]), env_rel(res[mode][1], res[0][1])
                        rp = env_rel(res[mode][2], res[0][2])
                        print("%d lines, boundaries at T%+.0f K%s: far-field mode %d vs exact mode, same boundaries: abs %.1e emi "
                              "%.1e pair tables %.1e" % (n_lines, dTb, ", linear weights" if linear else "", mode, ra, re, rp))
                        # (relative to the envelope of the NET spectrum: under non-LTE populations -- and more so under
                        # linearised weights -- the lines' weights nearly cancel, and the truncation bound, which is relative
                        # to a line's own contribution, is seen amplified: 19 x measured at degree 20; a dropped zone is 1e-5..1)
                        tol = far_tol(1e-11, amp=32.0)
                        assert ra < tol and re < tol and rp < tol, (n_lines, dTb, linear, mode)
                    if not linear:
                        for mode in (0, 3):
                            eng.set_far_field(mode)
                            a0, e0 = ls.abscoeff_layers(T, P, tvib=tv)
                            ra, re = env_rel(res[mode][0], a0), env_rel(res[mode][1], e0)
                            assert 0.0 < ra < 5e-2 and 0.0 < re < 5e-2, (n_lines, dTb, mode, ra, re)
            ls.close()
    finally:
        eng.set_far_field(eng.FAR_FIELD_DEFAULT)


def test_config3_3d_path_per_step_state(eng, oracle):
    """configs[3] in its 3-D form (radtran_3Dvs2D_sza30-80_test.py:353-379, use_tangent_sza = False): a coefficient
    row per LOS step with (P, T, T_vib) at the step's own local SZA, Jacobians per ALTITUDE layer.  (a) With a state
    that does not depend on the SZA the 3-D path must reproduce the 1-D one -- radiances and both Jacobians; (b) with
    it, the far and the near side of a ray differ and the coefficient rows of sampled steps match the oracle;
    (c) the per-altitude temperature Jacobian against finite differences of the whole chain (the temperature of one
    altitude layer changed in every step that crosses it)."""
    import torch
    import bench_configs as bc
    from spectrobot_amd import synthetic as syn
    n, nl = 20000, 30
    grid, L, atm, e_lev = bc.ch4_case(n, n, nl, config_id=3, w0=2950.0)
    ls = eng.LineSet(L, grid, 6, 1, syn.CH4_MM, e_lev)
    vm = np.full(nl, 0.0148)
    tz = np.array([atm["z"][2] + 7.0, atm["z"][9] + 3.0, atm["z"][17] + 11.0])
    az = np.array([0.0, 60.0, 150.0])
    L3 = syn.limb_los_3d(atm["z"], atm["nd"], [vm], tz, 65.0, az)
    L1 = syn.limb_los(atm["z"], atm["nd"], [vm], tz)
    assert np.array_equal(L3["seg_alt_layer"], L1["seg_layer"]) and np.array_equal(L3["seg_layer"], np.arange(len(L1["seg_layer"])))
    # mu: the tangent point sees SZA_t, the two ends of a ray differ unless it runs across the sun's direction
    s0, s1 = L3["seg_off"][0], L3["seg_off"][1]
    assert abs(L3["seg_mu"][(s0 + s1) // 2] - np.cos(np.deg2rad(65.0))) < 0.08 and abs(L3["seg_mu"][s0] - L3["seg_mu"][s1 - 1]) > 0.1
    W = bc.layer_vmr_weights(atm["z"], L1["alt"])
    pg = np.zeros(nl, np.int32)
    mk = lambda Lx: eng.LimbLOS(Lx["seg_off"], Lx["seg_layer"], Lx["pt_off"], Lx["x"], Lx["nd"], Lx["vmr"], col_scale=[syn.CH4_ISO_RATIO])
    los3, los1 = mk(L3), mk(L1)
    dT = 0.05

    def coef3(a):
        co = ls.abscoeff_layers(a["temps"], a["press"], tvib=a["tvib"])
        ap = ls.abscoeff_layers(a["temps"] + dT, a["press"], tvib=a["tvib"])
        am = ls.abscoeff_layers(a["temps"] - dT, a["press"], tvib=a["tvib"])
        return co, ((ap[0] - am[0]) / (2 * dT), (ap[1] - am[1]) / (2 * dT))

    def close(x, y, tol):
        # relative to the largest entry of the ray's whole Jacobian: rows of level parameters whose weight at the
        # ray's sample points is a rounding residue (1e-16 at a shell boundary) hold nothing but noise
        sc = y.abs().reshape(y.shape[0], -1).amax(dim=1).clamp_min(1e-300).reshape((-1,) + (1,) * (y.dim() - 1))
        return float(((x - y).abs() / sc).max()) < tol

    # (a) SZA-independent state: 3-D == 1-D (the coefficient rows agree to an ulp or two, their differences in T
    # carry that noise divided by dT)
    a1 = bc.sza_atmosphere(atm, 60.0)
    a3 = bc.step_atmosphere(atm, L3["seg_alt_layer"], np.full(len(L3["seg_mu"]), np.cos(np.deg2rad(60.0))))   # the same mu everywhere
    co1, dco1 = coef3(a1)
    co3, dco3 = coef3(a3)
    r1, jt1, jv1 = eng.limb_rays_jacobians(co1, los1, dcoeffs=dco1, par_gas=pg, par_w=W)
    r3, jt3, jv3 = eng.limb_rays_jacobians(co3, los3, dcoeffs=dco3, par_gas=pg, par_w=W, seg_jac_row=L3["seg_alt_layer"], n_jac_rows=nl)
    assert tuple(jt3.shape) == (3, nl, n)
    assert close(r3, r1, 1e-12) and close(jt3, jt1, 1e-11) and close(jv3, jv1, 1e-11)
    # (b) the real 3-D state
    a3 = bc.step_atmosphere(atm, L3["seg_alt_layer"], L3["seg_mu"])
    co3, dco3 = coef3(a3)
    sel = np.array([s0, s0 + 5, s1 - 1, L3["seg_off"][2] + 3])
    abo, emo = oracle.abscoeff_layers(L, syn.CH4_MM, e_lev, a3["temps"][sel], a3["press"][sel], _q(a3["temps"][sel]),
                                      a3["tvib"][:, sel], grid, mode=1, n_threads=4)
    assert relerr(co3[0][sel].cpu().numpy(), abo) < 1e-10 and relerr(co3[1][sel].cpu().numpy(), emo) < 1e-10
    assert L3["seg_alt_layer"][s0] == L3["seg_alt_layer"][s1 - 1] and relerr(co3[1][s0].cpu().numpy(), co3[1][s1 - 1].cpu().numpy()) > 1e-3
    r3, jt3, jv3 = eng.limb_rays_jacobians(co3, los3, dcoeffs=dco3, par_gas=pg, par_w=W, seg_jac_row=L3["seg_alt_layer"], n_jac_rows=nl)
    assert not close(r3, r1, 1e-4)
    # (c) per-altitude temperature Jacobian vs finite differences through the coefficient op
    for k in (4, 12, 22):
        steps = np.nonzero(L3["seg_alt_layer"] == k)[0]
        def run(sign):
            a2, e2 = co3[0].clone(), co3[1].clone()
            ak, ek = ls.abscoeff_layers(a3["temps"][steps] + sign * dT, a3["press"][steps], tvib=a3["tvib"][:, steps])
            a2[steps], e2[steps] = ak, ek
            return eng.limb_rays((a2, e2), los3)
        fd = (run(+1) - run(-1)) / (2 * dT)
        # (a deep layer of an opaque ray contributes e^-500 of the radiance: the difference quotient is exactly 0
        # there, the analytic value 1e-224; both are "nothing" on the scale of the ray's radiance per kelvin)
        sc = torch.maximum(fd.abs().amax(dim=1, keepdim=True), 1e-9 * r3.abs().amax(dim=1, keepdim=True))
        err = float(((jt3[:, k] - fd).abs() / sc).max())
        assert err < 1e-5, (k, err)
        assert float(fd[2].abs().max()) > 0 or k < 17
    assert float(jt3[2, :17].abs().max()) == 0.0


def test_config3_3d_level_factored_route(eng, oracle):
    """configs[3], 3-D form, as bench.py --config 3 --3d runs it since round 4: kinetic T on (latitude box, altitude)
    (radtran_3Dvs2D_sza30-80_test.py:66-90), T_vib by the local SZA.  The level-factored route -- pair tables on the
    DISTINCT (P, T) rows the rays touch, one combine for all steps (spect_main_module.py:2036-2106) -- against the
    folded coefficient op run on every step (<= 1e-12 of a row's largest value), against the oracle on sampled steps,
    and the radiances / Jacobians through both sets of coefficient rows."""
    import torch
    import bench_configs as bc
    from spectrobot_amd import synthetic as syn
    n, nl = 20000, 30
    grid, L, atm, e_lev = bc.ch4_case(n, n, nl, config_id=3, w0=2950.0)
    ls = eng.LineSet(L, grid, 6, 1, syn.CH4_MM, e_lev)
    vm = np.full(nl, 0.0148)
    tz = np.array([atm["z"][2] + 7.0, atm["z"][9] + 3.0, atm["z"][17] + 11.0])
    az = np.array([0.0, 60.0, 150.0])
    sets = [bc.los_3d_set(atm, vm, tz, sza, az) for sza in (37.0, 72.0)]
    # the north-south ray leaves the equatorial box, the steps of a ray see different suns
    assert len(np.unique(sets[0]["seg_box"])) >= 2
    a = sets[0]["state"]
    T_all = np.concatenate([S["state"]["temps"] for S in sets])
    P_all = np.concatenate([S["state"]["press"] for S in sets])
    tv_all = np.concatenate([S["state"]["tvib"] for S in sets], axis=1)
    T_rows, P_rows, row = eng.LevelFactored.unique_rows(T_all, P_all)
    assert len(T_rows) < len(row) // 3 and np.array_equal(T_rows[row], T_all) and np.array_equal(P_rows[row], P_all)
    dT = 0.002
    lf = eng.LevelFactored(ls, T_rows, P_rows, dT=dT, linear_weights=False)     # exact weights: the folded op's own quotient
    (ca, ce), (da, de) = lf.steps(row, tvib=tv_all, derivative=True)
    co = ls.abscoeff_layers(T_all, P_all, tvib=tv_all)
    sa, se = co[0].abs().amax(dim=1, keepdim=True), co[1].abs().amax(dim=1, keepdim=True)
    assert float(((ca - co[0]).abs() / sa).max()) < 1e-12 and float(((ce - co[1]).abs() / se).max()) < 1e-12
    sel = np.array([0, 7, len(sets[0]["seg_layer"]) - 1, len(row) - 3])
    abo, emo = oracle.abscoeff_layers(L, syn.CH4_MM, e_lev, T_all[sel], P_all[sel], _q(T_all[sel]), tv_all[:, sel], grid, mode=1, n_threads=4)
    assert float(np.max(np.abs(ca[sel].cpu().numpy() - abo) / np.abs(abo).max(axis=1, keepdims=True))) < 1e-11
    assert relerr(ce[sel].cpu().numpy(), emo) < 1e-10
    # the folded op's own two-op derivative (same dT, same frozen boundaries): equal up to the population part, which
    # the combine differentiates exactly and the difference quotient to first order (dT / 2 pop'' / pop' ~ 1e-5)
    _, (da_f, de_f) = eng.coefficients_dT(ls, T_all, P_all, tvib=tv_all, coeffs=co, scheme="forward", dT=dT)
    rel = lambda x, y: float(((x - y).abs().amax(dim=1) / y.abs().amax(dim=1)).max())
    assert rel(da, da_f) < 2e-4 and rel(de, de_f) < 2e-4
    # radiances and Jacobians of the first set through both
    S = sets[0]
    ns = len(S["seg_layer"])
    los = eng.LimbLOS(S["seg_off"], S["seg_layer"], S["pt_off"], S["x"], S["nd"], S["vmr"], col_scale=[syn.CH4_ISO_RATIO])
    W = bc.layer_vmr_weights(atm["z"], S["alt"])
    pg = np.zeros(nl, np.int32)
    jac = lambda c, d: eng.limb_rays_jacobians(c, los, dcoeffs=d, par_gas=pg, par_w=W, seg_jac_row=S["seg_alt_layer"], n_jac_rows=nl)
    r1, jt1, jv1 = jac((ca[:ns], ce[:ns]), (da[:ns], de[:ns]))
    r2, jt2, jv2 = jac((co[0][:ns], co[1][:ns]), (da[:ns], de[:ns]))
    sc = lambda y: y.abs().reshape(y.shape[0], -1).amax(dim=1).clamp_min(1e-300).reshape((-1,) + (1,) * (y.dim() - 1))
    assert float(((r1 - r2).abs() / sc(r2)).max()) < 1e-11 and float(((jt1 - jt2).abs() / sc(jt2)).max()) < 1e-10
    assert float(((jv1 - jv2).abs() / sc(jv2)).max()) < 1e-10


def test_parsed_hitran_file_through_the_hip_path(eng, oracle):
    """N3 end to end: read_line_database (spect_classes.py:1532-1601, HITRAN-2012 160-column records) ->
    lines_to_soa -> LineSet.abscoeff_layers against the oracle on the same parsed lines.  The fixture (written by the
    reference's own Print_hitran) holds a CH4 line with air broadening 0.000: the reader's default of 0.05
    (spect_classes.py:1578-1581) is what both sides must see; a zero Lorentz width would put ry = 0 into humliv_bb."""
    from spectrobot_amd import spect_classes as spcl, synthetic as syn
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "tests", "golden", "hitran_sample.par")
    raw = [float(ln[35:40]) for ln in open(path) if ln[:3] == " 61"]
    lines = spcl.read_line_database(path, mol=6, iso=1)
    assert len(lines) == len(raw) and min(raw) == 0.0                 # the file holds a zero air-broadening record ...
    soa = spcl.lines_to_soa(lines)
    assert soa["air_broad"].min() > 0.0 and np.any((np.array(raw) == 0.0) & (soa["air_broad"] == 0.05))   # ... the reader's default replaces it
    grid = syn.make_grid(2900.0, 1e-3, 201000)
    atm = syn.make_atmosphere(5, 0)
    T, P = atm["temps"], atm["press"] * np.array([30.0, 3.0, 1.0, 1.0, 1.0])
    ls = eng.LineSet(soa, grid, 6, 1, syn.CH4_MM, [])
    ab, em = ls.abscoeff_layers(T, P)
    abo, emo = oracle.abscoeff_layers(soa, syn.CH4_MM, np.zeros(0), T, P, _q(T), None, grid, mode=1, n_threads=4)
    assert abo.max() > 0 and relerr(ab.cpu().numpy(), abo) < 1e-10 and relerr(em.cpu().numpy(), emo) < 1e-10
    # the default matters: with the file's literal zero the widest-pressure layer differs at the 1e-2 level at that line
    k = int(np.argmax(np.array(raw) == 0.0))
    soa0 = dict(soa, air_broad=np.where(np.arange(len(raw)) == k, 1e-6, soa["air_broad"]))
    ab0, _ = eng.LineSet(soa0, grid, 6, 1, syn.CH4_MM, []).abscoeff_layers(T, P)
    assert relerr(ab0[0].cpu().numpy(), abo[0]) > 1e-3


def test_config0_co_nadir_and_slant_radiance(eng, oracle):
    """BASELINE configs[0] (radtran_test_CO.py: one CO band, ~500 lines, 1e4-point grid, 40 layers, no level table):
    the coefficients of all 40 layers against the oracle, then nadir and slant paths (synthetic.slant_los) over a
    Planck surface against the oracle's recursion, and Kirchhoff's law: an isothermal LTE atmosphere over a
    surface of the same temperature emits the Planck function, whatever its optical depth (emi / abs of a line is
    B at the line centre: 3 |nu - nu0| / nu0 <= 5e-3 off in the far wings)."""
    import torch
    from spectrobot_amd import synthetic as syn
    n, nl = 10000, 40
    grid = syn.make_grid(2100.0, 5e-4, n)
    L = syn.make_lines(500, grid, config_id=1, n_levels=0, co_like=True)
    atm = syn.make_atmosphere(nl, 0)
    nd = syn.number_density(atm["press"], atm["temps"])
    ls = eng.LineSet(L, grid, 5, 1, syn.CO_MM, [])
    T, P = atm["temps"], atm["press"]
    ab, em = ls.abscoeff_layers(T, P)
    abo, emo = oracle.abscoeff_layers(L, syn.CO_MM, np.zeros(0), T, P, _q_of(5, 1, T), None, grid, mode=1, n_threads=4)
    assert relerr(ab.cpu().numpy(), abo) < 1e-10 and relerr(em.cpu().numpy(), emo) < 1e-10
    vmr = np.full(nl, 5e-5)
    Ls = syn.slant_los(atm["z"], nd, [vmr], [0.0, 30.0, 60.0, 75.0])
    assert np.array_equal(Ls["seg_layer"][:nl], np.arange(nl)) and Ls["seg_off"].tolist() == [0, nl, 2 * nl, 3 * nl, 4 * nl]
    # path lengths: vertical = the shell thickness, slant longer by 1 / cos(zenith) at the bottom, less higher up (Titan's atmosphere is a third of its radius deep)
    dz = np.diff(np.append(atm["z"], atm["z"][-1] + (atm["z"][-1] - atm["z"][-2]))) * 1e5
    seg_len = Ls["x"][Ls["pt_off"][1:] - 1] - Ls["x"][Ls["pt_off"][:-1]]
    assert np.allclose(seg_len[:nl], dz, rtol=1e-12) and np.all(seg_len[2 * nl:3 * nl] > 1.3 * dz) and np.all(seg_len[2 * nl:3 * nl] < 2.0 * dz + 1.0) and abs(seg_len[2 * nl] / dz[0] - 2.0) < 0.05
    t_surf = 160.0
    los = eng.LimbLOS(Ls["seg_off"], Ls["seg_layer"], Ls["pt_off"], Ls["x"], Ls["nd"], Ls["vmr"], initial_temperature=t_surf)
    rad = eng.limb_rays((ab, em), los, grid=grid).cpu().numpy()
    col = np.array([oracle.curgod(2, Ls["nd"][a:b], Ls["x"][a:b], vmr=Ls["vmr"][0][a:b]) for a, b in zip(Ls["pt_off"][:-1], Ls["pt_off"][1:])])
    bb = np.array([oracle.calc_bb_single(nu, t_surf) for nu in grid])
    for r in range(4):
        sl = slice(Ls["seg_off"][r], Ls["seg_off"][r + 1])
        ro = oracle.radiance_ray(abo, emo, Ls["seg_layer"][sl], col[sl], rad0=bb.copy())
        assert relerr(rad[r], ro) < 1e-10, r
    assert np.any(np.abs(rad[3] / rad[0] - 1.0) > 1e-3)   # the slant path sees more gas
    # Kirchhoff: isothermal atmosphere, surface at the same temperature, a column thick enough to saturate the cores
    Tiso = np.full(nl, 150.0)
    ab2, em2 = ls.abscoeff_layers(Tiso, P)
    Lk = syn.slant_los(atm["z"], nd, [vmr], [0.0, 60.0])
    losk = eng.LimbLOS(Lk["seg_off"], Lk["seg_layer"], Lk["pt_off"], Lk["x"], Lk["nd"], Lk["vmr"], initial_temperature=150.0)
    rk = eng.limb_rays((ab2, em2), losk, grid=grid).cpu().numpy()
    bk = np.array([oracle.calc_bb_single(nu, 150.0) for nu in grid])
    tau0 = (ab2.cpu().numpy() * np.array([oracle.curgod(2, Lk["nd"][a:b], Lk["x"][a:b], vmr=Lk["vmr"][0][a:b])
                                          for a, b in zip(Lk["pt_off"][:nl], Lk["pt_off"][1:nl + 1])])[:, None]).sum(axis=0)
    assert tau0.max() > 50.0 and tau0.min() < 1.0          # saturated cores and thin windows in one spectrum
    assert np.max(np.abs(rk / bk - 1.0)) < 5e-3


def test_inversion_first_driver_direct_and_lut_route(eng):
    """spect_main_module.inversion (:2422-2595), the per-pixel driver of radtran_test_CO.py: the loop with the
    coefficients computed directly and through look-up tables (useLUTs=True, the reference's default: tables on a
    (P, T) lattice in HBM, bilinear interpolation per layer).  Both reduce chi square from >> 1 to ~1 and agree
    with each other within the interpolation error of the tables; the direct route agrees with the fast loop's
    first iterations (same forward model, chi square normalised with n_tot instead of the parameters in use)."""
    import copy
    import bench_configs as bc
    from spectrobot_amd import retrieval
    scene = bc.two_gas_scene(5000, 1200, 12000, 30)
    bs0, pixels, x_true = bc.retrieval_problem(scene, n_pix=4)
    res = {}
    for lut in (False, True):
        bs = copy.deepcopy(bs0)
        assert retrieval.inversion(scene, bs, pixels, max_it=12, useLUTs=lut, LUTopt=dict(temp_step=2.5, pres_step_log=0.5)) is None
        res[lut] = bs
        h = bs.history
        assert bs.stop in ("converged", "raised") and len(h) >= 3 and h[-1] < 0.2 * h[0] and h[-1] < 3.0, (lut, h)
    assert abs(res[True].history[0] / res[False].history[0] - 1.0) < 0.05          # interpolation error of the tables
    assert np.allclose(res[True].param_vector(), res[False].param_vector(), rtol=0.2)
    bsf = copy.deepcopy(bs0)
    retrieval.inversion_fast_limb(scene, bsf, pixels, max_it=12)
    n_obs = sum(len(p.observation.spectrum) for p in pixels)
    # same forward model: chi^2 (n_obs - n_used) = chi^2' (n_obs - n_tot); every parameter is in use here
    assert np.allclose(bsf.history[:2], res[False].history[:2], rtol=1e-9)


def test_group_observations_route(eng):
    """The reference's group_observations route (spect_main_module.py:2668-2670, 2908-2930, 3263-3273) through the GPU
    path: radtrans / simulate on a ladder of tangent altitudes (smm.make_group_observations), the pixels' three LOS
    read off quadratic splines in altitude (smm.make_radtran_spline -- both pinned to the reference by
    tests/golden/group_obs.npz, test_group_observations_golden), then the usual FOV integration.
      * EXACTLY the spline of the coarse set: the same spectra as the spline + FOV steps applied by hand to a plain
        simulation of the ladder's rays;
      * against the all-pixels run within the spline's own error: halving the ladder's step shrinks the deviation,
        and it is ~1e-2 of a spectrum's largest band at the reference's default 50 km;
      * derivatives too (inversion_fast_limb's deriv_splines), and the retrieval loop runs on the route."""
    import copy
    import bench_configs as bc
    from spectrobot_amd import retrieval, spect_main_module as smm
    scene = bc.two_gas_scene(5000, 1200, 12000, 30)
    bs, pixels, x_true = bc.retrieval_problem(scene, n_pix=5)
    pixels = sorted(pixels, key=lambda p: p.limb_tg_alt)
    direct = np.array([s.spectrum for s in retrieval.radtrans(scene, pixels)])
    dev = {}
    for step in (50.0, 25.0):
        got = np.array([s.spectrum for s in retrieval.radtrans(scene, pixels, group_observations=True, alt_step_sims=step)])
        # by hand: the ladder's rays as pixels without a field of view, spline, closed-form FOV
        alts, _ = smm.make_group_observations(list(pixels), alt_step=step)
        ladder = [retrieval.LimbPixel(a) for a in alts]
        coarse = np.array([s.spectrum for s in retrieval.radtrans(scene, ladder)])
        f = smm.make_radtran_spline(alts, coarse)
        for i, pix in enumerate(pixels):
            s3 = [f(a) for a in pix.los_alts()]
            want = smm.fov_closed_form(s3[0], s3[1], s3[2], pix.pixel_rot)
            assert np.max(np.abs(got[i] - want)) <= 1e-12 * np.max(np.abs(want)), (step, i)
        dev[step] = float(np.max(np.abs(got - direct) / np.max(np.abs(direct), axis=1, keepdims=True)))
    # (1.1e-2 / 7.5e-3 of a spectrum's largest band here: the radiance of a 30-shell atmosphere has a kink in tangent
    # altitude at every shell boundary, 27 km apart, which no spline through a coarser ladder follows)
    assert dev[25.0] < dev[50.0] and dev[50.0] < 2e-2 and dev[25.0] < 1e-2, dev
    # derivatives through their own splines, and the loop on the route: same stopping behaviour, parameters close to the
    # all-pixels retrieval's
    sims_d, der_d = retrieval.simulate(scene, pixels, bs)
    sims_g, der_g = retrieval.simulate(scene, pixels, bs, group=(25.0, None))
    for p in range(len(der_d[0])):
        # (relative to the parameter's largest derivative over the pixels: where a parameter's mask does not reach a
        # pixel's rays the direct derivative is exactly 0 and the spline rings at ~1e-7 of its neighbours)
        scale = max(np.max(np.abs(der_d[i][p].spectrum)) for i in range(len(pixels)))
        for i in range(len(pixels)):
            # (8 % at worst here: a derivative follows its parameter's triangular mask, kinked at the nodes, which a
            # 25 km ladder samples coarsely -- the route's own approximation, as in the reference)
            assert np.max(np.abs(der_g[i][p].spectrum - der_d[i][p].spectrum)) < 0.15 * scale, (i, p)
    r_d = retrieval.inversion_fast_limb(scene, copy.deepcopy(bs), pixels, max_it=8)
    r_g = retrieval.inversion_fast_limb(scene, copy.deepcopy(bs), pixels, max_it=8, group_observations=True, alt_step_sims=25.0)
    x_d, x_g = r_d[3].param_vector(), r_g[3].param_vector()
    # the loop runs on the route and descends; its forward model differs from the all-pixels one by the spline's ~1 %
    # (2.5 x the noise of these observations), so the two retrievals agree only roughly
    hist = r_g[3].history
    assert r_g[3].stop in ("converged", "raised", "max_it") and len(hist) >= 2 and np.all(np.isfinite(hist)) and min(hist) < hist[0]
    assert np.all(x_g > 0.5 * x_d) and np.all(x_g < 2.0 * x_d)


def test_retrieval_step_in_one_call(eng):
    """Round 6: an iteration of the retrieval loop in ONE library call (sr_retrieval_step_dev: the forward model of
    sr_retrieval_forward_dev, chi square against the observations, and the Levenberg-Marquardt step of the optimal-
    estimation algebra -- spect_main_module.inversion_algebra, :3433-3469 -- in the library's host code).
      * its algebra against smm.inversion_algebra_arrays (the mirror pinned to the reference's fixture,
        test_inversion_algebra_golden) on the same spectra and Jacobians, with and without a mask: dx, S_x, AVK, chi;
      * the loop on the route against the loop with the numpy algebra: the same chi-square history, stop and parameters.
    The forward model inside is checked against the separate calls by test_retrieval_forward_in_one_call (PARITY
    UNPINNED for the radiances)."""
    import copy
    import bench_configs as bc
    from spectrobot_amd import retrieval, spect_main_module as smm
    scene = bc.two_gas_scene(6000, 1500, 16000, 30)
    bs, pixels, x_true = bc.retrieval_problem(scene)
    pixels = sorted(pixels, key=lambda p: p.limb_tg_alt)
    rng = np.random.default_rng(3)
    for use_mask in (False, True):
        b = copy.deepcopy(bs)
        for name in b.sets.keys():
            scene.gas(name).add_clim(b.sets[name].profile())
        masks = None
        if use_mask:
            masks = [rng.random(len(scene.bands_nm)) > 0.25 for _ in pixels]
        obs = np.concatenate([p.observation.spectrum for p in pixels])
        noi = np.concatenate([p.noise.spectrum for p in pixels])
        masktot = None if masks is None else np.concatenate(masks)
        Sa_inv = np.linalg.inv(np.asarray(b.VCM_apriori(), dtype=float))
        oe = eng.OeProblem(obs, noi, masktot, Sa_inv, b.apriori_vector(), 0.1)
        alts = [a for pix in pixels for a in pix.los_alts()]
        los, par_gas, par_w = retrieval._one_call_batch(scene, pixels, b, alts, len(pixels))
        both, chi_sum, n_used, dx, S_x, AVK, _ = eng.retrieval_step(scene.coefficient_stack(), los, par_gas, par_w, b.param_vector(),
                                                                    scene.grid, scene.bands_nm, scene.widths_nm, oe, fov=scene._fov_fac)
        low, dlow = both[:, 0, :], both[:, 1:, :]
        n_par = dlow.shape[1]
        jac = np.transpose(dlow, (1, 0, 2)).reshape(n_par, -1)
        jac = (jac if masktot is None else jac[:, masktot]).T
        sel = slice(None) if masktot is None else masktot
        sim_vec, obs_vec, noi_vec = low.reshape(-1)[sel], obs[sel], noi[sel]
        assert n_used == obs_vec.size
        assert abs(chi_sum - np.sum(((obs_vec - sim_vec) / noi_vec) ** 2)) <= 1e-12 * chi_sum
        x0 = np.array(b.param_vector())
        ref = copy.deepcopy(b)
        smm.inversion_algebra_arrays(jac, obs_vec, sim_vec, noi_vec, ref, lambda_LM=0.1, Sa_inv=Sa_inv)
        # the mirror applied the update with its positivity rule: compare through the same rule
        chk = copy.deepcopy(b)
        chk.update_params(dx)
        assert np.max(np.abs(np.array(chk.param_vector()) - np.array(ref.param_vector())) / np.abs(x0)) < 1e-9, use_mask
        S_ref = np.linalg.inv(jac.T / noi_vec ** 2 @ jac + Sa_inv)
        assert np.max(np.abs(S_x - S_ref)) <= 1e-8 * np.max(np.abs(S_ref))
        assert np.max(np.abs(AVK - S_ref @ (jac.T / noi_vec ** 2 @ jac))) <= 1e-8
    hist = {}
    try:
        # the loop in one call, an iteration per call, the algebra in numpy; then a loop that max_it ends
        for key, (loop, on, max_it) in dict(loop=(True, True, 20), step=(False, True, 20), numpy=(False, False, 20),
                                             loop3=(True, True, 3), step3=(False, True, 3)).items():
            retrieval.LOOP_IN_ONE_CALL, retrieval.STEP_IN_ONE_CALL = loop, on
            r = retrieval.inversion_fast_limb(scene, copy.deepcopy(bs), pixels, max_it=max_it)
            b = r[3]
            hist[key] = (np.array(b.history), np.array(b.param_vector()), b.stop, np.array(b.av_kernel), np.array(b.VCM),
                         np.array(b.old_params, dtype=float), np.array([s.spectrum for s in r[2]]), np.array(b.jacobian),
                         [len(p.old_values) for p in b.params()], float(r[0]))
    finally:
        retrieval.STEP_IN_ONE_CALL = retrieval.LOOP_IN_ONE_CALL = True
    assert hist["step"][2] == hist["numpy"][2] and len(hist["step"][0]) == len(hist["numpy"][0]) >= 3
    assert np.max(np.abs(hist["step"][0] - hist["numpy"][0]) / hist["numpy"][0]) < 1e-9
    assert np.max(np.abs(hist["step"][1] - hist["numpy"][1]) / np.abs(hist["numpy"][1])) < 1e-8
    # the loop in the library runs the same calls in the same order as the loop in Python: the same numbers
    for a, c in (("loop", "step"), ("loop3", "step3")):
        assert hist[a][2] == hist[c][2] and hist[a][8] == hist[c][8] and hist[a][9] == hist[c][9]
        for q in (0, 1, 3, 4, 5, 6, 7):
            assert hist[a][q].shape == hist[c][q].shape and np.array_equal(hist[a][q], hist[c][q]), (a, q)
    assert hist["loop3"][2] == "max_it" and len(hist["loop3"][0]) == 3 and hist["loop3"][5].shape[0] == 3
    # the positivity rule (:616-624) inside the library: first guesses so small that the first steps would cross zero --
    # sr_retrieval_loop_dev against a loop of sr_retrieval_step_dev calls with the rule applied here; and its refusals
    b = copy.deepcopy(bs)
    for name in b.sets.keys():
        scene.gas(name).add_clim(b.sets[name].profile())
    obs = np.concatenate([p.observation.spectrum for p in pixels])
    noi = np.concatenate([p.noise.spectrum for p in pixels])
    oe = eng.OeProblem(obs, noi, None, np.linalg.inv(np.asarray(b.VCM_apriori(), dtype=float)), b.apriori_vector(), 0.1)
    alts = [a for pix in pixels for a in pix.los_alts()]
    los, par_gas, par_w = retrieval._one_call_batch(scene, pixels, b, alts, len(pixels))
    cst = scene.coefficient_stack()
    n_par = len(par_gas)
    x0 = np.array(b.param_vector(), dtype=float) * 3.0                      # far above the truth: large negative first steps
    pos = np.ones(n_par, bool)
    pos[-1] = False                                                         # one parameter free to go negative
    x, hist_py, xs, halved, chi_old, stop_py = x0.copy(), [], [x0.copy()], 0, None, ""
    for it in range(5):
        both, chi_sum, n_used, dx, S_x, AVK, _ = eng.retrieval_step(cst, los, par_gas, par_w, x, scene.grid, scene.bands_nm,
                                                                    scene.widths_nm, oe, fov=scene._fov_fac)
        chi = chi_sum / (n_used - n_par)
        hist_py.append(chi)
        stop_py = smm.retrieval_converged(chi, chi_old, 0.01)
        if stop_py:
            break
        chi_old = chi
        for p in range(n_par):
            d = dx[p]
            if pos[p]:
                while x[p] + d <= 0.0:
                    d /= 2
                    halved += 1
            x[p] = x[p] + d
        xs.append(x.copy())
    out_c, hist_c, xh_c, stop_c, S_c, A_c, _ = eng.retrieval_loop(cst, los, par_gas, par_w, x0, scene.grid, scene.bands_nm, scene.widths_nm,
                                                                  oe, pos, n_par, chi_threshold=0.01, max_it=5, fov=scene._fov_fac)
    assert halved > 0, "the case must exercise the rule"
    assert stop_c == stop_py and np.array_equal(hist_c, np.array(hist_py)) and np.array_equal(xh_c, np.array(xs))
    assert np.all(xh_c[:, :-1] > 0.0)
    with pytest.raises(Exception):                                          # a constrained parameter that is not positive
        eng.retrieval_loop(cst, los, par_gas, par_w, -x0, scene.grid, scene.bands_nm, scene.widths_nm, oe, pos, n_par, max_it=3,
                           fov=scene._fov_fac)
    r0 = eng.retrieval_loop(cst, los, par_gas, par_w, x0, scene.grid, scene.bands_nm, scene.widths_nm, oe, pos, n_par, max_it=0,
                            fov=scene._fov_fac)
    assert len(r0[1]) == 0 and r0[2].shape == (1, n_par) and r0[3] == "" and r0[4] is None
