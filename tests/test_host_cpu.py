"""Host-side logic and the C-ABI surface, no GPU needed (no kernel is launched)."""
import ctypes as C
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import relerr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from spectrobot_amd import _lib
    return _lib


def test_library_exports_every_declared_symbol(L):
    hdr = open(os.path.join(ROOT, "include", "spectrobot_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(sr_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 15
    raw = C.CDLL(L.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), "library does not export %s" % name
    assert declared == set(L.SYMBOLS), (declared ^ set(L.SYMBOLS))
    assert L.lib.sr_abi_version() == 1
    assert L.lib.sr_strerror(0) == b"ok" and L.lib.sr_strerror(-1) == b"bad argument"


def test_tips_tables_and_partition_sum(L, golden):
    """sr_bd_tips_2003 / sr_calc_partition_sum (host code) against the reference's Fortran tables
    and its Python CalcPartitionSum."""
    g = golden("tips2003")
    dp = L.dp
    for key, gi, tab in zip(g["keys"], g["gi"], g["q_tab"]):
        out_gi = C.c_double(0)
        t = np.zeros(119)
        q = np.zeros(119)
        assert L.lib.sr_bd_tips_2003(int(key[0]), int(key[1]), C.byref(out_gi), t.ctypes.data_as(dp),
                                     q.ctypes.data_as(dp)) == 0
        assert out_gi.value == gi and np.array_equal(t, g["t_grid"]) and np.array_equal(q, tab)
    assert L.lib.sr_bd_tips_2003(99, 1, None, None, None) == L.SR_ERR_TABLE
    for mol, iso, T, qref in g["samples"]:
        tt = np.array([T])
        q = np.zeros(1)
        assert L.lib.sr_calc_partition_sum(int(mol), int(iso), tt.ctypes.data_as(dp), 1, q.ctypes.data_as(dp)) == 0
        assert abs(q[0] - qref) <= 1e-13 * abs(qref)


def test_argument_checks_return_before_any_launch(L):
    dp, ip = L.dp, L.ip
    x = np.linspace(0.0, 1.0, 13010)
    y = np.zeros_like(x)
    xp, yp = x.ctypes.data_as(dp), y.ctypes.data_as(dp)
    assert L.lib.sr_humliv_bb(xp, 13010, 5, 4, 0.5, 1e-3, 1e-3, yp) == L.SR_ERR_ARG      # i1 > i2 (Fortran stop)
    assert L.lib.sr_humliv_bb(xp, 13010, 1, 13010, 0.5, 1e-3, 0.0, yp) == L.SR_ERR_ARG   # dw <= 0 (Fortran stop)
    assert L.lib.sr_curgod(5, xp, None, None, xp, None, 0, yp) == L.SR_ERR_ARG
    assert L.lib.sr_sum_all_lines(yp, 0, None, None, None, 0, 1) == L.SR_ERR_ARG
    init = np.array([0], np.int32)
    fin = np.array([3], np.int32)
    assert L.lib.sr_sum_all_lines(yp, 100, xp, init.ctypes.data_as(ip), fin.ctypes.data_as(ip), 1, 10) == L.SR_ERR_ARG
    # lineset: bad grid / too many grid points are refused before anything is uploaded
    ld = L.LinesDesc()
    ld.n_lines = 0
    iso = L.IsoMolecDesc(6, 1, 16.0, 0, None)
    h = C.c_void_p()
    assert L.lib.sr_lineset_create(C.byref(ld), C.byref(iso), C.byref(L.GridDesc(3000.0, -1.0, 100)),
                                   C.byref(h), None) == L.SR_ERR_ARG
    assert L.lib.sr_lineset_create(C.byref(ld), C.byref(iso), C.byref(L.GridDesc(3000.0, 5e-4, 2000001)),
                                   C.byref(h), None) == L.SR_ERR_LIMIT
    # a resident LOS: bad descriptions are refused before anything is staged
    hl = C.c_void_p()
    d = L.LosDesc()
    assert L.lib.sr_los_create(C.byref(d), 0, C.byref(hl)) == L.SR_ERR_ARG            # no layers
    assert L.lib.sr_los_create(C.byref(d), 4, C.byref(hl)) == L.SR_ERR_ARG            # empty description
    so, sl, po = np.array([0, 1], np.int32), np.array([7], np.int32), np.array([0, 2], np.int32)
    xx = np.array([0.0, 1.0])
    d.n_rays, d.n_gas = 1, 1
    d.seg_off, d.seg_layer, d.pt_off = (a.ctypes.data_as(ip) for a in (so, sl, po))
    d.x = d.nd = d.vmr = xx.ctypes.data_as(dp)
    assert L.lib.sr_los_create(C.byref(d), 4, C.byref(hl)) == L.SR_ERR_ARG and not hl.value   # seg_layer 7 of 4 layers
    assert L.lib.sr_los_destroy(None) == L.SR_OK
    assert L.lib.sr_limb_rays_los_dev(None, None, 4, 10, None, 0, None, None) == L.SR_ERR_ARG
    assert L.lib.sr_limb_step_dev(None, None, 0, 10, None, None, None, None, None) == L.SR_ERR_ARG
    assert L.lib.sr_retrieval_forward_dev(None, None, 4, 10, None, 0, None, 3000.0, 5e-4, None, None, 3, 5.0, 0, None, None,
                                          None, None) == L.SR_ERR_ARG                      # no batch, no parameters
    # the loop and the timing read-out refuse missing handles / descriptions before anything touches a device
    assert L.lib.sr_retrieval_loop_dev(None, None, 4, 10, None, 0, None, 3000.0, 5e-4, None, None, 3, 5.0, 0, None, None, None, None,
                                       None, None, None, None, None, None, None, None) == L.SR_ERR_ARG
    assert L.lib.sr_los_last_kernel_ms(None, None) == L.SR_ERR_ARG
    assert L.lib.sr_set_band_fusion(1) == L.SR_OK
    with pytest.raises(L.SpectRobotHipError):
        L.check(L.SR_ERR_ARG, "x")


def test_shard_bounds_cover_the_grid():
    from spectrobot_amd.distributed import shard_bounds
    for n in (100000, 99999, 8, 13):
        for w in (1, 2, 3, 4, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_all_gather_spectrum_gloo_world2(tmp_path):
    """The N > 1 reassembly path with the gloo backend, two CPU ranks."""
    script = tmp_path / "w.py"
    script.write_text(
        "import os, sys, torch\n"
        "sys.path.insert(0, %r)\n"
        "from spectrobot_amd import distributed as sd\n"
        "rank, local, world = sd.init_from_env(backend='gloo')\n"
        "for n, rays in ((1001, 3), (1000, 1), (1000, 3)):   # ragged shards; equal shards: in place / one copy\n"
        "    full = torch.arange(rays * n, dtype=torch.float64).reshape(rays, n)\n"
        "    lo, hi = sd.shard_bounds(n, world, rank)\n"
        "    out = sd.all_gather_spectrum(full[:, lo:hi].contiguous(), n, world, rank)\n"
        "    assert torch.equal(out, full), (rank, n, rays)\n"
        "    buf = torch.zeros((rays, n), dtype=torch.float64)\n"
        "    assert sd.all_gather_spectrum(full[:, lo:hi].contiguous(), n, world, rank, out=buf) is buf\n"
        "    assert torch.equal(buf, full), (rank, n, rays)\n"
        "# async_op: several steps in flight into the same buffer, valid after wait_gathers()\n"
        "n = 1000; lo, hi = sd.shard_bounds(n, world, rank); buf = torch.zeros((1, n), dtype=torch.float64)\n"
        "for step in range(7):\n"
        "    full = torch.arange(n, dtype=torch.float64).reshape(1, n) + 1000.0 * step\n"
        "    assert sd.all_gather_spectrum(full[:, lo:hi].contiguous(), n, world, rank, out=buf, async_op=True) is buf\n"
        "sd.wait_gathers()\n"
        "assert torch.equal(buf, full) and not sd._pending, rank\n"
        "torch.distributed.barrier()\n"
        "print('rank', rank, 'ok')\n" % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)


def test_bench_launches_its_own_ranks(tmp_path, capfd):
    """`python bench.py --gpus N` without WORLD_SIZE (the driver's form) starts its N ranks itself: bench.launch_ranks
    with a stand-in child (gloo, two CPU ranks; the bench's own child needs a GPU: tests/test_gpu_configs.py) -- the
    ranks rendezvous on the loopback port the launcher chose, only rank 0's line reaches stdout, the exit code is 0;
    a rank that fails takes the job down with its code and the launcher ends the rank left waiting."""
    import time
    sys.path.insert(0, ROOT)
    import bench
    ok = tmp_path / "ok.py"
    ok.write_text(
        "import os, sys, json, torch\n"
        "sys.path.insert(0, %r)\n"
        "from spectrobot_amd import distributed as sd\n"
        "assert os.environ['SR_BENCH_CHILD'] == '1' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
        "rank, local, world = sd.init_from_env(backend='gloo')\n"
        "t = torch.tensor([float(rank + 1)], dtype=torch.float64)\n"
        "torch.distributed.all_reduce(t)\n"
        "if rank == 0:\n"
        "    print(json.dumps({'dist': sd.dist_info(), 'sum': float(t.item()), 'argv': sys.argv[1:]}))\n"
        "torch.distributed.barrier()\n" % ROOT)
    capfd.readouterr()
    assert bench.launch_ranks(2, argv=[str(ok), "--steps", "5"], build=False) == 0
    lines = [ln for ln in capfd.readouterr().out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec["dist"] == {"backend": "gloo", "world_size": 2, "rank": 0} and rec["sum"] == 3.0
    assert rec["argv"] == ["--steps", "5"]
    bad = tmp_path / "bad.py"
    bad.write_text("import os, sys, time\n"
                   "if os.environ['RANK'] == '1':\n"
                   "    sys.exit(3)\n"
                   "time.sleep(600)\n")
    t0 = time.time()
    assert bench.launch_ranks(2, argv=[str(bad)], build=False) == 3
    assert time.time() - t0 < 30.0


def test_group_observations_golden():
    """smm.make_group_observations / make_radtran_spline (the group_observations route of the reference's drivers,
    spect_main_module.py:3290-3338, 3377-3396) against the reference's own functions run under Python 3
    (tests/golden/group_obs.npz, written by make_golden.py --group-obs): the ladder of simulated tangent altitudes --
    exact, incl. the clamp of alt_first_los and the in-place sort of the pixels -- and the spline's values."""
    sys.path.insert(0, ROOT)
    from spectrobot_amd import spect_main_module as smm, retrieval
    g = np.load(os.path.join(ROOT, "tests", "golden", "group_obs.npz"))
    for i in range(3):
        first = None if np.isnan(g["go%d_first" % i]) else float(g["go%d_first" % i])
        pix = [retrieval.LimbPixel(a, fov_half=float(g["go%d_half" % i])) for a in g["go%d_pix_alts" % i]]
        alts, mean = smm.make_group_observations(pix, alt_step=float(g["go%d_step" % i]), alt_first_los=first)
        assert np.array_equal(alts, g["go%d_alts" % i]) and np.array_equal(alts, g["go%d_los_alts" % i])
        assert [p.limb_tg_alt for p in pix] == list(g["go%d_sorted" % i]) and mean == {}
        for k, p in enumerate(pix):     # with the geometry attributes: the means the reference builds its LOS from
            p.limb_tg_lat, p.limb_tg_lon, p.limb_tg_sza = -40.0 + 3.0 * k, 120.0 + k, 55.0 + 2.0 * k
    f = smm.make_radtran_spline(g["spl_alts"], g["spl_spectra"])
    v = np.array([f(x) for x in g["spl_x"]])
    assert np.max(np.abs(v - g["spl_values"])) <= 4e-16 * np.max(np.abs(g["spl_values"]))
    # spectrum objects in, a spectrum object out (the reference's call shape)
    rads = [retrieval.Spectrum(sp, g["spl_grid"]) for sp in g["spl_spectra"]]
    out = smm.make_radtran_spline(g["spl_alts"], rads)(337.5)
    assert np.max(np.abs(out.spectrum - g["spl_values"][3])) <= 4e-16 * np.max(np.abs(g["spl_values"]))
    assert np.array_equal(out.spectral_grid.grid, g["spl_grid"])


def test_eight_ranks_through_the_launcher(tmp_path, capfd):
    """eight ranks had never run, even on gloo.  bench.launch_ranks with a CPU stand-in of the bench's
    N > 1 path: 8 ranks, an UNEQUAL split (n_grid = 100001: shard_bounds gives the first rank one point more; the
    gather pads), 6 steps of gather + the per-rank record exchange the bench line does (all_gather of [elapsed, enqueue,
    wait]), shard_costs' line counts; then the same job with a rank that dies in step 3: the launcher returns its code
    and ends the seven ranks left waiting in the collective."""
    import time
    sys.path.insert(0, ROOT)
    import bench
    body = (
        "import os, sys, json, time, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from spectrobot_amd import distributed as sd\n"
        "rank, local, world = sd.init_from_env(backend='gloo')\n"
        "assert world == 8\n"
        "n = 100001\n"
        "bounds = [sd.shard_bounds(n, world, r) for r in range(world)]\n"
        "lo, hi = bounds[rank]\n"
        "full = torch.zeros((1, n), dtype=torch.float64)\n"
        "t0 = time.perf_counter()\n"
        "for step in range(6):\n"
        "    if DIE and rank == 5 and step == 3:\n"
        "        os._exit(7)\n"
        "    ref = torch.arange(n, dtype=torch.float64).reshape(1, n) + 1e6 * step\n"
        "    out = sd.all_gather_spectrum(ref[:, lo:hi].contiguous(), n, world, rank, out=full, bounds=bounds, async_op=True)\n"
        "    sd.wait_gathers()\n"
        "    assert torch.equal(out, ref), (rank, step)\n"
        "mine = torch.tensor([time.perf_counter() - t0, 0.0, 0.0], dtype=torch.float64)\n"
        "every = [torch.zeros_like(mine) for _ in range(world)]\n"
        "torch.distributed.all_gather(every, mine)\n"
        "freq = 2975.0 + 50.0 * np.random.default_rng(1).random(5000)\n"
        "costs, lines = sd.shard_costs(freq, 2975.0 + 5e-4 * np.arange(n), bounds)\n"
        "if rank == 0:\n"
        "    print(json.dumps({'world': world, 'points': [b - a for a, b in bounds], 'n_times': len(every), 'lines': lines,\n"
        "                      'balance': max(costs) / (sum(costs) / len(costs))}))\n"
        "torch.distributed.barrier()\n" % ROOT)
    ok = tmp_path / "ok8.py"
    ok.write_text("DIE = False\n" + body)
    capfd.readouterr()
    assert bench.launch_ranks(8, argv=[str(ok)], build=False) == 0
    rec = json.loads([ln for ln in capfd.readouterr().out.splitlines() if ln.startswith("{")][-1])
    assert rec["world"] == 8 and rec["n_times"] == 8 and sum(rec["points"]) == 100001 and max(rec["points"]) - min(rec["points"]) == 1
    assert len(rec["lines"]) == 8 and sum(rec["lines"]) > 5000 and rec["balance"] >= 1.0     # (window halos are counted twice)
    bad = tmp_path / "bad8.py"
    bad.write_text("DIE = True\n" + body)
    t0 = time.time()
    assert bench.launch_ranks(8, argv=[str(bad)], build=False) == 7
    assert time.time() - t0 < 60.0


def test_hardware_queues_default_is_set_before_hip_and_respects_the_caller():
    """The coefficient op runs on six HIP streams; the package asks ROCm for eight hardware queues (GPU_MAX_HW_QUEUES) at
    import -- the runtime reads it at its first HIP call -- unless the caller has set the variable (spectrobot_amd/__init__.py)."""
    code = "import os, sys; sys.path.insert(0, %r); import spectrobot_amd; print(os.environ['GPU_MAX_HW_QUEUES'])" % ROOT
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0 and out.stdout.decode().strip() == "8", out.stderr.decode()[-500:]
    out = subprocess.run([sys.executable, "-c", code], env=dict(env, GPU_MAX_HW_QUEUES="4"), stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0 and out.stdout.decode().strip() == "4"


def test_c_abi_callers_learn_about_hardware_queues():
    """a C-ABI caller that never imports the Python package gets the runtime's 4 hardware queues and a
    3 % slower schedule unless told -- sr_recommended_hw_queues (no GPU needed) reports what to export and what the
    process runs with."""
    import ctypes as C
    code = ("import ctypes as C, sys; L = C.CDLL(%r); r, c = C.c_int(0), C.c_int(0); "
            "assert L.sr_recommended_hw_queues(C.byref(r), C.byref(c)) == 0; assert L.sr_recommended_hw_queues(None, None) == 0; "
            "print(r.value, c.value)" % os.path.join(ROOT, "spectrobot_amd", "lib", "libspectrobot_hip.so"))
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    for given, want in ((None, "8 4"), ("8", "8 8"), ("2", "8 2")):
        e = dict(env) if given is None else dict(env, GPU_MAX_HW_QUEUES=given)
        out = subprocess.run([sys.executable, "-c", code], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert out.returncode == 0 and out.stdout.decode().strip() == want, (given, out.stdout, out.stderr.decode()[-500:])


def test_async_gather_branch_bookkeeping(monkeypatch):
    """The asynchronous branch of all_gather_spectrum (taken for the RCCL backend only) with the collective
    replaced by a recorder: every Work handle is waited on exactly once -- on eviction (at most 4 in flight) or
    by wait_gathers() -- and its input shard stays referenced until then."""
    import gc
    import weakref
    import torch
    from spectrobot_amd import distributed as sd

    class Work(object):
        def __init__(self, log, i):
            self.log, self.i = log, i

        def wait(self):
            self.log.append(self.i)

    waited, issued, refs = [], [], []

    def fake_gather(out, inp, async_op=False):
        assert async_op
        out.copy_(inp.expand_as(out))
        issued.append(len(issued))
        return Work(waited, issued[-1])

    monkeypatch.setattr(sd.dist, "get_backend", lambda: "nccl")
    monkeypatch.setattr(sd.dist, "all_gather_into_tensor", fake_gather)
    monkeypatch.setattr(sd, "_pending", [])
    before = dict(sd.stats)
    buf = torch.zeros((1, 64), dtype=torch.float64)
    for step in range(7):
        shard = torch.full((1, 32), float(step), dtype=torch.float64)
        refs.append(weakref.ref(shard))
        assert sd.all_gather_spectrum(shard, 64, 2, 0, out=buf, async_op=True) is buf
        del shard
    gc.collect()
    assert sd.stats["async_gathers"] - before["async_gathers"] == 7                   # the async branch ran
    assert sd.stats["blocking_gathers"] == before["blocking_gathers"]
    assert waited == [0, 1, 2] and len(sd._pending) == 4                                  # evicted handles were waited on
    assert all(r() is None for r in refs[:3]) and all(r() is not None for r in refs[3:])  # in-flight inputs stay alive
    sd.wait_gathers()
    assert waited == list(range(7)) and not sd._pending
    gc.collect()
    assert all(r() is None for r in refs)
    # a one-rank group goes through the collective only when asked to (the hardware test of the RCCL branch)
    one = torch.ones((1, 8), dtype=torch.float64)
    assert sd.all_gather_spectrum(one, 8, 1, 0) is one and len(issued) == 7
    out = sd.all_gather_spectrum(one, 8, 1, 0, async_op=True, force_collective=True)
    sd.wait_gathers()
    assert len(issued) == 8 and torch.equal(out, one)
    with pytest.raises(ValueError):
        sd.all_gather_spectrum(one, 8, 2, 0, bounds=[(0, -4), (-4, 8)])


def test_shard_bounds_balanced_short_grid():
    """Grids shorter than world_size x align: boundaries get finer instead of negative / overlapping."""
    from spectrobot_amd import distributed as sd, synthetic as syn
    grid = syn.make_grid(2975.0, 5e-4, 300)
    freq = np.sort(np.random.default_rng(0).uniform(grid[0], grid[-1], 50))
    for w in (2, 8):
        b = sd.shard_bounds_balanced(freq, grid, w)
        assert b[0][0] == 0 and b[-1][1] == 300
        assert all(lo < hi for lo, hi in b) and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
    with pytest.raises(ValueError):
        sd.shard_bounds_balanced(freq, grid[:4], 8)


def test_sharded_band_integrals_add_up(oracle):
    """configs[4] on N GPUs (retrieval.simulate): every rank integrates the instrument bands over its spectral shard
    plus the next rank's first point; the partial integrals add up to the unsharded ones (each trapezoid belongs to
    the shard that owns its left point).  Checked with the oracle's hires_to_lowres on the CPU, shard bounds from
    shard_with_halo; the all-reduce itself under gloo with two ranks."""
    from spectrobot_amd import distributed as sd, synthetic as syn
    from spectrobot_amd.retrieval import shard_with_halo
    n = 6000
    grid = syn.make_grid(3290.0, 5e-4, n)
    rng = np.random.default_rng(2)
    spec_hi = rng.uniform(0.5, 1.5, n) * 1e-3
    lam = np.linspace(1e7 / grid[-1] + 0.3, 1e7 / grid[0] - 0.3, 9)
    wid = np.full(9, 0.25)
    full = oracle.hires_to_lowres(grid, spec_hi, lam, wid)
    for world in (2, 3, 8):
        parts = np.zeros_like(full)
        for r in range(world):
            lo, hi = shard_with_halo(n, *sd.shard_bounds(n, world, r))
            assert hi - lo >= 2
            parts += oracle.hires_to_lowres(grid[lo:hi], spec_hi[lo:hi], lam, wid)
        assert np.max(np.abs(parts - full)) < 1e-13 * np.max(np.abs(full)), world


def test_all_reduce_sum_gloo_world2(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(
        "import os, sys, torch\n"
        "sys.path.insert(0, %r)\n"
        "from spectrobot_amd import distributed as sd\n"
        "rank, local, world = sd.init_from_env(backend='gloo')\n"
        "t = torch.arange(12, dtype=torch.float64).reshape(3, 4) * (rank + 1)\n"
        "assert sd.all_reduce_sum(t) is t and torch.equal(t, torch.arange(12, dtype=torch.float64).reshape(3, 4) * 3)\n"
        "assert sd.dist_info() == {'backend': 'gloo', 'world_size': 2, 'rank': rank}\n"
        "torch.distributed.barrier()\n"
        "print('rank', rank, 'ok')\n" % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29536", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_shard_bounds_balanced_on_skewed_line_density(tmp_path):
    """Equal-work shards: a band head holding 70 % of the lines in 15 % of the grid.  The
    work model's per-shard cost spread drops from several-fold (equal width) to a few per cent; the
    gather with unequal shards reassembles the spectrum (gloo, two ranks)."""
    from spectrobot_amd import distributed as sd, synthetic as syn
    rng = np.random.default_rng(3)
    n = 100000
    grid = syn.make_grid(2975.0, 5e-4, n)
    freq = np.sort(np.concatenate([rng.uniform(grid[20000], grid[35000], 70000), rng.uniform(grid[0], grid[-1], 30000)]))

    def costs(bounds):
        ic = np.clip(np.rint((freq - grid[0]) / (grid[1] - grid[0])).astype(int), 0, n - 1)
        out = []
        for lo, hi in bounds:
            cover = np.clip(np.minimum(ic + 6504, hi - 1) - np.maximum(ic - 6505, lo) + 1, 0, None).sum()   # (line, point) pairs
            out.append(cover + 1.5 * 13010.0 * np.count_nonzero((ic >= lo) & (ic < hi)))
        return np.array(out)

    for w in (2, 4, 8):
        bal = sd.shard_bounds_balanced(freq, grid, w)
        assert bal[0][0] == 0 and bal[-1][1] == n and all(bal[i][1] == bal[i + 1][0] for i in range(w - 1))
        assert all((hi - lo) > 0 and lo % 64 == 0 for lo, hi in bal)
        c_eq = costs([sd.shard_bounds(n, w, r) for r in range(w)])
        c_bal = costs(bal)
        assert c_bal.max() / c_bal.mean() < 1.05, (w, c_bal / c_bal.mean())
        if w == 8:
            assert c_eq.max() / c_eq.mean() > 2.0
    script = tmp_path / "w.py"
    script.write_text(
        "import os, sys, torch\n"
        "sys.path.insert(0, %r)\n"
        "from spectrobot_amd import distributed as sd\n"
        "rank, local, world = sd.init_from_env(backend='gloo')\n"
        "bounds = [(0, 640), (640, 1000)]\n"
        "full = torch.arange(3 * 1000, dtype=torch.float64).reshape(3, 1000)\n"
        "lo, hi = bounds[rank]\n"
        "out = sd.all_gather_spectrum(full[:, lo:hi].contiguous(), 1000, world, rank, bounds=bounds)\n"
        "assert torch.equal(out, full), rank\n"
        "torch.distributed.barrier()\n"
        "print('rank', rank, 'ok')\n" % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29534", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_lines_to_soa_matches_linktomolec(golden):
    """Level resolution incl. the reference's quirks (unidentified and same-level lines)."""
    from spectrobot_amd import spect_classes as spcl, spect_base_module as sbm
    g = golden("e2e_ch4_levels")
    iso = sbm.IsoMolec(6, 1, float(g["mm"]))
    for i, e in enumerate(g["e_lev"]):
        iso.add_level("L%02d" % i, e)
    lines = []
    for i in range(len(g["line_freq"])):
        up = "L%02d" % g["line_lev_up"][i] if g["line_lev_up"][i] >= 0 else "??"
        lo = "L%02d" % g["line_lev_lo"][i] if g["line_lev_lo"][i] >= 0 else "??"
        lines.append(spcl.SpectLine([6, 1, g["line_freq"][i], 0.0, g["line_a_coeff"][i], g["line_air_broad"][i], 0.0,
                                     g["line_e_lower"][i], g["line_t_dep_broad"][i], 0.0, up, lo, "", "", "",
                                     g["line_g_up"][i], g["line_g_lo"][i]], nomi=spcl.cose_hit))
    soa = spcl.lines_to_soa(lines, iso)
    linked = np.array([l.LinkToMolec(iso) for l in lines])
    kept = (soa["lev_up"] >= 0) & (soa["lev_lo"] >= 0) & (soa["lev_up"] != soa["lev_lo"])
    assert np.array_equal(linked, kept)
    assert (~linked).sum() >= 4          # the fixture pins at least 4 dropped lines
    assert np.array_equal(soa["freq"], g["line_freq"])


def test_scalar_mirror_against_reference_values(golden):
    from spectrobot_amd import spect_classes as spcl
    g = golden("spcl_scalars")
    for k in ("h_cgs", "c_cgs", "k_cgs", "c2", "hpa_to_atm", "T_ref"):
        assert getattr(spcl, k) == float(g["const_" + k])
    for i in range(len(g["T"])):
        lw = spcl.Lorenz_width(g["T"][i], spcl.convert_to_atm(g["P"][i]), g["n_air"][i], g["gam"][i])
        assert abs(lw - g["lw"][i]) <= 1e-15 * g["lw"][i]
        assert abs(spcl.Doppler_width(g["T"][i], g["MM"][i], g["nu"][i]) - g["dw"][i]) <= 1e-15 * g["dw"][i]
        l = spcl.SpectLine([6, 1, g["nu"][i], 0.0, g["A"][i], g["gam"][i], 0.0, g["El"][i], g["n_air"][i], 0.0,
                            "a", "b", "", "", "", g["gu"][i], g["gl"][i]], nomi=spcl.cose_hit)
        G = [spcl.Einstein_A_to_Gcoeff_spem(l, g["T"][i], g["Evu"][i]),
             spcl.Einstein_A_to_Gcoeff_indem(l, g["T"][i], g["Evu"][i]),
             spcl.Einstein_A_to_Gcoeff_abs(l, g["T"][i], g["Evl"][i])]
        assert np.allclose(G, g["G"][i], rtol=1e-14, atol=0)
        assert abs(spcl.Calc_BB_single(g["nu"][i], g["T"][i]) - g["bb"][i]) <= 1e-14 * g["bb"][i]


def test_grid_params_and_synthetic():
    from spectrobot_amd import engine, synthetic as syn
    grid = syn.make_grid(2975.0, 5e-4, 100000)
    w0, step, n = engine.grid_params(grid)
    assert (w0, n) == (2975.0, 100000) and step == grid[1] - grid[0]
    with pytest.raises(ValueError):
        engine.grid_params(np.array([1.0, 2.0, 4.0]))
    a = syn.make_lines(100, grid, config_id=2)
    b = syn.make_lines(100, grid, config_id=2)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert np.all(np.diff(a["freq"]) >= 0)
    atm = syn.make_atmosphere(80, 12)
    assert atm["temps"].shape == (80,) and atm["tvib"].shape == (12, 80)
    sl, ln = syn.limb_path(atm["z"], 100.001)
    assert len(sl) == 160 and np.all(ln > 0) and sl[0] == 79 and sl[79] == 0 and sl[80] == 0


def test_no_product_import_of_the_oracle():
    """The product must not import, link or execute anything under oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "spectrobot_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".inc")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                for bad in ("import oracle", "from oracle", "sr_oracle", "liboracle", "oracle/_ref", "ref_fortran"):
                    assert bad not in txt, (f, bad)


def test_read_line_database_hitran(golden, tmp_path):
    """N3: the HITRAN 160-column reader against the reference's read_line_database
    (spect_classes.py:1532-1601) on a file written by the reference's Print_hitran."""
    from spectrobot_amd import spect_classes as spcl
    g = golden("hitran_sample")
    path = os.path.join(ROOT, "tests", "golden", "hitran_sample.par")
    lines = spcl.read_line_database(path)
    assert len(lines) == len(g["Freq"])
    for k in ("Mol", "Iso", "Freq", "Strength", "A_coeff", "Air_broad", "Self_broad", "E_lower", "T_dep_broad",
              "P_shift", "g_up", "g_lo"):
        assert np.array_equal(np.array([getattr(l, k) for l in lines], dtype=float), g[k]), k
    assert [l.Up_lev_str for l in lines] == list(g["Up_lev_str"])
    assert [l.Q_num_lo for l in lines] == list(g["Q_num_lo"])
    sel = spcl.read_line_database(path, mol=6, iso=1, freq_range=[2950.0, 3050.0])
    assert np.array_equal([l.Freq for l in sel], g["sel_freq"])
    frac = spcl.read_line_database(path, fraction_to_keep=0.5)
    assert np.array_equal([l.Freq for l in frac], g["frac_freq"])
    with pytest.raises(ValueError):
        spcl.read_line_database(path, db_format="xyz")
    # n_skip: a fixed number of header lines, or -1 = up to the header's '#' line (sbm.trova_spip, spcl:1558-1559)
    body = open(path).read()
    hdr = tmp_path / "with_header.par"
    hdr.write_text("line list printed for the test\ntwo lines of free text\n#\n" + body)
    for ns in (3, -1):
        got = spcl.read_line_database(str(hdr), n_skip=ns)
        assert np.array_equal([l.Freq for l in got], g["Freq"]), ns
    with pytest.raises(ValueError):
        spcl.read_line_database(path, n_skip=-1)      # no '#' line: an error, not an endless loop
    soa = spcl.lines_to_soa(sel)
    assert np.array_equal(soa["freq"], g["sel_freq"]) and soa["lev_up"].min() == -1


def test_inversion_algebra_against_reference(golden):
    """N4 algebra: one Levenberg-Marquardt step against the reference's inversion_algebra / chicalc."""
    from spectrobot_amd import spect_main_module as smm
    g = golden("inversion_algebra")

    class BS(object):
        def build_jacobian(self, masks=None): return g["K"]
        def param_vector(self): return g["xi"]
        def VCM_apriori(self): return g["S_ap"]
        def apriori_vector(self): return g["x_ap"]
        def update_params(self, dx): self.dx = dx
        def store_avk(self, a): self.avk = a
        def store_VCM(self, s): self.vcm = s

    class Sp(object):
        def __init__(self, v): self.spectrum = v
    h = len(g["obs"]) // 2
    o, s, nz = ([Sp(g[k][:h]), Sp(g[k][h:])] for k in ("obs", "sim", "noise"))
    for lam in ("0.1", "1"):
        bs = BS()
        smm.inversion_algebra(o, s, nz, bs, lambda_LM=float(lam))
        assert np.allclose(bs.dx, g["dx_" + lam], rtol=1e-9, atol=0)
        assert np.allclose(bs.avk, g["avk_" + lam], rtol=1e-9, atol=1e-14)
        assert np.allclose(bs.vcm, g["vcm_" + lam], rtol=1e-9, atol=0)
    assert abs(smm.chicalc(o, s, nz, None, len(g["xi"])) - float(g["chi"])) < 1e-12 * float(g["chi"])


def test_retrieval_parameter_space_against_reference(golden):
    """BayesSet / RetParam / LinearProfile_1D_new / alt_triangle / lat_box (smm:169-665) and
    FOV_integr_1D (smm:3342-3374) against the reference's own classes run under Python 3
    (tests/golden/make_golden.py --retrieval)."""
    from spectrobot_amd import spect_main_module as smm
    g = golden("retrieval_classes")
    alts = g["alts"]
    pa = smm.LinearProfile_1D_new("CH4", alts, list(g["nodes_a"]), g["ap_a"], g["er_a"], first_guess_prof=g["fg_a"])
    pb = smm.LinearProfile_1D_new("HCN", alts, list(g["nodes_b"]), g["ap_b"], g["er_b"])
    assert np.array_equal(pa.mask_matrix(), g["masks_a"]) and np.array_equal(pb.mask_matrix(), g["masks_b"])
    assert np.array_equal(smm.alt_triangle(alts, 420.0, step=75.0).mask, g["tri_step"])
    assert np.allclose(pa.mask_matrix().sum(0), 1.0)  # the triangles are a partition of unity
    inv = np.array([[pa.check_involved(k, {"alt": (lo, lo + 50.0)}) for lo in (100.0, 320.0, 650.0, 850.0)]
                    for k in g["nodes_a"]])
    assert np.array_equal(inv, g["involved_a"])
    lat = list(g["lat_limits"])
    boxes = np.array([smm.lat_box(lat, la).mask for la in (-75.0, -30.0, 10.0, 59.9, 75.0)])
    assert np.array_equal(boxes, g["lat_boxes"]) and np.allclose(smm.centre_boxes(lat), g["lat_centres"])

    class Sp(object):
        def __init__(self, v):
            self.spectrum = np.array(v, dtype=float)

    bs = smm.BayesSet(tag="golden")
    bs.add_set(pa)
    bs.add_set(pb)
    assert bs.n_tot == 8 and np.array_equal(bs.param_vector(), g["x0"])
    for ip, par in enumerate(bs.params()):
        for num in range(g["ders"].shape[1]):
            par.store_deriv(Sp(g["ders"][ip, num]), num)
    masks = [m for m in g["pix_masks"]]
    assert np.array_equal(bs.build_jacobian(), g["jac"])
    assert np.array_equal(bs.build_jacobian(masks=masks), g["jac_masked"])
    assert np.array_equal(bs.VCM_apriori(), g["S_ap"]) and np.array_equal(bs.apriori_vector(), g["x_ap"])
    bs.update_params(g["dx_pos"])  # two steps are halved until the parameter stays positive
    assert np.array_equal(bs.param_vector(), g["x_after_pos"]) and np.array_equal(bs.old_params[0], g["old_params_0"])
    assert (bs.param_vector() > 0).all()
    for par in bs.params():
        par.set_used()
    sim, noi, obs = ([Sp(v) for v in g[k]] for k in ("sim", "noi", "obs"))
    assert abs(smm.chicalc(obs, sim, noi, masks, bs.n_used_par()) - float(g["chi"])) < 1e-12 * float(g["chi"])
    smm.inversion_algebra(obs, sim, noi, bs, lambda_LM=0.1, masks=masks)
    bs.update_parerror()
    assert np.allclose(bs.param_vector(), g["x_after_lm"], rtol=1e-9, atol=0)
    assert np.allclose(bs.VCM, g["vcm"], rtol=1e-8, atol=0) and np.allclose(bs.av_kernel, g["avk"], rtol=1e-7, atol=1e-12)
    assert np.allclose([p.ret_error for p in bs.params()], g["ret_error"], rtol=1e-8)
    assert smm.retrieval_converged(1.0, None) == "" and smm.retrieval_converged(1.005, 1.0) == "converged"
    assert smm.retrieval_converged(1.3, 1.0) == "raised" and smm.retrieval_converged(0.7, 1.0) == ""

    class Rad(object):
        def __init__(self, v):
            self.spectrum = np.array(v, dtype=float)

    for rot, want in zip(g["fov_rot"], g["fov_out"]):
        rads = [Rad(v) for v in g["fov_spe"]]
        got = smm.FOV_integr_1D(rads, pixel_rot=float(rot)).spectrum
        assert np.allclose(got, want, rtol=1e-12, atol=0), (rot, np.abs(got / want - 1).max())
        # the reference's quadrature error at its default tolerance: 2.5e-4 for a rotated pixel
        exact = smm.FOV_integr_1D(rads, pixel_rot=float(rot), closed_form=True).spectrum
        assert np.abs(exact / want - 1).max() < (1e-12 if rot == 0 else 1.5e-3)


def test_molparam_and_key_input_readers(tmp_path, monkeypatch):
    """N3: HITRAN molparam.txt -> MM / isotopic ratio, and the drivers' [key] input files.  Both readers
    belong to the absent spect_base_module: checked against the files themselves and the values the
    reference hard-codes from them (spect_main.py:152: CH4 ratio 0.98827; SURVEY 8-d: MM 16.0313)."""
    from spectrobot_amd import spect_base_module as sbm
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    mp = os.path.join(here, "molparam_first7.txt")
    tab = sbm.read_molparam(mp)
    assert len([k for k in tab if k[0] == 1]) == 6 and len([k for k in tab if k[0] == 2]) == 11
    ch4 = sbm.find_molec_metadata(6, 1, filename=mp)
    assert ch4["mol_name"] == "CH4" and ch4["iso_name"] == "211" and ch4["iso_MM"] == 16.0313
    assert abs(ch4["iso_ratio"] - 0.98827) < 5e-6 and ch4["gj"] == 1 and ch4["Q_296"] == 590.52
    co = sbm.find_molec_metadata(5, 1, filename=mp)
    assert co["iso_MM"] == 27.994915 and co["iso_name"] == "26"
    monkeypatch.setenv("SPECTROBOT_MOLPARAM", mp)
    assert sbm.find_molec_metadata(6, 3)["iso_MM"] == 17.037475
    with pytest.raises(ValueError):
        sbm.find_molec_metadata(6, 9)
    monkeypatch.delenv("SPECTROBOT_MOLPARAM")
    with pytest.raises(ValueError):
        sbm.find_molec_metadata(6, 1)

    keys = "cart_atm cart_LUTS hitran_db n_threads test n_split wn_range tangent_alts".split()
    itype = [str, str, str, int, bool, int, [float, float], float]
    defaults = ["/a/", "/luts/", None, 4, False, None, None, None]
    inp = sbm.read_inputs(os.path.join(here, "sample_inputs.in"), keys, itype=itype, defaults=defaults,
                          n_lines=[1, 1, 1, 1, 1, 1, 1, 3])
    assert inp["cart_atm"] == "/data/titan/atm/" and inp["cart_LUTS"] == "/luts/" and inp["n_threads"] == 8
    assert inp["test"] is True and inp["n_split"] is None and inp["wn_range"] == [2850.0, 3450.0]
    assert inp["tangent_alts"] == [150.0, 300.0, 450.0]
    bad = tmp_path / "bad.in"
    bad.write_text("[n_threads]\n[test]\nTrue\n")
    with pytest.raises(ValueError):
        sbm.read_inputs(str(bad), ["n_threads"], itype=[int])


def test_containers_and_lut_planner_vs_reference(golden):
    """A11 remainder and the LUT planner against the reference's own classes (make_golden.py --containers): unit
    conversions of grid + spectral density, __getitem__, __truediv__, interp_to_grid, intensity units, Calc_BB,
    SpectralGcoeff.interpolate, calc_PT_couples_atmosphere."""
    from spectrobot_amd import spect_classes as spcl, spect_main_module as smm, spect_base_module as sbm
    g = golden("containers")
    g0, sp0 = g["grid_cm"], g["spec_cm"]
    for units in ("nm", "mum", "hz", "cm_1"):
        o = spcl.SpectralObject(sp0.copy(), spcl.SpectralGrid(g0, units="cm_1"), units="cm_1")
        grid, spec = o.convert_grid_to(units)
        assert o.spectral_grid.units == units
        assert np.array_equal(grid, g["conv_%s_grid" % units]) and np.array_equal(spec, g["conv_%s_spec" % units]), units
        if units != "cm_1":
            o.convertto_cm_1()
            assert np.array_equal(o.spectral_grid.grid, g["back_%s_grid" % units]), units
            assert np.array_equal(o.spectrum, g["back_%s_spec" % units]), units
    with pytest.raises(ValueError):
        o.convert_grid_to("furlongs")
    o = spcl.SpectralObject(sp0.copy(), spcl.SpectralGrid(g0, units="cm_1"), units="cm_1")
    sub = o[(2000.5, 2001.25)]
    assert np.array_equal(sub.spectral_grid.grid, g["getitem_grid"]) and np.array_equal(sub.spectrum, g["getitem_spec"])
    assert o[(3000.0, 3001.0)] is None and bool(g["getitem_none"][0])
    assert np.array_equal((o / 2.5).spectrum, g["div_scalar"])
    assert np.array_equal((o / spcl.SpectralObject(sp0[::-1].copy(), o.spectral_grid)).spectrum, g["div_obj"])
    ng = spcl.SpectralGrid(g["interp_grid"], units="cm_1")
    assert np.array_equal(o.interp_to_grid(ng).spectrum, g["interp_spec"])
    for u in ("Wm2", "nWcm2"):
        si = spcl.SpectralIntensity(sp0.copy(), spcl.SpectralGrid(g0, units="cm_1"), units="ergscm2")
        assert np.array_equal(si.convertto(u), g["intens_" + u]) and si.units == u
    with pytest.raises(ValueError):
        si.convertto("lumens")
    sgr = spcl.SpectralGrid(g0, units="cm_1")
    for T, want in zip(g["bb_T"], g["bb"]):
        assert relerr(spcl.Calc_BB(sgr, float(T)).spectrum, want) < 1e-15
    assert relerr(spcl.Calc_BB(sgr, 150.0, units="Wm2").spectrum, g["bb_Wm2"]) < 1e-15
    a = spcl.SpectralGcoeff("absorption", sgr, 6, 1, 16.0, "L01", spectrum=sp0.copy(), Pres=1.0, Temp=150.0)
    b = spcl.SpectralGcoeff("absorption", sgr, 6, 1, 16.0, "L01", spectrum=sp0[::-1].copy(), Pres=4.0, Temp=150.0)
    c = spcl.SpectralGcoeff("absorption", sgr, 6, 1, 16.0, "L01", spectrum=sp0[::-1].copy(), Pres=1.0, Temp=160.0)
    gp, gt = a.interpolate(b, Pres=2.2), a.interpolate(c, Temp=153.0)
    assert np.array_equal(gp.spectrum, g["gint_P"]) and np.array_equal(gt.spectrum, g["gint_T"])
    assert (gp.pres, gp.temp, gt.pres, gt.temp) == (2.2, 150.0, 1.0, 153.0) and gp.lev_string == "L01"
    with pytest.raises(ValueError):
        b.interpolate(c, Temp=155.0)          # both P and T differ
    assert a.interpolate(None, Pres=2.0) is None

    class Atm(object):
        pass
    A = Atm()
    A.pres, A.temp = g["pt_press"], g["pt_temps"]
    lines = [spcl.SpectLine([6, 1, g["line_freq"][i], 0.0, g["line_a_coeff"][i], g["line_air_broad"][i], 0.0,
                             g["line_e_lower"][i], g["line_t_dep_broad"][i], 0.0, "??", "??", "", "", "",
                             g["line_g_up"][i], g["line_g_lo"][i]], nomi=spcl.cose_hit) for i in range(len(g["line_freq"]))]
    iso = sbm.IsoMolec(6, 1, 16.0313)
    for key, kw in (("a", dict()), ("b", dict(pres_step_log=1.0, temp_step=10.0, max_pres=2.0)),
                    ("c", dict(thres=0.5, add_lowpres=False))):
        pt = np.array(smm.calc_PT_couples_atmosphere(lines, iso, A, **kw))
        assert pt.shape == g["pt_" + key].shape and np.array_equal(pt, g["pt_" + key]), key


def test_linear_profile_1d_and_2d_vs_reference(golden):
    """LinearProfile_1D (older constructor: middle nodes take the a priori of the node before them, sic) and
    LinearProfile_2D (altitude nodes x latitude boxes; masks merged as lat x alt) against the reference's classes."""
    from spectrobot_amd import spect_main_module as smm
    g = golden("retrieval_classes")

    class Grid(object):
        pass

    class Atmo(object):
        pass
    atmo = Atmo()
    atmo.grid = Grid()
    atmo.grid.grid = [g["alts"]]
    atmo.grid.coords = {"alt": g["alts"]}
    p1 = smm.LinearProfile_1D("CH4", atmo, list(g["nodes_a"]), list(g["ap_a"]), list(g["er_a"]), first_guess_prof=list(g["fg_a"]))
    assert np.array_equal(np.array([p.maskgrid.mask for p in p1.set]), g["lp1d_masks"])
    assert np.array_equal(np.array([p.apriori for p in p1.set]), g["lp1d_apriori"])
    assert np.array_equal(np.array([p.apriori_err for p in p1.set]), g["lp1d_err"])
    assert np.array_equal(np.array([p.value for p in p1.set]), g["lp1d_value"])
    assert p1.set[2].apriori == g["ap_a"][1]          # the reference's unsliced zip
    p2 = smm.LinearProfile_2D("CH4", atmo, list(g["nodes_a"]), list(g["lp2d_lat_limits"]), list(g["lp2d_aps"]),
                              list(g["lp2d_ers"]), first_guess_profs=[g["fg_a"]] * 3)
    assert p2.n_par == 15 and len(p2.set) == 15
    assert np.array_equal(np.array([p.maskgrid.mask for p in p2.set]), g["lp2d_masks"])
    assert np.array_equal(np.array([list(p.key) for p in p2.set]), g["lp2d_keys"])
    assert np.array_equal(np.array([p.apriori for p in p2.set]), g["lp2d_apriori"])
    assert np.array_equal(np.array([p.value for p in p2.set]), g["lp2d_value"])
    inv = np.array([[p2.check_involved(p.key, {"alt": (lo, lo + 50.0), "lat": la}) for p in p2.set]
                    for lo, la in ((100.0, (-80.0, -70.0)), (320.0, (-40.0, -20.0)), (650.0, (40.0, 50.0)), (850.0, (-10.0, 10.0)))])
    assert np.array_equal(inv, g["lp2d_involved"])
    prof = p2.profile()
    assert prof.shape == (3, len(g["alts"])) and np.allclose(prof[1], sum(m * v for m, v in zip(g["masks_a"], g["lp2d_aps"][1])))


def test_limb_los_geometry_cache():
    """synthetic.limb_los keeps the ray geometry between calls (a retrieval loop changes the VMR profiles only):
    cached and fresh results agree, and other levels / tangent heights are not served from the wrong entry."""
    from spectrobot_amd import synthetic as syn
    z = np.linspace(100.0, 400.0, 16)
    nd = 1e15 * np.exp(-(z - 100.0) / 60.0)
    v1, v2 = np.full((2, 16), 1e-4), np.vstack([np.linspace(1e-4, 3e-4, 16), np.full(16, 2e-6)])
    syn._LOS_GEOMETRY.clear()
    a = syn.limb_los(z, nd, v1, [120.0, 250.0])
    assert len(syn._LOS_GEOMETRY) == 1
    b = syn.limb_los(z, nd, v2, [120.0, 250.0])           # same geometry, new profiles
    assert len(syn._LOS_GEOMETRY) == 1 and b["x"] is a["x"]
    syn._LOS_GEOMETRY.clear()
    c = syn.limb_los(z, nd, v2, [120.0, 250.0])           # fresh
    for k in ("seg_off", "seg_layer", "pt_off", "x", "nd", "vmr", "alt"):
        assert np.array_equal(b[k], c[k]), k
    assert not np.array_equal(a["vmr"], b["vmr"])
    d = syn.limb_los(z, nd, v2, [120.0, 260.0])
    e = syn.limb_los(z + 1.0, nd, v2, [120.0, 250.0])
    assert len(syn._LOS_GEOMETRY) == 3
    assert not np.array_equal(d["x"], c["x"]) and not np.array_equal(e["alt"], c["alt"])


def test_jacobians_entry_refuses_initial_intensity_without_rad(L):
    """sr_limb_rays_jacobians_dev: init_mode 1 reads the initial intensity from `rad`; with rad = NULL that was a NULL
    read on the device in the one-pass kernel.  The argument check runs on the host before anything
    is staged or launched, so it is testable without a GPU: SR_ERR_ARG (-1)."""
    seg_off = np.array([0, 1], np.int32)
    seg_layer = np.zeros(1, np.int32)
    pt_off = np.array([0, 2], np.int32)
    x = np.array([0.0, 1.0e5])
    nd = np.array([1.0e12, 0.9e12])
    vmr = np.array([1.0e-3, 1.0e-3])
    par_w = np.ones((1, 2))
    par_gas = np.zeros(1, np.int32)
    d = L.LosDesc()
    d.n_rays, d.n_gas = 1, 1
    d.seg_off, d.seg_layer, d.pt_off = (a.ctypes.data_as(L.ip) for a in (seg_off, seg_layer, pt_off))
    d.x, d.nd, d.vmr = (a.ctypes.data_as(L.dp) for a in (x, nd, vmr))
    d.col_scale = None
    d.los_order, d.solo_absorption, d.init_mode, d.t_init, d.w0, d.step, d.g_lo = 0, 0, 1, 0.0, 0.0, 0.0, 0
    fake = C.c_void_p(4096)          # never dereferenced: the call must return before touching the device
    rc = L.lib.sr_limb_rays_jacobians_dev(fake, fake, None, None, 1, 64, C.byref(d), None, 0, 1,
                                          par_gas.ctypes.data_as(L.ip), par_w.ctypes.data_as(L.dp), None, None, fake, None)
    assert rc == -1 and b"init_mode 1" in L.lib.sr_last_error()


def test_calc_radtran_steps_geometry(oracle):
    """geometry.calc_radtran_steps, the stand-in for the absent sbm LineOfSight.calc_radtran_steps with the
    reference's knobs (radtran_opt: max_T_variation, max_Plog_variation, max_opt_depth: radtran_test_CO.py:184-186,
    spect_main_module.py:2760-2762).  (i) All bounds off = limb_los's crossings; (ii) with the temperature / log-pressure
    bounds every step stays inside them, the steps of a ray chain up without gaps and cover the same path;
    (iii) the Curtis-Godson columns (oracle's curgod_fort_2) of the refined path converge: the sum over a crossing's
    steps approaches a 200-sub-interval reference as the bounds tighten; (iv) the optical-depth bound halves thick
    steps until none is left; (v) 3-D state of a path: SZA and latitude at the tangent point, latitude boxes."""
    from spectrobot_amd import geometry as geo, synthetic as syn
    atm = syn.make_atmosphere(40, 0)
    z, T, P = atm["z"], atm["temps"], atm["press"]
    nd = syn.number_density(P, T)
    vm = [np.linspace(1e-4, 3e-4, 40)]
    zt = [z[3] + 4.0, z[20] + 9.0]
    L0 = geo.limb_los(z, nd, vm, zt)
    La = geo.calc_radtran_steps(z, T, P, nd, vm, zt)
    assert all(np.array_equal(La[k], L0[k]) for k in ("seg_off", "pt_off", "x", "nd", "vmr")) and np.array_equal(La["seg_alt_layer"], L0["seg_layer"])
    zz = np.append(z, z[-1] + (z[-1] - z[-2]))
    tt, lp = np.append(T, T[-1]), np.append(np.log(P), np.log(P[-1]) + (np.log(P[-1]) - np.log(P[-2])))
    totals = []
    for mt, mp in ((None, None), (2.0, 0.25), (0.5, 0.06)):
        Lb = geo.calc_radtran_steps(z, T, P, nd, vm, zt, max_T_variation=mt, max_Plog_variation=mp)
        po, so, x = Lb["pt_off"], Lb["seg_off"], Lb["x"]
        a_end, b_end = Lb["alt"][po[:-1]], Lb["alt"][po[1:] - 1]
        if mt:
            assert np.all(np.abs(np.interp(a_end, zz, tt) - np.interp(b_end, zz, tt)) <= mt + 1e-9)
            assert np.all(np.abs(np.interp(a_end, zz, lp) - np.interp(b_end, zz, lp)) <= mp + 1e-9)
            assert len(Lb["seg_layer"]) > len(L0["seg_layer"])
        for r in range(2):
            first, last = so[r], so[r + 1]
            assert np.array_equal(x[po[first + 1:last]], x[po[first + 1:last] - 1])        # no gaps between steps
            assert x[po[first]] == L0["x"][L0["pt_off"][L0["seg_off"][r]]] and x[po[last] - 1] == L0["x"][L0["pt_off"][L0["seg_off"][r + 1]] - 1]
        col = np.array([oracle.curgod(2, Lb["nd"][a:b], Lb["x"][a:b], vmr=Lb["vmr"][0][a:b]) for a, b in zip(po[:-1], po[1:])])
        totals.append(np.array([col[so[r]:so[r + 1]].sum() for r in range(2)]))
        assert np.all(Lb["step_temp"] > 0) and np.all(np.diff(Lb["seg_off"]) > 0)
    ref = geo.limb_los(z, nd, vm, zt, n_sub=200)
    colr = np.array([oracle.curgod(2, ref["nd"][a:b], ref["x"][a:b], vmr=ref["vmr"][0][a:b]) for a, b in zip(ref["pt_off"][:-1], ref["pt_off"][1:])])
    tot_ref = np.array([colr[ref["seg_off"][r]:ref["seg_off"][r + 1]].sum() for r in range(2)])
    e = [np.max(np.abs(t - tot_ref) / tot_ref) for t in totals]
    assert e[2] < 0.3 * e[1] < 0.3 * e[0] and e[2] < 5e-4, e
    # (iv) optical depth: tau of a step = 1e-3 per km of path here
    tau_of = lambda Lx: 1e-3 * (Lx["x"][Lx["pt_off"][1:] - 1] - Lx["x"][Lx["pt_off"][:-1]]) * 1e-5
    Lc = geo.calc_radtran_steps(z, T, P, nd, vm, zt, max_opt_depth=0.05, opt_depth_of=tau_of)
    assert np.all(tau_of(Lc) <= 0.05) and len(Lc["seg_layer"]) > len(L0["seg_layer"])
    assert abs(tau_of(Lc).sum() - tau_of(L0).sum()) < 1e-9
    # (v) 3-D state
    sun = geo.sun_in_local_frame(5.0, -12.0, 51.0)
    assert abs(sun[0] - np.cos(np.deg2rad(51.0))) < 1e-12 and abs(np.linalg.norm(sun) - 1.0) < 1e-12
    L3 = geo.limb_los_3d(z, nd, vm, zt, 51.0, [0.0, 90.0], tangent_lat_deg=5.0, subsolar_lat_deg=-12.0)
    mid = (L3["seg_off"][0] + L3["seg_off"][1]) // 2
    assert abs(L3["seg_mu"][mid] - np.cos(np.deg2rad(51.0))) < 0.05 and abs(L3["seg_lat"][mid] - 5.0) < 4.0
    s0, s1 = L3["seg_off"][0], L3["seg_off"][1]
    assert L3["seg_lat"][s1 - 1] - L3["seg_lat"][s0] > 20.0                      # the northward ray climbs in latitude
    assert np.ptp(L3["seg_lat"][L3["seg_off"][1]:]) < 8.0                          # the eastward one stays near its own
    assert list(geo.lat_box_index([-89.0, -30.0, 0.0, 29.9, 30.0, 74.0, 90.0])) == [0, 3, 3, 3, 4, 5, 6]
    with pytest.raises(ValueError):
        geo.sun_in_local_frame(60.0, -12.0, 10.0)


def test_partition_sum_derivative_and_level_rows():
    """CalcPartitionSum_dT: the exact derivative of the Lagrange cubic CalcPartitionSum evaluates (spect_classes.py:1692-1710),
    against central differences inside an interval of the table, one-sided at a table temperature (the interpolant
    changes its four points there); LevelFactored.unique_rows: distinct (P, T) couples and every step's row."""
    from spectrobot_amd import spect_classes as spcl, engine
    for mol, iso in ((6, 1), (23, 1), (5, 1)):
        for T in (61.0, 97.3, 150.0, 212.5, 296.0):
            h = 1e-4
            fd = (spcl.CalcPartitionSum(mol, iso, T + h) - spcl.CalcPartitionSum(mol, iso, T - h)) / (2 * h)
            assert abs(spcl.CalcPartitionSum_dT(mol, iso, T) - fd) <= 1e-8 * abs(fd), (mol, T)
        fd = (spcl.CalcPartitionSum(mol, iso, 110.0 + 1e-5) - spcl.CalcPartitionSum(mol, iso, 110.0)) / 1e-5
        assert abs(spcl.CalcPartitionSum_dT(mol, iso, 110.0) - fd) <= 1e-5 * abs(fd)
    v = spcl.CalcPartitionSum_dT(6, 1, np.array([150.0, 97.3, 150.0]))
    assert v.shape == (3,) and v[0] == v[2] == spcl.CalcPartitionSum_dT(6, 1, 150.0)
    T = np.array([150.0, 160.0, 150.0, 150.0, 160.0])
    P = np.array([1.0, 1.0, 2.0, 1.0, 1.0])
    Tr, Pr, row = engine.LevelFactored.unique_rows(T, P)
    assert len(Tr) == 3 and np.array_equal(Tr[row], T) and np.array_equal(Pr[row], P) and row.dtype == np.int32
    assert row[0] == row[3] and row[1] == row[4] and len(set(row)) == 3


def test_fov_closed_form_on_stacks():
    """smm.fov_closed_form (the retrieval loop integrates a pixel's radiances and all its derivatives at once) is
    FOV_integr_1D(closed_form=True) element by element, linear in the three spectra, and the integral the reference's
    quadrature approximates (2.5e-4 at its default tolerances, FOV_integr_1D's docstring)."""
    from spectrobot_amd import spect_main_module as smm
    from spectrobot_amd.retrieval import Spectrum
    rng = np.random.default_rng(3)
    stack = rng.uniform(1e-7, 3e-6, (3, 5, 14))        # three LOS x (1 + 4 derivatives) x 14 bands
    for rot in (0.0, 7.5, 20.0, 45.0):
        got = smm.fov_closed_form(stack[0], stack[1], stack[2], rot)
        assert got.shape == (5, 14)
        for p in range(5):
            one = smm.FOV_integr_1D([Spectrum(stack[q, p]) for q in range(3)], rot, closed_form=True)
            assert np.array_equal(one.spectrum, got[p])
        lin = smm.fov_closed_form(2.0 * stack[0] + stack[1], 2.0 * stack[1] + stack[2], 2.0 * stack[2] + stack[0], rot)
        ref = 2.0 * got + smm.fov_closed_form(stack[1], stack[2], stack[0], rot)
        assert np.max(np.abs(lin - ref) / np.abs(ref)) < 1e-14
    quad = smm.FOV_integr_1D([Spectrum(stack[q, 0]) for q in range(3)], 20.0, closed_form=False)
    assert np.max(np.abs(quad.spectrum / smm.fov_closed_form(stack[0, 0], stack[1, 0], stack[2, 0], 20.0) - 1)) < 1e-3
