"""Spectral-window sharding over the GPUs of one node (one process per GPU).

The path shards naturally: every output grid point depends only on
the lines within -6504..+6505 grid points and on its own layer stack -- the
reference itself already splits the grid into n_split contiguous chunks
(spect_main_module.py:1619-1630, 2814-2818).  Rank r of W owns the contiguous
grid range shard_bounds(n_grid, W, r); the line list (8 MB) and the atmosphere
are replicated; the only exchange step is ONE all-gather of the radiance shards
(RCCL over xGMI with backend "nccl", gloo on CPU for tests).
"""
import os

import torch
import torch.distributed as dist


def shard_bounds(n_grid, world_size, rank):
    """Contiguous, balanced: the first n_grid % W ranks own one extra point."""
    q, r = divmod(int(n_grid), int(world_size))
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_bounds_balanced(line_freq, grid, world_size, zone_weight=1.5, align=64, point_weight=0.0):
    """Contiguous shards of (about) equal WORK instead of equal width (balance by line-window
    work): the cost of a grid point is the number of lines whose 13010-point window covers it (far-field and
    near-wings kernels) plus zone_weight x 13010 x the number of line centres on it (the zones kernel's
    ~240 region-2/3/4 evaluations per line and layer cost about zone_weight times the line's whole
    region-1 share) plus point_weight x 13010 for the grid point itself.  Measured on the band-head list of
    tools/balanced_shards.py (profiles/r03_shards_one_gpu.txt): point_weight = 0 gives max / mean 1.12 over the 8
    shards, 2.0 gives 1.41 -- a sparse stretch is cheaper per line than the model says, and below ~0.6 ms a step is
    bound by the host's enqueue time (0.45 ms), which no split of the grid changes.  Real HITRAN line lists bunch in
    band centres: equal-width shards then differ several-fold.
    Returns [(lo, hi)] for all ranks; boundaries are multiples of `align` grid points."""
    import numpy as np
    grid = np.asarray(grid, dtype=float)
    n, half = grid.size, 6505
    world_size, align = int(world_size), int(align)
    if n < world_size:
        raise ValueError("%d grid points cannot be split over %d ranks" % (n, world_size))
    while align > 1 and n < world_size * align:   # short grids: finer boundaries instead of negative / overlapping shards
        align //= 2
    ic = np.clip(np.rint((np.asarray(line_freq, dtype=float) - grid[0]) / (grid[1] - grid[0])).astype(np.int64), 0, n - 1)
    centres = np.bincount(ic, minlength=n).astype(float)
    cum = np.concatenate([[0.0], np.cumsum(centres)])
    cover = cum[np.minimum(np.arange(n) + half, n)] - cum[np.maximum(np.arange(n) - half + 1, 0)]   # lines covering j
    cost = cover + zone_weight * 13010.0 * centres + point_weight * 13010.0
    c = np.concatenate([[0.0], np.cumsum(cost)])
    cuts = [0]
    for r in range(1, int(world_size)):
        j = int(np.searchsorted(c, c[-1] * r / world_size))
        j = int(round(j / align)) * align
        cuts.append(min(max(j, cuts[-1] + align), n - (world_size - r) * align))
    cuts.append(n)
    assert all(0 <= a < b <= n for a, b in zip(cuts[:-1], cuts[1:])), cuts
    return [(cuts[r], cuts[r + 1]) for r in range(world_size)]


def shard_costs(line_freq, grid, bounds, zone_weight=1.5, point_weight=0.0):
    """The work model of shard_bounds_balanced summed over each of the given shards [(lo, hi)]: per-shard cost (arbitrary
    units) and the number of lines whose 13010-point window meets the shard (what the rank prepares records for)."""
    import numpy as np
    grid = np.asarray(grid, dtype=float)
    n, half = grid.size, 6505
    ic = np.clip(np.rint((np.asarray(line_freq, dtype=float) - grid[0]) / (grid[1] - grid[0])).astype(np.int64), 0, n - 1)
    centres = np.bincount(ic, minlength=n).astype(float)
    cum = np.concatenate([[0.0], np.cumsum(centres)])
    cover = cum[np.minimum(np.arange(n) + half, n)] - cum[np.maximum(np.arange(n) - half + 1, 0)]
    c = np.concatenate([[0.0], np.cumsum(cover + zone_weight * 13010.0 * centres + point_weight * 13010.0)])
    costs = [float(c[hi] - c[lo]) for lo, hi in bounds]
    lines = [int(cum[min(hi + half, n)] - cum[max(lo - (half - 1), 0)]) for lo, hi in bounds]
    return costs, lines


def init_from_env(backend=None, single_rank_group=False):
    """torch.distributed rendezvous from RANK / WORLD_SIZE / MASTER_* (torchrun).  single_rank_group: form a
    process group even for WORLD_SIZE = 1 (the RCCL hardware test on a one-GPU box)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or single_rank_group) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # SR_DIST_BACKEND=gloo lets several ranks share ONE GPU (rehearsal on a 1-GPU box;
            # RCCL needs one device per rank)
            backend = os.environ.get("SR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            # (the device the rank's engine uses: engine.set_device(local % device_count) -- a launcher that narrows
            # each rank's HIP_VISIBLE_DEVICES to one GPU leaves LOCAL_RANK beyond the visible count)
            kw["device_id"] = torch.device("cuda", local % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local, world


def dist_info():
    """What the process group really is (bench lines: shows that RCCL saw N ranks)."""
    if not dist.is_initialized():
        return {"backend": None, "world_size": 1, "rank": 0}
    return {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rank": dist.get_rank()}


_pending = []   # (Work, shard, out) of collectives issued with async_op=True, oldest first (wait_gathers)
_MAX_IN_FLIGHT = 4
stats = {"async_gathers": 0, "blocking_gathers": 0, "evicted_waits": 0}   # which branch ran (tests, bench line)


def wait_gathers():
    """Make the current stream wait for every all-gather issued with async_op=True; the handles and the
    references that kept their input / output tensors alive are dropped only after their wait()."""
    while _pending:
        work, _shard, _out = _pending.pop(0)
        work.wait()


def all_gather_spectrum(shard, n_grid, world_size, rank, out=None, bounds=None, async_op=False, force_collective=False):
    """Reassemble [n_rays, n_grid] from per-rank shards [n_rays, hi-lo] with a single
    all-gather.  Shards may differ in size (by one point with shard_bounds, freely with
    bounds = shard_bounds_balanced(...)), so each rank contributes a buffer padded to the
    largest shard.

    async_op=True (RCCL backend, equal shards, one ray; otherwise ignored): the collective is enqueued behind the work that
    produces `shard` but the caller's stream does not wait for it -- the next step's kernels run beside it
    (at 8 GPUs the latency-bound 100 KB gather is a few per cent of a 1 ms step).  `out` is valid after
    wait_gathers() (or a device synchronisation); consecutive gathers into the same `out` are ordered (RCCL
    runs a communicator's collectives in issue order on one stream).  Every Work handle is kept, together with
    its input shard and its output, until it has been waited on: at most _MAX_IN_FLIGHT are outstanding, older
    ones are waited on (a stream-side wait on a collective that finished steps ago) before the next is issued.

    force_collective: go through the collective even when world_size == 1 (a one-rank process group: the
    hardware test of the RCCL branch on a one-GPU box)."""
    if world_size == 1 and not force_collective:
        return shard
    n_rays = shard.shape[0]
    if bounds is None:
        bounds = [shard_bounds(n_grid, world_size, r) for r in range(world_size)]
    bounds = [(int(lo), int(hi)) for lo, hi in bounds]
    if len(bounds) != world_size or bounds[0][0] != 0:
        raise ValueError("shard bounds must be one (lo, hi) per rank starting at 0, got %r for %d ranks" % (bounds, world_size))
    for (lo, hi), (lo2, _hi2) in zip(bounds, list(bounds[1:]) + [(n_grid, n_grid)]):
        if not (0 <= lo <= hi <= n_grid and hi == lo2):
            raise ValueError("shard bounds must tile [0, n_grid) in rank order, got %r" % (bounds,))
    if shard.shape[1] != bounds[rank][1] - bounds[rank][0]:
        raise ValueError("rank %d holds %d points, its bounds %r say %d" % (rank, shard.shape[1], bounds[rank],
                                                                             bounds[rank][1] - bounds[rank][0]))
    q = max(hi - lo for lo, hi in bounds)
    staged = shard.is_cuda and dist.get_backend() == "gloo"  # rehearsal only: stage through the host
    if out is None:
        out = torch.empty((n_rays, n_grid), dtype=shard.dtype, device=shard.device)
    if all(hi - lo == q for lo, hi in bounds) and not staged and shard.is_contiguous():
        # equal shards (the bench: 1e5 points over 1, 2, 4, 8 ranks): no padding, no per-rank copies
        if n_rays == 1:
            if async_op and dist.get_backend() == "nccl":   # one collective stream: in order (gloo's workers are not)
                while len(_pending) >= _MAX_IN_FLIGHT:
                    work, _s, _o = _pending.pop(0)
                    work.wait()
                    stats["evicted_waits"] += 1
                work = dist.all_gather_into_tensor(out.view(world_size, q), shard.view(1, q), async_op=True)
                _pending.append((work, shard, out))     # shard / out stay referenced until the wait
                stats["async_gathers"] += 1
                return out
            dist.all_gather_into_tensor(out.view(world_size, q), shard.view(1, q))  # lands in place
            stats["blocking_gathers"] += 1
            return out
        flat = torch.empty((world_size, n_rays, q), dtype=shard.dtype, device=shard.device)
        dist.all_gather_into_tensor(flat.view(world_size * n_rays, q), shard)
        stats["blocking_gathers"] += 1
        out.view(n_rays, world_size, q).copy_(flat.permute(1, 0, 2))                # one kernel
        return out
    pad = torch.zeros((n_rays, q), dtype=shard.dtype, device=shard.device)
    pad[:, :shard.shape[1]] = shard
    if staged:
        flat_h = torch.empty((world_size * n_rays, q), dtype=shard.dtype)
        dist.all_gather_into_tensor(flat_h, pad.cpu())
        flat = flat_h.to(shard.device)
    else:
        flat = torch.empty((world_size * n_rays, q), dtype=shard.dtype, device=shard.device)
        dist.all_gather_into_tensor(flat, pad)  # concatenation along dim 0, rank-major
    stats["blocking_gathers"] += 1
    gathered = flat.view(world_size, n_rays, q)
    for r, (lo, hi) in enumerate(bounds):
        out[:, lo:hi] = gathered[r, :, :hi - lo]
    return out


def all_reduce_sum(t, force_collective=False):
    """In-place sum of a tensor over the ranks (the one exchange of a sharded retrieval iteration: partial
    instrument-band sums of radiances and Jacobians, spectrobot_amd.retrieval.simulate); a no-op without a
    process group.  A CUDA tensor under a gloo group (several ranks rehearsing on one GPU) goes through the host.
    force_collective: go through the collective even in a one-rank group (the RCCL hardware test on a one-GPU box)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force_collective):
        return t
    if t.is_cuda and dist.get_backend() == "gloo":
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
        return t
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t
