"""Sharded multi-start likelihood evaluation (reference loop GpHparaX0.py:33-59).

Each restart row is an independent likelihood evaluation sharing only the read-only data, so rows are
split in contiguous blocks over the ranks (one process per GPU), every rank evaluates its block on its
own device with no data-path collective, and ONE all_gather of the per-rank ln_lkd slices (8 bytes per
row; latency-bound, not a bandwidth result) gives every rank the full table for the nanargmax.
With a process group on the "nccl" backend that all_gather is RCCL over xGMI.

Sharding is OPT-IN: a caller passes the process group it wants the rows split over (`group=`, or
`GaussianProcess.shard_restarts(group)` for the optimiser's calls).  Without a group every function here
works on the calling process alone, whatever else the program uses torch.distributed for.

A rank whose local evaluation raises still takes part in the collective (its rows travel as NaN together
with an error flag), so the other ranks never block in the all_gather; the failing rank re-raises
afterwards and the others get `RankFailure` naming it.
"""
import time

import numpy as np

# wall time of the most recent collective of this process (seconds) -- read by bench.py
last_collective_s = 0.0


class RankFailure(RuntimeError):
    """Another rank's local evaluation failed; its rows are NaN in the gathered table."""


def shard_rows(m, world_size, rank):
    """Contiguous block [lo, hi) of m rows owned by `rank` (blocks differ by at most one row)."""
    base, rem = divmod(m, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _group_info(group):
    """(world, rank, dist) of an explicit process group; (1, 0, None) when the caller did not ask for sharding."""
    if group is None:
        return 1, 0, None
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError('a process group was given but torch.distributed is not initialised')
    return dist.get_world_size(group), dist.get_rank(group), dist


def _collective_device(dist, group, device):
    import torch
    if device is not None:
        return torch.device(device)
    if dist.get_backend(group) == 'nccl':                      # RCCL: the buffers must live on this rank's GPU
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


def _gather_table(local_rows, m, w, group, device, failed):
    """all_gather of equal-size padded blocks [width, w + 1] (last column: 1 where the owning rank failed)."""
    global last_collective_s
    import torch
    world, rank, dist = _group_info(group)
    width = -(-m // world)
    dev = _collective_device(dist, group, device)
    host = np.full((width, w + 1), np.nan)
    host[:, w] = 1.0 if failed else 0.0
    k = local_rows.shape[0]
    if k > 0 and not failed:
        host[:k, :w] = local_rows
    buf = torch.from_numpy(host).to(dev)
    gathered = [torch.empty_like(buf) for _ in range(world)]
    t0 = time.perf_counter()
    dist.all_gather(gathered, buf, group=group)
    out = np.full((m, w), np.nan)
    bad = []
    for r in range(world):
        lo, hi = shard_rows(m, world, r)
        blk = gathered[r].cpu().numpy()
        if blk[0, w] > 0.5:
            bad.append(r)
        out[lo:hi] = blk[:hi - lo, :w]
    last_collective_s = time.perf_counter() - t0
    return out, bad


def gather_rows(local_rows, m, group=None, device=None, error=None):
    """All ranks of `group` contribute their contiguous block of an [m, w] table (shard_rows partition); every rank
    gets the full table.  One all_gather of equal-size padded blocks (latency-bound: a few hundred bytes per rank).
    `error`: the exception this rank's local work raised, if any -- the rank still joins the collective, then
    re-raises; the other ranks raise RankFailure.  group=None: no collective, the local rows are the table."""
    local_rows = np.atleast_2d(np.asarray(local_rows, dtype=np.float64))
    world, rank, dist = _group_info(group)
    if world == 1 and dist is None:
        if error is not None:
            raise error
        return local_rows
    out, bad = _gather_table(local_rows, m, local_rows.shape[1], group, device, error is not None)
    if error is not None:
        raise error
    if bad:
        raise RankFailure(f'rank(s) {bad} failed in their local evaluation; their rows are NaN')
    return out


def select_best_restart(hp_x0, eval_fn, group=None, device=None, return_failed=False):
    """Evaluate ln_lkd for every row of hp_x0 across the ranks of `group` and pick the best row.

    eval_fn(rows) -> ln_lkd[len(rows)] runs on the calling rank's device; the product passes
    `GaussianProcess.calc_lkd_batch`.  Returns (hp_best[1, n_hp], ln_lkd_all[m], idx_max) on every rank.
    Failed factorisations contribute NaN and are skipped by nanargmax (GpHparaX0.py:34,43-45,58).
    group=None: every row is evaluated by the calling process (no collective).
    """
    hp_x0 = np.atleast_2d(np.asarray(hp_x0, dtype=np.float64))
    m = hp_x0.shape[0]
    world, rank, dist = _group_info(group)
    if dist is None:
        ln_all = np.asarray(eval_fn(hp_x0), dtype=np.float64)
        bad = []
    else:
        lo, hi = shard_rows(m, world, rank)
        local, err = np.zeros(0), None
        try:
            if hi > lo:
                local = np.asarray(eval_fn(hp_x0[lo:hi]), dtype=np.float64)
        except Exception as e:          # noqa: BLE001 -- the rank must still reach the collective
            err = e
        table, bad = _gather_table(local.reshape(-1, 1), m, 1, group, device, err is not None)
        if err is not None:
            raise err
        ln_all = table[:, 0]
        if bad and not return_failed:
            raise RankFailure(f'rank(s) {bad} failed in their local evaluation; their rows are NaN')
    if np.all(np.isnan(ln_all)):
        raise RuntimeError('every restart row failed (all ln_lkd are NaN): no start point can be selected')
    idx = int(np.nanargmax(ln_all))
    if return_failed:
        return hp_x0[idx][None, :], ln_all, idx, bad
    return hp_x0[idx][None, :], ln_all, idx
