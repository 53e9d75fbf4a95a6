"""Sharded multi-start likelihood evaluation (reference loop GpHparaX0.py:33-59).

Each restart row is an independent likelihood evaluation sharing only the read-only data, so rows are
split in contiguous blocks over the ranks (one process per GPU), every rank evaluates its block on its
own device with no data-path collective, and ONE all_gather of the per-rank ln_lkd slices (8 bytes per
row; latency-bound, not a bandwidth result) gives every rank the full table for the nanargmax.
With torch.distributed initialised on the "nccl" backend that all_gather is RCCL over xGMI.
"""
import numpy as np


def shard_rows(m, world_size, rank):
    """Contiguous block [lo, hi) of m rows owned by `rank` (blocks differ by at most one row)."""
    base, rem = divmod(m, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def select_best_restart(hp_x0, eval_fn, group=None, device=None):
    """Evaluate ln_lkd for every row of hp_x0 across the ranks of `group` and pick the best row.

    eval_fn(rows) -> ln_lkd[len(rows)] runs on the calling rank's device; the product passes
    `GaussianProcess.calc_lkd_batch`.  Returns (hp_best[1, n_hp], ln_lkd_all[m], idx_max) on every rank.
    Failed factorisations contribute NaN and are skipped by nanargmax (GpHparaX0.py:34,43-45,58).
    """
    hp_x0 = np.atleast_2d(np.asarray(hp_x0, dtype=np.float64))
    m = hp_x0.shape[0]
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        ln_all = np.asarray(eval_fn(hp_x0), dtype=np.float64)
    else:
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        lo, hi = shard_rows(m, world, rank)
        local = np.asarray(eval_fn(hp_x0[lo:hi]), dtype=np.float64) if hi > lo else np.zeros(0)
        width = -(-m // world)                      # equal-size slots for all_gather
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' \
                else torch.device('cpu')
        buf = torch.full((width,), float('nan'), dtype=torch.float64, device=device)
        if hi > lo:
            buf[:hi - lo] = torch.from_numpy(local).to(device)
        gathered = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(gathered, buf, group=group)
        ln_all = np.full(m, np.nan)
        for r in range(world):
            rlo, rhi = shard_rows(m, world, r)
            ln_all[rlo:rhi] = gathered[r][:rhi - rlo].cpu().numpy()
    idx = int(np.nanargmax(ln_all))
    return hp_x0[idx][None, :], ln_all, idx


def gather_rows(local_rows, m, group=None, device=None):
    """All ranks contribute their contiguous block of an [m, w] table (shard_rows partition); every rank gets the full
    table.  One all_gather of equal-size padded blocks (latency-bound: a few hundred bytes per rank)."""
    import torch
    import torch.distributed as dist
    local_rows = np.atleast_2d(np.asarray(local_rows, dtype=np.float64))
    if not (dist.is_available() and dist.is_initialized()):
        return local_rows
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    w = local_rows.shape[1]
    width = -(-m // world)
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' \
            else torch.device('cpu')
    buf = torch.full((width, w), float('nan'), dtype=torch.float64, device=device)
    if local_rows.shape[0] > 0:
        buf[:local_rows.shape[0]] = torch.from_numpy(local_rows).to(device)
    gathered = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf, group=group)
    out = np.full((m, w), np.nan)
    for r in range(world):
        lo, hi = shard_rows(m, world, r)
        out[lo:hi] = gathered[r][:hi - lo].cpu().numpy()
    return out
