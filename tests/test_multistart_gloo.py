"""CPU: the N>1 multi-start path (row sharding + one all_gather + nanargmax) over gloo, world_size 2.
The per-rank evaluator is the oracle here (this is a test of the collective logic, not of the kernels)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import GOLDEN_DIR, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fixture, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from gpgradpy_amd.multistart import select_best_restart, shard_rows
    from oracle import gp_oracle as orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z = np.load(fixture)
    X, y = z["x"], orc.make_data_vec(z["f"], z["g"])
    calls = []

    def eval_fn(rows):
        calls.append(len(rows))
        ln, _ = orc.multistart_lkd(X, y, "SqExp", float(z["etaK"]), rows)
        # rows that fail on this rank stay NaN
        return ln

    # inject one failing row (NaN) to exercise nanargmax on rank 0's shard
    def eval_with_failure(rows):
        ln = eval_fn(rows)
        if rank == 0:
            ln[0] = np.nan
        return ln

    hp_best, ln_all, idx = select_best_restart(z["hp_x0"], eval_with_failure)
    lo, hi = shard_rows(len(z["hp_x0"]), world, rank)
    assert calls == [hi - lo]
    np.save(os.path.join(out_dir, f"ln_{rank}.npy"), ln_all)
    np.save(os.path.join(out_dir, f"idx_{rank}.npy"), np.array([idx]))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_multistart_two_ranks_gloo(tmp_path):
    fixture = os.path.join(GOLDEN_DIR, "multistart_SqExp_n64_d4.npz")
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), fixture, str(tmp_path)), nprocs=world, join=True)
    z = np.load(fixture)
    ln0, ln1 = np.load(tmp_path / "ln_0.npy"), np.load(tmp_path / "ln_1.npy")
    np.testing.assert_array_equal(ln0, ln1)          # every rank holds the full table
    assert np.isnan(ln0[0])
    np.testing.assert_allclose(ln0[1:], z["ln_lkd_all"][1:], rtol=1e-8)
    assert int(np.load(tmp_path / "idx_0.npy")[0]) == int(z["idx_max"]) == int(np.load(tmp_path / "idx_1.npy")[0])
