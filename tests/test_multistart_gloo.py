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

    hp_best, ln_all, idx = select_best_restart(z["hp_x0"], eval_with_failure, group=dist.group.WORLD)
    lo, hi = shard_rows(len(z["hp_x0"]), world, rank)
    assert calls == [hi - lo]
    # sharding is opt-in: without a group every rank evaluates all rows itself, whatever torch.distributed is used for
    calls.clear()
    _, ln_solo, idx_solo = select_best_restart(z["hp_x0"][:6], eval_fn)
    assert calls == [6] and idx_solo == int(np.nanargmax(ln_solo))
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


def _optz_worker(rank, world, port, out_dir):
    """The optimiser's multi-start (gpgradpy_amd/hpara_optz.py::optz_hp_max_lkd) sharded over two ranks.  The objective
    is an analytic stand-in for the device likelihood (this tests the sharding + gather, not the kernels)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from scipy.optimize import Bounds
    from gpgradpy_amd.hpara import HparaOptzInfo
    from gpgradpy_amd.hpara_optz import HparaOptz
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class Fake(HparaOptz):
        started = []

        def __init__(self):
            self.hp_info_optz_lkd = HparaOptzInfo(n_hp=2, has_theta=True, idx_theta=np.array([0, 1]), has_kernel=False, idx_kernel=None,
                                                  has_varK=False, idx_varK=None, has_var_fval=False, idx_var_fval=None,
                                                  has_var_fgrad=False, idx_var_fgrad=None, bvec_log_optz=np.array([True, True]))

        # two basins: the deeper one at (1, -1)
        def return_optz_val(self, x):
            x = np.atleast_1d(x)
            return float(min((x[0] + 2) ** 2 + (x[1] - 1) ** 2 + 1.0, 2 * (x[0] - 1) ** 2 + (x[1] + 1) ** 2))

        def return_optz_grad(self, x):
            x = np.atleast_1d(x)
            a = (x[0] + 2) ** 2 + (x[1] - 1) ** 2 + 1.0
            b = 2 * (x[0] - 1) ** 2 + (x[1] + 1) ** 2
            return np.array([2 * (x[0] + 2), 2 * (x[1] - 1)]) if a < b else np.array([4 * (x[0] - 1), 2 * (x[1] + 1)])

    def rows(self, X):           # the lock-step path's batched evaluator: (ln_lkd, gradient) = minus the objective
        self.batches.append(len(X))
        return [(-self.return_optz_val(x), -self.return_optz_grad(x)) for x in X]

    Fake.batches = []
    Fake._final_cond = lambda self, x: np.nan          # no device here
    Fake._objective_rows = rows
    gp = Fake()
    gp.shard_restarts = lambda g: setattr(gp, "restart_group", g)
    gp.shard_restarts(dist.group.WORLD)
    x0 = np.array([[-2.5, 1.5], [-1.5, 0.5], [0.5, -0.5], [2.0, -2.0], [0.9, -1.2]])
    best, cond, info = gp.optz_hp_max_lkd(x0, Bounds([-5, -5], [5, 5], keep_feasible=True))
    # rank 0 owns starts 0..2, rank 1 starts 3..4: their SLSQP runs advanced together, one batched call per round
    assert Fake.batches and max(Fake.batches) == (3 if rank == 0 else 2) and gp.optz_lockstep_batches == len(Fake.batches)
    gp.optz_lockstep = False     # and the sequential loop (the reference's order) ends at the same optimum
    best_seq, _, _ = gp.optz_hp_max_lkd(x0, Bounds([-5, -5], [5, 5], keep_feasible=True))
    np.testing.assert_array_equal(best_seq, best)
    np.save(os.path.join(out_dir, f"best_{rank}.npy"), best)
    np.save(os.path.join(out_dir, f"obj_{rank}.npy"), gp.optz_obj_all_last)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_optimiser_multistart_sharded_two_ranks_gloo(tmp_path):
    world = 2
    mp.spawn(_optz_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    b0, b1 = np.load(tmp_path / "best_0.npy"), np.load(tmp_path / "best_1.npy")
    o0, o1 = np.load(tmp_path / "obj_0.npy"), np.load(tmp_path / "obj_1.npy")
    np.testing.assert_array_equal(b0, b1)                      # every rank ends with the same optimum ...
    np.testing.assert_array_equal(o0, o1)                      # ... and the full table of the five runs
    np.testing.assert_allclose(b0, [1.0, -1.0], atol=1e-5)     # the deeper basin, found by a start of rank 1's block
    assert np.all(np.isfinite(o0)) and o0.shape == (5,)
    assert np.isclose(o0[:2], 1.0, atol=1e-8).all() and np.isclose(o0[2:], 0.0, atol=1e-8).all()


def _failing_worker(rank, world, port, out_dir):
    """Rank 1's local evaluation raises: it must still join the all_gather (no hang on rank 0), then re-raise; rank 0
    gets RankFailure naming the failed rank."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import datetime
    import torch.distributed as dist
    from gpgradpy_amd.multistart import RankFailure, gather_rows, select_best_restart
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))

    def eval_fn(rows):
        if rank == 1:
            raise ValueError("device evaluation failed on this rank")
        return -np.sum(rows ** 2, axis=1)

    x0 = np.arange(12.0).reshape(6, 2)
    outcome = []
    try:
        select_best_restart(x0, eval_fn, group=dist.group.WORLD)
        outcome.append("returned")
    except ValueError:
        outcome.append("own-error")
    except RankFailure as e:
        outcome.append("rank-failure:" + str(e))
    # tolerant form: the caller asks for the table and the list of failed ranks
    if rank == 0:
        hp_best, ln_all, idx, bad = select_best_restart(x0, eval_fn, group=dist.group.WORLD, return_failed=True)
        assert bad == [1] and np.isnan(ln_all[3:]).all() and idx == 0
    else:
        try:
            select_best_restart(x0, eval_fn, group=dist.group.WORLD, return_failed=True)
        except ValueError:
            pass
    try:
        gather_rows(np.ones((3, 2)) * rank, 6, group=dist.group.WORLD, error=RuntimeError("boom") if rank == 1 else None)
        outcome.append("returned")
    except RuntimeError as e:
        outcome.append(type(e).__name__)
    with open(os.path.join(out_dir, f"outcome_{rank}.txt"), "w") as fh:
        fh.write("|".join(outcome))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_failing_rank_still_joins_the_collective(tmp_path):
    world = 2
    mp.spawn(_failing_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    o0 = open(tmp_path / "outcome_0.txt").read().split("|")
    o1 = open(tmp_path / "outcome_1.txt").read().split("|")
    assert o0[0].startswith("rank-failure:") and "[1]" in o0[0] and o0[1] == "RankFailure"
    assert o1 == ["own-error", "RuntimeError"]


def test_all_rows_failed_is_a_clear_error():
    from gpgradpy_amd.multistart import select_best_restart
    with pytest.raises(RuntimeError, match="every restart row failed"):
        select_best_restart(np.zeros((3, 2)), lambda rows: np.full(len(rows), np.nan))
