"""GPU (MI355X): the multi-GPU leg's code path on the ONE GPU a test box has -- torch.distributed on the `nccl`
backend (RCCL) with world_size 1: the restart sharding, the device-tensor all_gather, the optimiser's sharded
multi-start and bench.py's own `nccl` branch all execute here, so that the first 8-GPU run is not their first run
(reference loop GpHparaX0.py:33-59)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN_DIR, ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def nccl_world1():
    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def test_select_best_restart_over_rccl(nccl_world1):
    dist = nccl_world1
    import gpgradpy_amd
    from gpgradpy_amd import multistart
    assert dist.get_backend() == "nccl"
    z = np.load(os.path.join(GOLDEN_DIR, "multistart_SqExp_n64_d4.npz"))
    n, d = z["x"].shape
    GP = gpgradpy_amd.GaussianProcess(d, True, "SqExp", "precon")
    GP.set_data(z["x"], z["f"], np.zeros(n), z["g"], np.zeros((n, d)))
    seen = []
    real = multistart._collective_device

    def spy(dist_, group, device):
        dev = real(dist_, group, device)
        seen.append(dev.type)
        return dev
    multistart._collective_device = spy
    try:
        hp_best, ln_all, idx = multistart.select_best_restart(z["hp_x0"], GP.calc_lkd_batch, group=dist.group.WORLD)
        rows = np.column_stack((np.arange(5.0), np.arange(5.0) ** 2))
        table = multistart.gather_rows(rows, 5, group=dist.group.WORLD)
    finally:
        multistart._collective_device = real
    assert seen == ["cuda", "cuda"]                        # the collectives ran on device tensors (RCCL), not on the host
    np.testing.assert_allclose(ln_all, z["ln_lkd_all"], rtol=1e-8)          # the reference's own table (GpHparaX0.py:39-45)
    assert idx == int(z["idx_max"])
    np.testing.assert_array_equal(hp_best[0], z["hp_x0"][idx])
    np.testing.assert_array_equal(table, rows)
    assert multistart.last_collective_s > 0.0


def test_optimiser_sharded_over_rccl_matches_unsharded(nccl_world1):
    """set_hpara('optz') with the restarts sharded over the (one-rank) RCCL group gives the optimum of the unsharded run."""
    dist = nccl_world1
    import gpgradpy_amd
    from oracle import gp_oracle as orc
    n, d = 16, 2
    X, f, g = orc.synthetic_design(n, d, seed=3)
    out = []
    for group in (None, dist.group.WORLD):
        GP = gpgradpy_amd.GaussianProcess(d, True, "SqExp", "precon")
        GP.lkd_hp_best_n_eval = 12
        GP.lkd_optz_start_mtd = "lhs"
        GP.optz_n_x0 = 3
        GP.init_optz_surr(2)
        GP.shard_restarts(group)
        GP.set_data(X, f, np.zeros(n), g, np.zeros((n, d)))
        GP.set_hpara("optz", 0)
        out.append((GP.hp_vals.theta.copy(), GP.optz_obj_all_last.copy()))
    np.testing.assert_allclose(out[1][0], out[0][0], rtol=1e-9)
    np.testing.assert_allclose(out[1][1], out[0][1], rtol=1e-9)


def test_bench_force_dist_runs_the_nccl_branch():
    """bench.py --gpus 1 --force-dist initialises `nccl` with world_size 1 and goes through the sharded selection."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--config", "cfg2",
                        "--steps", "16", "--warmup", "16", "--no-cpu-baseline"], capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["collective_backend"] == "nccl"
    assert line["collective_ms"] > 0.0 and len(line["per_rank_ms"]) == 2
    assert line["value"] > 0


def _bench_line(r):
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks():
    """`python3 bench.py --gpus 2` with NO launcher and no WORLD_SIZE (the shape of the driver's command): the parent,
    which never touches the GPU, starts two fresh rank processes of itself, relays rank 0's single JSON line and
    returns 0.  Two ranks on the box's one GPU, hence gloo; on an 8-GPU node the same code picks nccl (RCCL)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--config", "cfg2",
                        "--steps", "8", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    line = _bench_line(r)
    assert line["n_gpus"] == 2 and line["config"]["collective_backend"] == "gloo"
    assert line["factor_fallbacks"] == 0
    assert len(line["per_rank_ms"]) == 2 and line["value"] > 0 and line["steps"] == 8


def test_bench_under_torch_distributed_run():
    """The documented launcher form still works: WORLD_SIZE is set, so bench.py does not spawn anything itself."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2",
                        "--backend", "gloo", "--config", "cfg2", "--steps", "8", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900)
    line = _bench_line(r)
    assert line["n_gpus"] == 2 and line["factor_fallbacks"] == 0


def test_bench_self_launch_reports_a_failing_rank():
    """A rank that cannot run (nccl with more ranks than GPUs) makes the self-launching parent return non-zero."""
    import torch
    ndev = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ndev + 1), "--backend", "nccl",
                        "--config", "cfg2", "--steps", "2", "--warmup", "0", "--no-cpu-baseline"], capture_output=True,
                       text=True, env=env, timeout=600)
    assert r.returncode != 0 and "rank(s) failed" in r.stderr


def test_bench_refuses_more_ranks_than_gpus_under_nccl():
    import torch
    ndev = torch.cuda.device_count()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK=str(ndev), LOCAL_RANK=str(ndev),
               WORLD_SIZE=str(ndev + 1))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ndev + 1), "--backend", "nccl",
                        "--config", "cfg2", "--steps", "2", "--warmup", "0", "--no-cpu-baseline"], capture_output=True,
                       text=True, env=env, timeout=300)
    assert r.returncode != 0 and "LOCAL_RANK" in r.stderr


def test_multi_device_driver_in_one_process():
    """gpg_multi_*: the restart table sharded over the devices of ONE process, below Python.  A test box has
    one GPU, so the device list names it twice: two contexts factorise their halves of the reference's 64-row table
    CONCURRENTLY on one device (which the ticket-scheduled dataflow launches allow) and must reproduce the reference's
    table and its argmax."""
    import ctypes as C
    import gpgradpy_amd
    from gpgradpy_amd import _lib
    lib = _lib.load()
    z = np.load(os.path.join(GOLDEN_DIR, "multistart_SqExp_n64_d4.npz"))
    n, d = z["x"].shape
    GP = gpgradpy_amd.GaussianProcess(d, True, "SqExp", "precon")          # host side only: row decoding and the nugget
    GP.set_data(z["x"], z["f"], np.zeros(n), z["g"], np.zeros((n, d)))
    rows = np.ascontiguousarray(GP._rows_from_hp_x0(z["hp_x0"]))
    y = np.ascontiguousarray(GP.make_data_vec(z["f"], z["g"]))
    x = np.ascontiguousarray(z["x"])
    for devices in ([0], [0, 0], [0, 0, 0]):
        m = C.c_void_p()
        devs = (C.c_int * len(devices))(*devices)
        rc = lib.gpg_multi_create(C.byref(m), len(devices), devs, n, d, 1, 0)
        assert rc == 0, lib.gpg_multi_last_error(None)
        try:
            assert lib.gpg_multi_count(m) == len(devices)
            assert lib.gpg_multi_set_data(m, _lib.as_dp(x), _lib.as_dp(y), None) == 0
            outs = (_lib.GpgLkdOut * len(rows))()
            best = C.c_int(-2)
            rc = lib.gpg_multi_lkd_batch(m, len(rows), _lib.as_dp(rows), rows.shape[1], float(GP._etaK), 1, 1, outs, C.byref(best))
            assert rc == 0, lib.gpg_multi_last_error(m)
            ln = np.array([o.ln_lkd for o in outs])
            np.testing.assert_allclose(ln, z["ln_lkd_all"], rtol=1e-8)
            assert best.value == int(z["idx_max"])
        finally:
            lib.gpg_multi_destroy(m)
    bad = C.c_void_p()
    assert lib.gpg_multi_create(C.byref(bad), 1, (C.c_int * 1)(99), n, d, 1, 0) != 0 and b"device" in lib.gpg_multi_last_error(None)
