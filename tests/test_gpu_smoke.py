import pytest


@pytest.mark.gpu
def test_graft_entry_smoke():
    import __graft_entry__ as g
    g.smoke()


def _shared_gpu_worker(rank, q):
    import numpy as np
    import gpgradpy_amd
    from oracle import gp_oracle as orc
    n, d = 1100, 8                                    # N = 9900 > 9216: the 128-tile dataflow kernel
    X, f, g = orc.synthetic_design(n, d, seed=11)
    GP = gpgradpy_amd.GaussianProcess(d, True, 'SqExp', 'precon')
    GP.set_data(X, f, np.zeros(n), g, np.zeros((n, d)))
    hp_x0 = np.random.default_rng(3).uniform(-2.0, -0.7, (6, d))
    ln = GP.calc_lkd_batch(hp_x0)
    # posterior solves are dataflow launches too: a burst of single-point evaluations next to the other process
    hp = GP.optz_closed_form_hp(GP.hp_vec2dataclass(GP.hp_info_optz_lkd, hp_x0[0]))
    GP.set_hpara('set', 0, hp_vals=hp)
    xq = np.random.default_rng(5).uniform(-1.0, 1.0, (40, d))
    post = np.array([GP.eval_model(xq[i:i + 1])[:2] for i in range(40)]).reshape(40, 2)
    q.put((rank, np.concatenate([ln, post.ravel()]), GP.factor_fallbacks()))


@pytest.mark.gpu
def test_two_processes_share_one_gpu():
    """Two processes run dataflow launches on the same device at the same time.  The persistent workgroups take their
    tasks from a ticket counter and only ever wait for smaller tickets, so a launch needs nobody else's slots to make
    progress (cholesky_dataflow.hip: tile_chol_kernel): neither process may have timed out into the blocked fallback
    (factor_fallbacks() == 0), and both must get the single-process numbers."""
    import multiprocessing as mp
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_shared_gpu_worker, args=(r, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=300) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    q1 = ctx.Queue()
    p = ctx.Process(target=_shared_gpu_worker, args=(0, q1))
    p.start()
    _, ref, fb = q1.get(timeout=300)
    p.join(timeout=60)
    assert fb == 0
    for _, ln, fb2 in res:
        assert fb2 == 0                                # sharing the device costs time, never the schedule
        assert np.all(np.isfinite(ln))
        np.testing.assert_allclose(ln[:6], ref[:6], rtol=1e-9)
        np.testing.assert_allclose(ln[6:], ref[6:], rtol=1e-6, atol=1e-8)
