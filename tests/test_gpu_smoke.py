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


@pytest.mark.gpu
@pytest.mark.parametrize("nwg", [1, 3, 17])
def test_any_number_of_resident_workgroups_makes_progress(nwg):
    """The progress argument of the ticket-scheduled dataflow launches (cholesky_dataflow.hip: tile_chol_kernel) does not
    depend on how many workgroups are resident: with the grid capped at 1, 3 or 17 workgroups -- far below what a task's
    dependencies would need if tasks were bound to workgroups in advance -- every kernel (factorisations on both tile sizes,
    batched launches, the two sweeps of the inverse, tile and vector solves) still finishes with the full-grid numbers."""
    import numpy as np
    import gpgradpy_amd
    from oracle import gp_oracle as orc
    n, d = 150, 8                                    # N = 1350: 22 tile columns of 64, 11 of 128
    X, f, g = orc.synthetic_design(n, d, seed=2)
    rows = np.random.default_rng(4).uniform(-2.0, -0.7, (5, d))
    xq = np.random.default_rng(5).uniform(-1, 1, (70, d))
    out = {}
    for cap in (0, nwg):
        res = []
        for mode in ("tile64", "tile128"):
            GP = gpgradpy_amd.GaussianProcess(d, True, "SqExp", "precon")
            GP.set_data(X, f, np.zeros(n), g, np.zeros((n, d)))
            GP.set_factor_mode(mode)
            GP.set_max_workgroups(cap)
            res.append(GP.calc_lkd_batch(rows))                                   # batched factorisation
            hp = GP.hp_vec2dataclass(GP.hp_info_optz_lkd, rows[0])
            info = GP.calc_lkd_all(hp, calc_grad=True)[0]                         # factorisation + inverse sweeps + vector solve
            res.append(np.concatenate(([info.ln_lkd], info.ln_lkd_grad)))
            GP.set_hpara("set", 0, hp_vals=GP.optz_closed_form_hp(hp))
            res.append(np.concatenate(GP.eval_model(xq)[:2]))                     # tile solves (70 points)
            res.append(np.concatenate(GP.eval_model(xq[:1], calc_grad=True)[:2])) # vector solves
            assert GP.factor_fallbacks() == 0
        out[cap] = res
    for a, b in zip(out[0], out[nwg]):
        np.testing.assert_array_equal(a, b)           # same tasks, same arithmetic: bit for bit
