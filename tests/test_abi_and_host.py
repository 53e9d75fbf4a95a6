"""CPU: the C-ABI library loads and exports every symbol include/gpgrad.h declares (no compute calls),
the host-side logic mirrors the reference's bookkeeping, and the product fails loudly without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN_DIR, ROOT
import gpgradpy_amd
from gpgradpy_amd import _lib
from gpgradpy_amd.multistart import shard_rows


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "gpgrad.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gpg_[a-z_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    syms = _header_symbols()
    assert set(syms) == set(_lib.ABI_SYMBOLS)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/gpgrad.h but not exported"


def test_struct_layout_matches_header():
    assert ctypes.sizeof(_lib.GpgLkdOut) == 5 * 8 + 2 * 4
    assert ctypes.sizeof(_lib.GpgHp) == 8 + 4 * 8 + 2 * 4 + 8          # ... + hp_kernel (alpha of RatQu), appended last
    assert _lib.GpgHp.hp_kernel.offset == 48


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


@pytest.mark.skipif(not _no_gpu(), reason="only meaningful on a machine without a GPU")
def test_product_path_fails_loudly_without_gpu():
    GP = gpgradpy_amd.GaussianProcess(2, True, 'SqExp', 'precon')
    x = np.random.default_rng(0).uniform(-1, 1, (5, 2))
    with pytest.raises(_lib.GpgError):
        GP.set_data(x, x[:, 0], np.zeros(5), x, np.zeros((5, 2)))


def _gp_host_only(dim, kernel, n, noise):
    """Run set_data up to the device push (which raises on a CPU-only box) to inspect the host state."""
    GP = gpgradpy_amd.GaussianProcess(dim, True, kernel, 'precon')
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, (n, dim))
    f, g = x[:, 0] ** 2, rng.standard_normal((n, dim))
    if noise == 'none':
        sf, sg = np.zeros(n), np.zeros((n, dim))
    elif noise == 'known':
        sf, sg = np.full(n, 1e-2), np.full((n, dim), 1e-1)
    else:
        sf, sg = None, None
    try:
        GP.set_data(x, f, sf, g, sg)
    except _lib.GpgError:
        pass
    return GP


@pytest.mark.parametrize("noise,n_hp", [("none", 3), ("known", 4), ("unknown", 6)])
def test_hp_index_map(noise, n_hp):
    """SURVEY.md 8a15 probe (d=3, Ma5f2): n_hp = 3 / 4 / 6, all entries log10 by default."""
    GP = _gp_host_only(3, 'Ma5f2', 6, noise)
    info = GP.hp_info_optz_lkd
    assert info.n_hp == n_hp and info.bvec_log_optz.all()
    row = np.linspace(-2, -1, n_hp)
    hp = GP.hp_vec2dataclass(info, row)
    np.testing.assert_allclose(hp.theta, 10.0 ** row[:3])
    rows = GP._rows_from_hp_x0(row[None, :])
    np.testing.assert_allclose(rows[0, :3], 10.0 ** row[:3])
    if noise == 'none':
        assert hp.varK is None and rows[0, 3] == 1.0 and rows[0, 4] == -1.0 and rows[0, 5] == -1.0
    elif noise == 'known':
        assert np.isclose(hp.varK, 10.0 ** row[3]) and rows[0, 4] == -1.0
    else:
        assert np.isclose(hp.var_fval, 10.0 ** row[4]) and np.isclose(rows[0, 5], 10.0 ** row[5])


def test_nugget_table_matches_reference():
    rows = np.load(os.path.join(GOLDEN_DIR, "nugget_table.npz"))["rows"]
    for k, n, d, eb, eg in rows:
        GP = gpgradpy_amd.GaussianProcess(int(d), True, 'SqExp' if k == 0 else 'Ma5f2', 'precon')
        b, g = GP.calc_nugget(int(n))
        assert np.isclose(b, eb, rtol=1e-15) and np.isclose(g, eg, rtol=1e-14)


def test_hp_index_map_with_kernel_hyperparameter():
    """RatQu carries its own hyperparameter between theta and varK (GpHparaOptz.py:90-96); values probed from the
    reference (d=3, known noise): n_hp = 5, idx_kernel = [3], idx_varK = 4, all log10, etaK as SqExp."""
    GP = _gp_host_only(3, 'RatQu', 6, 'known')
    i = GP.hp_info_optz_lkd
    assert i.n_hp == 5 and list(i.idx_theta) == [0, 1, 2] and list(i.idx_kernel) == [3] and i.idx_varK == 4
    assert i.bvec_log_optz.all()
    hp = GP.hp_vec2dataclass(i, np.array([-1.0, -0.5, -0.2, 0.3, 0.1]))
    np.testing.assert_allclose(hp.theta, [0.1, 0.31622777, 0.63095734], rtol=1e-7)
    np.testing.assert_allclose(hp.kernel, [1.99526231], rtol=1e-8)
    np.testing.assert_allclose(hp.varK, 1.2589254117941673, rtol=1e-14)
    assert np.isclose(GP._etaK, 9.677056687390654e-10, rtol=1e-14)
    rows = GP._rows_from_hp_x0(np.array([[-1.0, -0.5, -0.2, 0.3, 0.1]]))
    np.testing.assert_allclose(rows[0], [0.1, 0.31622777, 0.63095734, 1.2589254117941673, -1.0, -1.0, 1.99526231], rtol=1e-7)


def test_out_of_scope_features_raise():
    GPr = gpgradpy_amd.GaussianProcess(2, True, 'RatQu')               # value path only: its own hyperparameter alpha
    assert GPr.kernel_has_hp and GPr.hp_kernel_default == 2 and GPr.hp_kernel_range == [1e-3, 10]   # KernelRatQuad.py:849-850
    with pytest.raises(Exception):
        gpgradpy_amd.GaussianProcess(2, True, 'Cubic')                  # Kernel.py:108-109
    with pytest.raises(AssertionError):
        gpgradpy_amd.GaussianProcess(2, True, 'SqExp', 'req_vmin')      # rejected like GaussianProcess.py:194
    GP = _gp_host_only(2, 'SqExp', 4, 'none')
    hp = GP.make_hp_class(theta=np.array([0.5, 0.5]))
    GP.cond_norm = 'one'
    with pytest.raises(Exception, match='cond_norm must be either 2 or "fro"'):
        GP.calc_lkd_all(hp, calc_cond=True)                            # GpHparaCon.py:159


def test_data_vec_layout():
    f = np.array([1.0, 2.0])
    g = np.array([[3.0, 5.0], [4.0, 6.0]])
    np.testing.assert_array_equal(gpgradpy_amd.GaussianProcess.make_data_vec(f, g), [1, 2, 3, 4, 5, 6])


def test_shard_rows_partition():
    for m in (1, 7, 64, 65):
        for w in (1, 2, 3, 8):
            spans = [shard_rows(m, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == m
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_rows(64, 8, 3) == (24, 32)     # BASELINE cfg4: 8 rows per GPU in rank order


def test_lanczos_reports_non_convergence():
    """cond_number.lanczos_largest must not hand back an unconverged Ritz value silently (k_max reached)."""
    import warnings
    from gpgradpy_amd.cond_number import LanczosNotConverged, lanczos_largest
    rng = np.random.default_rng(0)
    Q, _ = np.linalg.qr(rng.standard_normal((200, 200)))
    A = (Q * np.linspace(1.0, 1.0001, 200)) @ Q.T               # clustered spectrum: five steps cannot resolve the top
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        lanczos_largest(lambda v: A @ v, 200, k_max=5, k_cap=5, max_restarts=0)
    assert any(issubclass(x.category, LanczosNotConverged) for x in w)
    B = (Q * np.linspace(1.0, 100.0, 200)) @ Q.T
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        lam = lanczos_largest(lambda v: B @ v, 200)
    assert not w and abs(lam - 100.0) < 1e-6


def test_lanczos_continues_until_converged():
    """What the first 120 steps do not resolve is continued (larger basis, then restarts from the Ritz vector) instead of being
    handed back as a lower bound with a warning: the reference stores np.linalg.cond, an exact value (Kernel.py:239-245)."""
    import warnings
    from gpgradpy_amd.cond_number import lanczos_largest
    rng = np.random.default_rng(2)
    n = 3000
    # K^-1 of a preconditioned covariance matrix: most eigenvalues packed below the top, which is only slightly separated
    lam = np.concatenate((np.linspace(1.0, 0.999, 40) ** 2 * 1e10, 10.0 ** rng.uniform(0, 9.9, n - 40)))
    lam[0] = 1.0003e10
    H = rng.standard_normal((n, 3))
    H, _ = np.linalg.qr(H)

    def apply(v):                                               # diag(lam) in a rotated basis (three Householder reflections)
        w = v.copy()
        for k in range(3):
            w -= 2.0 * H[:, k] * (H[:, k] @ w)
        w = lam * w
        for k in (2, 1, 0):
            w -= 2.0 * H[:, k] * (H[:, k] @ w)
        return w
    with warnings.catch_warnings(record=True) as w120:
        warnings.simplefilter("always")
        lanczos_largest(apply, n, k_cap=120, max_restarts=0)
    assert w120, "the case must be one that 120 steps do not resolve"
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        val, vec = lanczos_largest(apply, n, want_vector=True)
    assert not w, [str(x.message) for x in w]
    assert abs(val - lam.max()) <= 1e-9 * lam.max()
    assert np.linalg.norm(apply(vec) - val * vec) <= 1e-6 * val


def test_rtensor_init_is_built_on_demand():
    """Rtensor_init (GaussianProcess.py:363) costs nothing until somebody reads it; then it is the reference's tensor
    R[k, a, b] = x[a, k] - x[b, k] (CommonFun.py:56-84), and get_scl_x_w_dist hands it out like the reference."""
    GP = _gp_host_only(3, 'SqExp', 5, 'none')
    assert GP._Rtensor_init is None
    x, R = GP.get_scl_x_w_dist()
    assert R.shape == (3, 5, 5) and GP._Rtensor_init is R
    for k in range(3):
        np.testing.assert_array_equal(R[k], x[:, k][:, None] - x[:, k][None, :])
    Y = np.arange(6.0).reshape(2, 3)
    R2 = GP.calc_Rtensor(x, Y, 1)
    assert R2.shape == (3, 5, 2) and R2[1, 4, 1] == x[4, 1] - Y[1, 1]


def test_dropped_model_is_collected_without_the_cycle_collector():
    """A GaussianProcess owns GBs of device memory: dropping the last reference must free it at once (no reference cycle through
    KernEta_chofac or the rescaling callback), and close() frees it explicitly."""
    import gc
    import weakref
    was = gc.isenabled()
    gc.disable()
    try:
        for wellcond in ('precon', 'rescale_origin'):
            GP = _gp_host_only(2, 'SqExp', 5, 'none') if wellcond == 'precon' else gpgradpy_amd.GaussianProcess(2, True, 'SqExp', wellcond)
            if wellcond != 'precon':
                x = np.random.default_rng(0).uniform(-1, 1, (5, 2))
                try:
                    GP.set_data(x, x[:, 0], np.zeros(5), x, np.zeros((5, 2)))
                except _lib.GpgError:
                    pass
                assert GP.DataScl.on_change is not None
            GP.KernEta_chofac = gpgradpy_amd.gaussian_process.DeviceChoFactor(GP)
            r = weakref.ref(GP)
            GP.close()                                   # idempotent, also without a context
            del GP
            assert r() is None
    finally:
        if was:
            gc.enable()


def test_history_export_and_load_round_trip(tmp_path):
    """export_data_surr / load_data_surr (GpParaDef.py:115-217): keys `surr_name + array name`, npz without pickle."""
    GP = _gp_host_only(2, 'SqExp', 4, 'none')
    GP.path_data_surr = str(tmp_path / 'hist')
    GP.init_optz_surr(5)
    hp = GP.make_hp_class(beta=np.array([0.5]), theta=np.array([0.1, 0.2]), varK=2.0)
    GP.store_new_para_surr(0, hp, None, 123.0, 0.1, 0.05, 0.01)
    GP.store_new_para_surr(1, GP.make_hp_class(beta=np.array([0.7]), theta=np.array([0.3, 0.4]), varK=3.0), None, 456.0)
    GP.finish_optz_surr(2)
    d = GP.export_data_surr()
    assert set(d) >= {'obj_hp_theta_all', 'obj_hp_varK_all', 'obj_Kcov_cond_all', 'obj_xvec_rescaling_all', 'obj_vmin_init_all'}
    assert os.path.isfile(str(tmp_path / 'hist.npz'))
    GP.export_data_surr()                                       # the previous file is kept as ..._old.npz
    assert os.path.isfile(str(tmp_path / 'hist_old.npz'))
    G2 = _gp_host_only(2, 'SqExp', 4, 'none')
    G2.path_data_surr = GP.path_data_surr
    G2.init_optz_surr(5)
    G2.load_data_surr()
    np.testing.assert_array_equal(G2.hp_theta_all[:2], [[0.1, 0.2], [0.3, 0.4]])
    np.testing.assert_array_equal(G2.Kcov_cond_all[:2], [123.0, 456.0])
    assert np.isnan(G2.hp_theta_all[2]).all()
    G2.set_hp_from_idx(1)
    np.testing.assert_array_equal(G2.hp_vals.theta, [0.3, 0.4])


def test_lanczos_survives_exhausted_krylov_space_and_wide_spectra():
    """Small operators exhaust the Krylov space (what is left of the next vector is rounding noise), and K^-1 of an ill-conditioned K
    gives a tridiagonal matrix LAPACK's stemr can fail on: both must end in the exact answer, not in an exception."""
    from gpgradpy_amd.cond_number import cond_from_factor, _tridiag_eigh
    rng = np.random.default_rng(1)
    for n, cond in ((3, 1e3), (6, 1e12), (12, 1e16), (40, 1e14)):
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        lam = np.logspace(0, -np.log10(cond), n)
        K, Ki = (Q * lam) @ Q.T, (Q / lam) @ Q.T
        c = cond_from_factor(lambda v: K @ v, lambda v: Ki @ v, n)
        assert np.isclose(c, cond, rtol=1e-5), (n, c, cond)
        c2, lam_min, v_max, v_min = cond_from_factor(lambda v: K @ v, lambda v: Ki @ v, n, want_vectors=True)
        assert np.isclose(c2, cond, rtol=1e-5) and np.isclose(lam_min, lam[-1], rtol=1e-5)
        assert abs(abs(v_max @ Q[:, 0]) - 1) < 1e-6 and abs(abs(v_min @ Q[:, -1]) - 1) < 1e-4
    ev, vec = _tridiag_eigh(np.array([1.0, 2.0, 3.0]), np.array([0.1, 0.2]))          # the largest pair only
    T = np.diag([1.0, 2.0, 3.0]) + np.diag([0.1, 0.2], 1) + np.diag([0.1, 0.2], -1)
    assert np.isclose(ev, np.linalg.eigvalsh(T)[-1], rtol=1e-14) and np.allclose(T @ vec, ev * vec, atol=1e-13)


def test_bench_self_launch_starts_ranks_and_propagates_their_exit_code():
    """`python3 bench.py --gpus 2` with no launcher: the parent starts two rank processes of itself before anything touches
    the GPU.  Without a GPU every rank stops with bench.py's "no GPU" code 3 -- the parent must hand exactly that back (with a
    GPU this is tests/test_gpu_dist.py::test_bench_starts_its_own_ranks)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: covered by the GPU test")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "cfg2", "--steps", "2",
                        "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 3, (r.returncode, r.stderr[-1000:])
    assert r.stderr.count("no GPU visible") == 2 and "rank(s) failed: [(0, 3), (1, 3)]" in r.stderr
