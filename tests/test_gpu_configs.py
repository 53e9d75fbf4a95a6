"""GPU (MI355X): BASELINE.json configurations pinned exactly, and the drop-in state semantics of the posterior.

cfg1 / cfg2 (SURVEY.md 8(d) inputs, reference-generated fixtures cfg1_full / cfg2_full) also run through the generic
golden test of test_gpu_parity.py; here: cfg3 at FULL size asserted against one evaluation of the CPU oracle, the
interleaving of likelihood and posterior calls the reference allows (GpEvalModel.py:17-57 keeps KernEta_chofac),
eval_model_var (GpEvalModel.py:200-317), setup_eval_model(calc_cond=True), and the failed-Cholesky objective of the
optimiser (OptzLkd.py:74-77)."""
import os

import numpy as np
import pytest
from scipy.linalg import cho_solve

from conftest import GOLDEN_DIR, load_case
import tolerances as tol

pytestmark = pytest.mark.gpu


def _design(n, d):
    """SURVEY.md 8(d) / BASELINE.md section 3 inputs (same generator as bench.py)."""
    import bench
    return bench.make_workload(n, d, "cfg3")


def test_cfg2_exact_configuration():
    """BASELINE configs[1] exactly as SURVEY.md 8(d) defines it: n = 500, d = 4, noise-free, theta = 0.5, seed 0 --
    the reference's own evaluation (probe value of SURVEY.md 8c: ln_lkd = -3.765538651174e+03)."""
    import gpgradpy_amd
    c = load_case(os.path.join(GOLDEN_DIR, "cfg2_full.npz"))
    X, f, g, _ = _design(500, 4)
    np.testing.assert_array_equal(X, c["x"])                       # bench.py's generator IS the fixture's input
    np.testing.assert_allclose(f, c["f"], rtol=1e-14)
    assert abs(c["ln_lkd"] - (-3.765538651174e+03)) < 1e-6
    GP = gpgradpy_amd.GaussianProcess(4, True, "SqExp", "precon")
    GP.set_data(X, f, np.zeros(500), g, np.zeros((500, 4)))
    for mode in ("auto", "tile128", "blocked"):
        GP.set_factor_mode(mode)
        info, ok = GP.calc_lkd_all(GP.make_hp_class(theta=0.5 * np.ones(4)))
        assert ok
        tol.check_scalars(info.hp_beta[0], info.hp_varK, info.ln_det_Kmat, info.ln_lkd, c, GP.n_data, False)
    ln = GP.calc_lkd_batch(np.log10(0.5) * np.ones((3, 4)))          # the batched path at the exact configuration
    np.testing.assert_allclose(ln, c["ln_lkd"], rtol=tol.LN_LKD_RTOL)


def test_cfg1_exact_configuration():
    """BASELINE configs[0]: gradient-free SqExp, n = 200, d = 2 (reference probe: ln_lkd = 2.992208556656e+02)."""
    import gpgradpy_amd
    c = load_case(os.path.join(GOLDEN_DIR, "cfg1_full.npz"))
    assert abs(c["ln_lkd"] - 2.992208556656e+02) < 1e-7 and c["wellcond"] == "base"
    GP = gpgradpy_amd.GaussianProcess(2, False, "SqExp", "precon")   # 'precon' is coerced to 'base' (GaussianProcess.py:202-203)
    GP.set_data(c["x"], c["f"], np.zeros(200))
    info, ok = GP.calc_lkd_all(GP.make_hp_class(theta=0.5 * np.ones(2)))
    assert ok and GP.wellcond_mtd == "base"
    tol.check_scalars(info.hp_beta[0], info.hp_varK, info.ln_det_Kmat, info.ln_lkd, c, 200, False)


@pytest.mark.timeout(900)
def test_cfg3_full_size_against_oracle():
    """BASELINE configs[2] at full size (n = 2000, d = 8, N = 18000): every scalar of the likelihood, alpha and the
    posterior at 8 points asserted against ONE evaluation of the CPU oracle (about 10 s of LAPACK on the box's host
    cores), under the tolerances of tests/tolerances.py; value + gradient against central differences."""
    import gpgradpy_amd
    from oracle import gp_oracle as orc
    n, d = 2000, 8
    X, f, g, hp_table = _design(n, d)
    theta = 10.0 ** hp_table[0]
    GP = gpgradpy_amd.GaussianProcess(d, True, "SqExp", "precon")
    GP.set_data(X, f, np.zeros(n), g, np.zeros((n, d)))
    hp = GP.make_hp_class(theta=theta)
    info, ok = GP.calc_lkd_all(hp)
    y = orc.make_data_vec(f, g)
    N = y.size
    r = orc.calc_lkd(X, y, theta, "SqExp", True, "precon", GP._etaK, np.zeros(N), False)
    assert ok and r.ok and N == 18000
    ref = dict(hp_beta=r.hp_beta, hp_varK=r.hp_varK, ln_det_Kmat=r.ln_det_Kmat, ln_lkd=r.ln_lkd)
    tol.check_scalars(info.hp_beta[0], info.hp_varK, info.ln_det_Kmat, info.ln_lkd, ref, N, False)
    # the batched launch (what bench.py times) gives the same number for this row
    ln = GP.calc_lkd_batch(hp_table[:8])
    assert abs(ln[0] - r.ln_lkd) <= tol.LN_LKD_RTOL * abs(r.ln_lkd)
    # posterior: alpha residual against the oracle's covariance matrix, mu / sig at 8 points
    hp2 = GP.optz_closed_form_hp(hp)
    GP.set_hpara("set", 0, hp_vals=hp2)
    alpha = GP.invKernEta_fdiff
    Kcov = r.factor.Kcov
    res = y.copy()
    res[:n] -= hp2.beta[0]
    rel = np.linalg.norm(Kcov @ alpha - res) / (np.linalg.norm(Kcov, "fro") * np.linalg.norm(alpha))
    assert rel <= tol.ALPHA_RESIDUAL, rel
    assert np.linalg.norm(alpha - r.alpha) <= tol.ALPHA_NORMWISE * np.linalg.norm(r.alpha)
    xq = np.random.default_rng(5).uniform(-2, 2, (8, d))
    m = orc.setup_eval_model(X, y, theta, "SqExp", True, "precon", GP._etaK, np.zeros(N), r.hp_beta, hp2.varK)
    mu_o, sig_o = orc.eval_model(m, xq)
    mu, sig, dmudx, dsigdx = GP.eval_model(xq, calc_grad=True)[:4]
    np.testing.assert_allclose(mu, mu_o, rtol=tol.MU_RTOL, atol=tol.MU_ATOL_SCALE * max(1.0, np.abs(mu_o).max()))
    np.testing.assert_allclose(sig, sig_o, rtol=tol.SIG_RTOL, atol=tol.SIG_ATOL_SCALE * np.sqrt(hp2.varK))
    _, _, dmu_o, dsig_o = orc.eval_model_grad(m, xq[:2])
    tol.check_post_grad(dmudx[:2], dsigdx[:2], dict(dmudx=dmu_o, dsigdx=dsig_o))
    # likelihood gradient at full size: adjoint on the device against central differences of the device likelihood
    info_g, ok_g = GP.calc_lkd_all(hp, calc_grad=True)
    assert ok_g and abs(info_g.ln_lkd - info.ln_lkd) <= 1e-12 * abs(info.ln_lkd)
    k = int(np.argmax(np.abs(info_g.ln_lkd_grad)))
    h = 1e-4 * theta[k]
    tp, tm = theta.copy(), theta.copy()
    tp[k] += h
    tm[k] -= h
    fd = (GP.calc_lkd_all(GP.make_hp_class(theta=tp))[0].ln_lkd - GP.calc_lkd_all(GP.make_hp_class(theta=tm))[0].ln_lkd) / (2 * h)
    assert abs(fd - info_g.ln_lkd_grad[k]) <= 1e-6 * abs(fd), (fd, info_g.ln_lkd_grad[k])   # measured 3e-8 (tools/grad_time.py)


def test_cfg5_combination_at_tile_scale_against_oracle():
    """BASELINE configs[4]'s own combination -- Matern-5/2 + known noise on f and grad f (varK a hyperparameter) + d = 16 -- at
    n = 1000 (N = 17000: the 128-tile kernel, D = 16 assembly instantiation, noisy likelihood) against ONE evaluation of the CPU
    oracle, inputs from bench.make_workload(1000, 16, "cfg5").  (The same combination at n = 120 is pinned by the reference itself:
    cfg5_combo_n120_d16.npz through the generic golden tests; N = 68000 runs by properties in test_gpu_parity.py.)
    Reference: KernelMatern5f2.py:352-450, CalcLkd.py:185-251."""
    import bench
    import gpgradpy_amd
    from oracle import gp_oracle as orc
    n, d = 1000, 16
    X, f, g, hp_table = bench.make_workload(n, d, "cfg5")
    row = hp_table[0]
    theta, varK = 10.0 ** row[:d], 10.0 ** row[d]
    std_f, std_g = np.full(n, 1e-2), np.full((n, d), 1e-1)
    GP = gpgradpy_amd.GaussianProcess(d, True, "Ma5f2", "precon")
    GP.set_data(X, f, std_f, g, std_g)
    hp = GP.make_hp_class(theta=theta, varK=varK)
    info, ok = GP.calc_lkd_all(hp)
    assert ok and GP.last_factor()[0] == "tile128" and GP.n_data == 17000
    y = orc.make_data_vec(f, g)
    N = y.size
    nv = orc.calc_noise_vec(n, d, True, std_f, std_g, None, None)
    r = orc.calc_lkd(X, y, theta, "Ma5f2", True, "precon", GP._etaK, nv, True, varK=varK)
    assert r.ok
    ref = dict(hp_beta=r.hp_beta, hp_varK=varK, ln_det_Kmat=r.ln_det_Kmat, ln_lkd=r.ln_lkd)
    tol.check_scalars(info.hp_beta[0], info.hp_varK, info.ln_det_Kmat, info.ln_lkd, ref, N, True)
    # the batched entry point (what bench.py --config cfg5 times) gives the same number for this row
    ln = GP.calc_lkd_batch(hp_table[:2])
    assert abs(ln[0] - r.ln_lkd) <= tol.LN_LKD_RTOL * abs(r.ln_lkd)
    # posterior: the model's matrix is the EVALUATION-time one (the reference builds it with varK := 1 before the noise is divided by
    # it, Kernel.py:196-197,218 -- not the likelihood's matrix when the data are noisy): alpha against the oracle's model, its
    # residual through the oracle's factor of that matrix, mu / sig at 8 points
    hp2 = GP.optz_closed_form_hp(hp)
    GP.set_hpara("set", 0, hp_vals=hp2)
    alpha = GP.invKernEta_fdiff
    res = y.copy()
    res[:n] -= hp2.beta[0]
    m = orc.setup_eval_model(X, y, theta, "Ma5f2", True, "precon", GP._etaK, nv, r.hp_beta, hp2.varK)
    Lo = np.tril(m.chofac[0]) if m.chofac[1] else np.triu(m.chofac[0]).T
    rel = np.linalg.norm(Lo @ (Lo.T @ alpha) - res) / (np.linalg.norm(Lo, "fro") ** 2 * np.linalg.norm(alpha))   # ||L L^T||_F <= ||L||_F^2
    assert rel <= tol.ALPHA_RESIDUAL, rel
    assert np.linalg.norm(alpha - m.alpha) <= tol.ALPHA_NORMWISE * np.linalg.norm(m.alpha)
    xq = np.random.default_rng(5).uniform(-2, 2, (8, d))
    mu_o, sig_o = orc.eval_model(m, xq)
    mu, sig = GP.eval_model(xq)[:2]
    np.testing.assert_allclose(mu, mu_o, rtol=tol.MU_RTOL, atol=tol.MU_ATOL_SCALE * max(1.0, np.abs(mu_o).max()))
    np.testing.assert_allclose(sig, sig_o, rtol=tol.SIG_RTOL, atol=tol.SIG_ATOL_SCALE * np.sqrt(hp2.varK))
    assert GP.factor_fallbacks() == 0


def test_likelihood_calls_do_not_disturb_the_posterior():
    """The reference keeps KernEta_chofac / invKernEta_fdiff across later calc_lkd_all calls (GpEvalModel.py:17-57; BO
    loops interleave them).  Every kind of likelihood call between setup_eval_model and eval_model must leave the
    posterior answering from ITS factor."""
    import gpgradpy_amd
    c = load_case(os.path.join(GOLDEN_DIR, "SqExp_none_n64_d8.npz"))
    GP = gpgradpy_amd.GaussianProcess(c["d"], True, "SqExp", "precon")
    GP.set_data(c["x"], c["f"], c["std_f"], c["g"], c["std_g"])
    hp = GP.optz_closed_form_hp(GP.make_hp_class(theta=c["theta"]))
    GP.set_hpara("set", 0, hp_vals=hp)
    other = GP.make_hp_class(theta=c["theta"] * 3.0)
    rows = np.log10(c["theta"])[None, :] + np.linspace(-0.5, 0.5, 9)[:, None]
    GP.calc_lkd_all(other)                                            # single evaluation
    GP.calc_lkd_all(other, calc_grad=True)                            # value + gradient
    GP.calc_lkd_batch(rows)                                           # batched launch
    GP.calc_Kern_w_chofac(None, other, materialize=True)              # 7-tuple with downloads
    mu, sig, dmudx, dsigdx = GP.eval_model(c["xq"], calc_grad=True)[:4]
    np.testing.assert_allclose(mu, c["mu"], rtol=tol.MU_RTOL, atol=tol.MU_ATOL_SCALE * max(1.0, np.abs(c["mu"]).max()))
    np.testing.assert_allclose(sig, c["sig"], rtol=tol.SIG_RTOL, atol=tol.SIG_ATOL_SCALE * np.sqrt(hp.varK))
    tol.check_post_grad(dmudx, dsigdx, c)
    # KernEta_chofac is SciPy's (c, lower) pair, downloaded on first use from the posterior's own workspace --
    # after all those likelihood calls it still is the factor of the model's matrix
    y = GP.make_data_vec(c["f"], c["g"])
    r = y.copy()
    r[:c["n"]] -= hp.beta[0]
    a = cho_solve(GP.KernEta_chofac, r)
    assert np.linalg.norm(a - GP.invKernEta_fdiff) <= 1e-6 * np.linalg.norm(a)
    L, lower = GP.KernEta_chofac
    assert lower is True and L.shape == (GP.n_data, GP.n_data) and np.allclose(np.triu(L, 1), 0.0)
    # and the likelihood side still answers for ITS hyperparameters
    info, ok = GP.calc_lkd_all(GP.make_hp_class(theta=c["theta"]))
    assert ok and abs(info.ln_lkd - c["ln_lkd"]) <= tol.LN_LKD_RTOL * abs(c["ln_lkd"])
    # a new data set invalidates the model, as in the reference (eval_model asserts on the factor)
    GP.set_data(c["x"][:10], c["f"][:10], c["std_f"][:10], c["g"][:10], c["std_g"][:10])
    with pytest.raises(AssertionError, match="Cholesky decomposition is required"):      # GpEvalModel.py:117 (set_data drops the factor)
        GP.eval_model(c["xq"])


def test_setup_eval_model_with_cond():
    """setup_eval_model(calc_cond=True) / set_hpara(..., calc_cond=True) (GpEvalModel.py:39-41, GaussianProcess.py:365,395):
    condK = 2-norm condition number of the matrix the model factorised."""
    import gpgradpy_amd
    from oracle import gp_oracle as orc
    c = load_case(os.path.join(GOLDEN_DIR, "Ma5f2_known_n17_d4.npz"))
    GP = gpgradpy_amd.GaussianProcess(c["d"], True, "Ma5f2", "precon")
    GP.set_data(c["x"], c["f"], c["std_f"], c["g"], c["std_g"])
    hp = GP.optz_closed_form_hp(GP.make_hp_class(theta=c["theta"], varK=c["varK_in"]))
    GP.set_hpara("set", 0, hp_vals=hp, calc_cond=True)
    nv = orc.calc_noise_vec(c["n"], c["d"], True, c["std_f"], c["std_g"])
    fac = orc.calc_all_K_w_chofac(c["x"], c["theta"], "Ma5f2", True, "precon", c["etaK"], nv, varK=1.0)   # b_normlz_w_varK: varK := 1
    want = np.linalg.cond(fac.Kcov / np.outer(fac.pvec, fac.pvec))     # Kcov_precon = P^-1 Kcov P^-1 (Kernel.py:236,240)
    np.testing.assert_allclose(GP.condK, want, rtol=1e-6)
    GP.cond_norm = "fro"                                             # np.linalg.cond(Kcov_precon, 'fro'), Kernel.py:240
    GP.setup_eval_model(calc_cond=True)
    Kp = fac.Kcov / np.outer(fac.pvec, fac.pvec)
    np.testing.assert_allclose(GP.condK, np.linalg.norm(Kp, "fro") * np.linalg.norm(np.linalg.inv(Kp), "fro"), rtol=1e-8)


EVAR = ["evar_SqExp_none_n20_d3", "evar_Ma5f2_known_n25_d2", "evar_RatQu_none_n30_d2_nograd"]


@pytest.mark.parametrize("name", EVAR)
def test_eval_model_var_against_reference(name):
    import gpgradpy_amd
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    d, use_grad, kernel = int(z["d"]), bool(z["use_grad"]), str(z["kernel"])
    GP = gpgradpy_amd.GaussianProcess(d, use_grad, kernel, str(z["wellcond"]) if use_grad else "base")
    if use_grad:
        GP.set_data(z["x"], z["f"], z["std_f"], z["g"], z["std_g"])
    else:
        GP.set_data(z["x"], z["f"], z["std_f"])
    hp = GP.make_hp_class(beta=z["beta"], theta=z["theta"], kernel=None if np.isnan(z["hp_kernel"]) else float(z["hp_kernel"]),
                          varK=float(z["varK"]))
    GP.set_hpara("set", 0, hp_vals=hp)
    sig2, dsig2dx, h = GP.eval_model_var(z["xq"], calc_grad=True)
    varK = float(z["varK"])
    assert h is None and dsig2dx.shape == z["dsig2dx"].shape
    np.testing.assert_allclose(sig2, z["sig2"], rtol=1e-5, atol=(tol.SIG_ATOL_SCALE ** 2) * varK + 1e-7 * varK)
    scale = np.abs(z["dsig2dx"]).max()
    np.testing.assert_allclose(dsig2dx, z["dsig2dx"], rtol=1e-5, atol=1e-6 * scale)
    s0, g0, _ = GP.eval_model_var(z["xq"][1], calc_grad=True, squeeze_nx=True)
    assert np.isclose(s0, sig2[1], rtol=1e-6, atol=1e-9 * varK) and g0.shape == (d,)   # 1 - diag(...) cancels: another solve kernel, other rounding
    s_only, none_g, _ = GP.eval_model_var(z["xq"])
    assert none_g is None and np.allclose(s_only, sig2, rtol=1e-6, atol=1e-9 * varK)
    # consistency with eval_model: sig = sqrt(sig2), dsig2dx = 2 sig dsigdx where sig > 0
    mu, sig, _, dsigdx = GP.eval_model(z["xq"], calc_grad=True)[:4]
    np.testing.assert_allclose(sig ** 2, sig2, rtol=1e-6, atol=1e-9 * varK)
    np.testing.assert_allclose(2 * sig[:, None] * dsigdx, dsig2dx, rtol=1e-6, atol=1e-9 * scale)
    with pytest.raises(Exception, match="d2sig2dx2"):
        GP.eval_model_var(z["xq"][0], calc_grad=True, calc_hess=True)


def test_failed_cholesky_objective():
    """OptzLkd.py:74-77: when the Cholesky fails the optimiser's objective is minus the condition number, its slope
    minus the condition number's gradient.  At a failure the matrix is numerically singular (cond ~ 1e21 here), so the
    reference's own numbers are rounding noise in the last digits: orders of magnitude and finiteness are pinned."""
    import gpgradpy_amd
    z = np.load(os.path.join(GOLDEN_DIR, "failobj_SqExp_n20_d2.npz"))
    n, d = z["x"].shape
    GP = gpgradpy_amd.GaussianProcess(d, True, "SqExp", "base")
    GP.set_data(z["x"], z["f"], np.zeros(n), z["g"], np.zeros((n, d)))
    GP._etaK = GP._eta_Kgrad = float(z["etaK"])
    GP._last_hp_vec = np.full((1, d), np.nan)
    val, grad, cond, cond_grad = GP.calc_store_likelihood(z["hp_vec"])
    assert not GP._last_chofac_good
    assert np.isfinite(val) and val == -cond and cond > 1e15
    assert abs(np.log10(cond) - np.log10(float(z["cond"]))) < 3.0
    assert grad.shape == (d,) and np.all(np.isfinite(grad)) and np.array_equal(grad, -cond_grad)
    # the same through 'precon' (the reference has no cond_grad there and stops with a TypeError): finite, zero slope
    GP2 = gpgradpy_amd.GaussianProcess(d, True, "SqExp", "precon")
    GP2.set_data(z["x"], z["f"], np.zeros(n), z["g"], np.zeros((n, d)))
    GP2._etaK = GP2._eta_Kgrad = -0.9                                   # makes the preconditioned matrix indefinite
    GP2._last_hp_vec = np.full((1, d), np.nan)
    v2, g2 = GP2.calc_store_likelihood(z["hp_vec"])[:2]
    assert np.isfinite(v2) and v2 < 0 and np.all(g2 == 0.0)
    # SLSQP over a start row that fails does not stall on NaN and does not raise
    from scipy.optimize import Bounds
    best, _, info = GP.optz_hp_max_lkd(np.array([z["hp_vec"]]), Bounds(z["hp_vec"] - 1.0, z["hp_vec"] + 3.0, keep_feasible=True))
    assert np.all(np.isfinite(best))


@pytest.mark.parametrize("kernel,hp_kernel", [("SqExp", np.nan), ("Ma5f2", np.nan), ("RatQu", 1.5)])
def test_kernel_table_against_finite_differences(kernel, hp_kernel):
    """The design of the reference's unit_test/test_Kfull.py:42-129 on the device kernel table: every block of
    calc_KernGrad against finite differences of calc_KernBase -- first derivatives in either argument, mixed second
    derivatives -- here with two DIFFERENT point sets (the cross-kernel case the posterior uses) as well as X1 = X2."""
    import gpgradpy_amd
    dim, n1, n2, eps = 3, 7, 5, 1e-4
    rng = np.random.default_rng(42)
    theta = np.linspace(1, 2, dim)                                    # test_Kfull.py:30
    X1 = rng.uniform(-1, 1, (n1, dim))
    for X2 in (rng.uniform(-1, 1, (n2, dim)), X1):
        m2 = X2.shape[0]
        GP = gpgradpy_amd.GaussianProcess(dim, True, kernel, "precon")
        R = GP.calc_Rtensor(X1, X2, 1)
        Kbase = GP.calc_KernBase(R, theta, hp_kernel)
        Kfull = GP.calc_KernGrad(R, theta, hp_kernel)
        assert Kbase.shape == (n1, m2) and Kfull.shape == (n1 * (dim + 1), m2 * (dim + 1))
        np.testing.assert_array_equal(Kfull[:n1, :m2], Kbase)
        fd = np.zeros_like(Kfull)
        fd[:n1, :m2] = Kbase

        def base(Xa, Xb):
            return GP.calc_KernBase(GP.calc_Rtensor(Xa, Xb, 1), theta, hp_kernel)
        for i in range(dim):
            e = np.zeros(dim)
            e[i] = eps
            fd[n1 * (i + 1):n1 * (i + 2), :m2] = (base(X1 + e, X2) - base(X1 - e, X2)) / (2 * eps)
            fd[:n1, m2 * (i + 1):m2 * (i + 2)] = (base(X1, X2 + e) - base(X1, X2 - e)) / (2 * eps)
            for j in range(dim):
                f = np.zeros(dim)
                f[j] = eps
                fd[n1 * (i + 1):n1 * (i + 2), m2 * (j + 1):m2 * (j + 2)] = \
                    (base(X1 + e, X2 + f) + base(X1 - e, X2 - f) - base(X1 + e, X2 - f) - base(X1 - e, X2 + f)) / (4 * eps ** 2)
        if kernel == "Ma5f2" and X2 is X1:
            # Matern-5/2 is only twice differentiable at R = 0: the central second difference on the diagonal of the
            # gradient-gradient blocks has an O(eps) error term of its own (|R|^3 in the kernel); compare off the diagonal
            mask = np.ones_like(Kfull, dtype=bool)
            for i in range(1, dim + 1):
                for j in range(1, dim + 1):
                    mask[n1 * i + np.arange(n1), m2 * j + np.arange(n1)] = False
            np.testing.assert_allclose(Kfull[mask], fd[mask], rtol=1e-4, atol=1e-4)      # test_Kfull.py:33-34
            np.testing.assert_allclose(Kfull[~mask], fd[~mask], rtol=1e-2, atol=1e-2)
        else:
            np.testing.assert_allclose(Kfull, fd, rtol=1e-4, atol=1e-4)                  # test_Kfull.py:33-34
    # gradient masks on either side (bvec_use_grad1 / 2): the masked matrix is the matching sub-matrix of the full one
    X2 = rng.uniform(-1, 1, (n2, dim))
    R = GP.calc_Rtensor(X1, X2, 1)
    full = GP.calc_KernGrad(R, theta, hp_kernel)
    b1 = np.array([1, 0, 1, 1, 0, 1, 0], dtype=bool)
    b2 = np.array([0, 1, 1, 0, 1], dtype=bool)
    rows = np.concatenate([np.arange(n1)] + [n1 * (i + 1) + np.flatnonzero(b1) for i in range(dim)])
    cols = np.concatenate([np.arange(n2)] + [n2 * (j + 1) + np.flatnonzero(b2) for j in range(dim)])
    if kernel != "RatQu":       # the reference's RatQu cross kernel fails with a mask on one side only (KernelRatQuad.py:497-499)
        np.testing.assert_array_equal(GP.calc_KernGrad(R, theta, hp_kernel, b1, None), full[rows, :])
        np.testing.assert_array_equal(GP.calc_KernGrad(R, theta, hp_kernel, None, b2), full[:, cols])
    np.testing.assert_array_equal(GP.calc_KernGrad(R, theta, hp_kernel, b1, b2), full[np.ix_(rows, cols)])
    # the self-kernel agrees with the fused assembly of the likelihood path (Kern of the 7-tuple)
    GP.set_data(X1, np.zeros(n1), np.zeros(n1), np.zeros((n1, dim)), np.zeros((n1, dim)))
    hp = GP.make_hp_class(theta=theta, kernel=None if np.isnan(hp_kernel) else hp_kernel)
    Kern = GP.calc_Kern_w_chofac(None, hp, materialize=True)[0]
    np.testing.assert_allclose(GP.calc_Kern(GP.calc_Rtensor(X1, X1), theta, hp_kernel), Kern, rtol=tol.KERN_RTOL, atol=tol.KERN_ATOL)
    GPb = gpgradpy_amd.GaussianProcess(dim, False, kernel, "base")
    np.testing.assert_array_equal(GPb.calc_Kern(R, theta, hp_kernel), GPb.calc_KernBase(R, theta, hp_kernel))


@pytest.mark.parametrize("kernel,noise,n,d", [("SqExp", "none", 300, 8), ("Ma5f2", "known", 140, 7), ("RatQu", "unknown", 90, 3)])
def test_lkd_grad_batch_matches_one_at_a_time(kernel, noise, n, d):
    """gpg_lkd_grad_batch (one launch each of the factorisations and of the two inverse sweeps per group of rows) against
    gpg_lkd_grad row by row: same values and gradients; a row whose Cholesky fails only fails itself."""
    import gpgradpy_amd
    from oracle import gp_oracle as orc
    X, f, g = orc.synthetic_design(n, d, seed=n)
    std_f = std_g = None
    if noise == "none":
        std_f, std_g = np.zeros(n), np.zeros((n, d))
    elif noise == "known":
        std_f, std_g = np.full(n, 1e-2), np.full((n, d), 1e-1)
    GP = gpgradpy_amd.GaussianProcess(d, True, kernel, "precon")
    GP.set_data(X, f, std_f, g, std_g)
    hi = GP.hp_info_optz_lkd
    rng = np.random.default_rng(7)
    rows = np.zeros((5, hi.n_hp))
    rows[:, hi.idx_theta] = rng.uniform(-2.0, -0.7, (5, d))
    if hi.has_kernel:
        rows[:, hi.idx_kernel] = rng.uniform(-0.3, 0.5, (5, 1))
    if hi.has_varK:
        rows[:, hi.idx_varK] = rng.uniform(-0.5, 0.5, 5)
    if hi.has_var_fval:
        rows[:, hi.idx_var_fval] = rng.uniform(-5, -3, 5)
    if hi.has_var_fgrad:
        rows[:, hi.idx_var_fgrad] = rng.uniform(-3, -1, 5)
    ln, grad, ok = GP.calc_lkd_grad_batch(rows)
    assert ok.all() and grad.shape == (5, hi.n_hp)
    for i in range(5):
        info, good = GP.calc_lkd_all(GP.hp_vec2dataclass(hi, rows[i]), calc_grad=True)
        assert good
        np.testing.assert_allclose(ln[i], info.ln_lkd, rtol=1e-12)
        np.testing.assert_allclose(grad[i], info.ln_lkd_grad, rtol=1e-9, atol=1e-9 * np.abs(info.ln_lkd_grad).max())
    # value-only batch agrees too
    np.testing.assert_allclose(GP.calc_lkd_batch(rows), ln, rtol=1e-12)
    # the optimiser's batched objective = the one-row objective (log10 chain rule included)
    vals = GP._objective_rows(rows[:3])
    for i in range(3):
        GP._last_hp_vec = None
        v, gvec = GP.calc_store_likelihood(rows[i])[:2]
        np.testing.assert_allclose(vals[i][0], v, rtol=1e-12)
        np.testing.assert_allclose(vals[i][1], gvec, rtol=1e-9, atol=1e-9 * np.abs(gvec).max())


def test_lkd_grad_batch_with_a_failing_row():
    import gpgradpy_amd
    from oracle import gp_oracle as orc
    n, d = 70, 5
    X, f, g = orc.synthetic_design(n, d, seed=21)
    GP = gpgradpy_amd.GaussianProcess(d, True, "SqExp", "base")
    GP.set_data(X, f, np.zeros(n), g, np.zeros((n, d)))
    GP._etaK = GP._eta_Kgrad = 0.0
    rows = np.random.default_rng(8).uniform(-1.0, 0.0, (6, d))
    rows[2] = -9.0                                                  # theta = 1e-9: numerically singular without nugget
    ln, grad, ok = GP.calc_lkd_grad_batch(rows)
    assert not ok[2] and np.isnan(ln[2]) and ok[[0, 1, 3, 4, 5]].all() and np.all(np.isfinite(grad[[0, 1, 3, 4, 5]]))
    vals = GP._objective_rows(rows)                                 # the failed row falls back on the condition-number objective
    assert np.isfinite(vals[2][0]) and vals[2][0] < -1e10 and np.all(np.isfinite(vals[2][1]))


def test_lockstep_multistart_equals_sequential():
    """The SLSQP starts of set_hpara('optz') in lock step (one batched value + gradient call per round) end exactly where
    the reference's one-after-the-other loop (OptzLkd.py:249-290) ends."""
    import gpgradpy_amd
    from oracle import gp_oracle as orc
    n, d = 60, 4
    X, f, g = orc.synthetic_design(n, d, seed=12)
    res = {}
    for lock in (True, False):
        GP = gpgradpy_amd.GaussianProcess(d, True, "SqExp", "precon")
        GP.lkd_optz_start_mtd = "lhs"
        GP.optz_n_x0 = 5
        GP.optz_lockstep = lock
        GP.init_optz_surr(2)
        GP.set_data(X, f, np.zeros(n), g, np.zeros((n, d)))
        GP.set_hpara("optz", 0)
        res[lock] = (GP.optz_sol_all_last.copy(), GP.optz_obj_all_last.copy(), GP.hp_vals.theta.copy())
    assert GP.optz_lockstep is False
    np.testing.assert_allclose(res[True][1], res[False][1], rtol=1e-9)
    np.testing.assert_allclose(res[True][0], res[False][0], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(res[True][2], res[False][2], rtol=1e-6)


@pytest.mark.parametrize("n,d", [(455, 8), (470, 8), (130, 3)])
def test_dataflow_inverse_against_blocked_inverse(n, d):
    """The explicit inverse behind the adjoint gradient exists three times: W = L^-T on 64 x 64 tiles (up to 4096 padded
    columns: N = 4095 here), on 128 x 128 tiles (N = 4230), and the blocked sweeps of the 'blocked' factor mode.  Independent
    kernels, same gradient."""
    import gpgradpy_amd
    from oracle import gp_oracle as orc
    X, f, g = orc.synthetic_design(n, d, seed=n)
    theta = 10.0 ** np.random.default_rng(n).uniform(-2.0, -0.7, d)
    out = {}
    for mode in ("auto", "blocked"):
        GP = gpgradpy_amd.GaussianProcess(d, True, "SqExp", "precon")
        GP.set_data(X, f, np.zeros(n), g, np.zeros((n, d)))
        GP.set_factor_mode(mode)
        info, ok = GP.calc_lkd_all(GP.make_hp_class(theta=theta), calc_grad=True)
        assert ok
        out[mode] = (info.ln_lkd, info.ln_lkd_grad)
    assert abs(out["auto"][0] - out["blocked"][0]) <= 1e-9 * abs(out["blocked"][0])
    # both contract with an explicitly formed inverse of a matrix with cond up to 1e10: agreement to cond * eps of the largest component
    np.testing.assert_allclose(out["auto"][1], out["blocked"][1], rtol=1e-5, atol=1e-5 * np.abs(out["blocked"][1]).max())


_FRO = np.load(os.path.join(GOLDEN_DIR, "cond_fro_table.npz"))


@pytest.mark.parametrize("name", [str(x) for x in _FRO["names"]])
def test_frobenius_condition_number_against_reference(name):
    """cond_norm = 'fro' (GaussianProcess.py:104): np.linalg.cond(., 'fro') of Kernel.py:239-245 / 279-285 and the gradient of
    calc_cond_fronorm_w_grad (GpHparaCon.py:209-236) from the reference (tests/golden/cond_fro_table.npz), on the device:
    norms by reductions, K^-2 and K^-3 by two full MFMA products, derivative entries recomputed in the contraction."""
    from test_gpu_parity import _gp_from_case, _hp_from_case
    c = load_case(os.path.join(GOLDEN_DIR, name + ".npz"))
    GP = _gp_from_case(c)
    GP.cond_norm = "fro"
    hp = _hp_from_case(GP, c)
    want = float(_FRO["cond"][list(_FRO["names"]).index(name)])
    info, ok = GP.calc_lkd_all(hp, calc_cond=True)
    assert ok and abs(info.ln_lkd - c["ln_lkd"]) <= tol.LN_LKD_RTOL * abs(c["ln_lkd"])
    # both sides form an explicit inverse: agreement to cond * eps
    np.testing.assert_allclose(info.cond, want, rtol=max(1e-9, 1e-15 * want))
    key = "grad_" + name
    if key in _FRO.files:
        info_g, ok_g = GP.calc_lkd_all(hp, calc_cond=True, calc_grad=True)
        assert ok_g and np.isclose(info_g.cond, info.cond, rtol=1e-12)
        slots = tol.lkd_grad_slots_to_check(c)                     # gradient-free Matern-5/2: the reference's d K / d theta is not the derivative
        g, w = info_g.cond_grad[slots], _FRO[key][slots]
        np.testing.assert_allclose(g, w, rtol=max(1e-7, 1e-14 * want), atol=max(1e-7, 1e-14 * want) * np.abs(w).max())
    elif GP.wellcond_mtd == "precon":
        with pytest.raises(AssertionError):                        # as the reference: no gradient with the preconditioner
            GP.calc_lkd_all(hp, calc_cond=True, calc_grad=True)


def test_caller_supplied_noise_vector():
    """calc_all_K_w_chofac(..., noise_vec=...) (Kernel.py:140-143, 207-208, 218): the caller's per-row noise variances are used for that
    call only; the model's own noise and a posterior that was set up before are untouched."""
    import scipy.linalg
    import gpgradpy_amd
    from oracle import gp_oracle as orc
    n, d = 14, 2
    X, f, g = orc.synthetic_design(n, d, seed=9)
    std_f, std_g = np.full(n, 1e-2), np.full((n, d), 1e-1)
    GP = gpgradpy_amd.GaussianProcess(d, True, 'SqExp', 'precon')
    GP.set_data(X, f, std_f, g, std_g)
    hp = GP.make_hp_class(theta=np.array([0.3, 0.6]), varK=1.5)
    ln0 = GP.calc_lkd_all(hp)[0].ln_lkd
    hp_m = GP.optz_closed_form_hp(GP.make_hp_class(theta=np.array([0.3, 0.6]), varK=1.5))
    GP.set_hpara('set', 0, hp_vals=hp_m)
    xq = np.array([[0.1, -0.2], [1.0, 0.5]])
    mu0, sig0 = GP.eval_model(xq)[:2]
    nv = np.linspace(1e-3, 5e-2, n * (d + 1))
    Kern, _, Kcov, chofac, _, eta, _ = GP.calc_all_K_w_chofac(None, hp, noise_vec=nv, materialize=True)
    Kref = orc.kern_grad(X, X, hp.theta, 'SqExp')
    np.testing.assert_allclose(Kern, Kref, rtol=1e-13, atol=1e-15)
    Kw = Kref + np.diag(nv / hp.varK)
    p = np.sqrt(np.diag(Kw))
    Kcov_ref = hp.varK * (Kw + eta * np.diag(p ** 2))              # P (Kcor + eta I) P
    np.testing.assert_allclose(Kcov, Kcov_ref, rtol=1e-12, atol=1e-14)
    L = np.tril(chofac[0])
    np.testing.assert_allclose(L @ L.T, Kcov_ref, rtol=1e-10, atol=1e-12)
    b = np.arange(1.0, n * (d + 1) + 1)
    np.testing.assert_allclose(scipy.linalg.cho_solve(chofac, b), np.linalg.solve(Kcov_ref, b), rtol=1e-7)
    # back to the model's own noise, posterior untouched
    assert GP.calc_lkd_all(hp)[0].ln_lkd == ln0
    mu1, sig1 = GP.eval_model(xq)[:2]
    np.testing.assert_array_equal(mu1, mu0)
    np.testing.assert_array_equal(sig1, sig0)


def test_posterior_at_many_points_one_dataflow_launch():
    """eval_model at thousands of points: the forward / backward row solves take up to 32768 tile tasks in ONE dataflow launch (4096 in
    round 1, with a 2x cliff at the limit); beyond that, groups of row tiles / the blocked sweep.  The three regimes must agree: row
    tiles are independent, so the whole call equals the call in pieces; a sample is checked against the CPU oracle."""
    import gpgradpy_amd
    from oracle import gp_oracle as orc
    n, d = 300, 4                                              # N = 1500, 24 tile columns
    X, f, g = orc.synthetic_design(n, d, seed=21)
    GP = gpgradpy_amd.GaussianProcess(d, True, 'SqExp', 'precon')
    GP.set_data(X, f, np.zeros(n), g, np.zeros((n, d)))
    hp = GP.optz_closed_form_hp(GP.make_hp_class(theta=np.full(d, 0.2)))
    GP.set_hpara('set', 0, hp_vals=hp)
    rng = np.random.default_rng(5)
    xq = rng.uniform(-2, 2, (90000, d))
    pieces = {}
    for nx in (12000, 90000):                                  # 4512 tasks (new single-launch regime), 33768 (> 32768: groups / blocked sweep)
        mu, sig, dmu, dsig = GP.eval_model(xq[:nx], calc_grad=True)[:4]
        mu_p = np.concatenate([GP.eval_model(xq[i:i + 2000])[0] for i in range(0, 12000, 2000)])
        out_p = [GP.eval_model(xq[i:i + 2000], calc_grad=True)[:4] for i in range(0, 12000, 2000)]
        np.testing.assert_allclose(mu[:12000], mu_p, rtol=1e-12, atol=1e-12 * np.abs(mu_p).max())
        for k, ref in enumerate((mu, sig, dmu, dsig)):
            got = np.concatenate([o[k] for o in out_p])
            # (the slice count of the predictive reductions depends on the call's size: sums in another order, and dsigdx divides by sigma)
            np.testing.assert_allclose(ref[:12000], got, rtol=1e-7, atol=1e-8 * max(1.0, np.abs(got).max()))
        pieces[nx] = (mu, sig)
    np.testing.assert_allclose(pieces[90000][0][:12000], pieces[12000][0], rtol=1e-12, atol=1e-12 * np.abs(pieces[12000][0]).max())
    y = orc.make_data_vec(f, g)
    r = orc.calc_lkd(X, y, hp.theta, 'SqExp', True, 'precon', GP._etaK, np.zeros(y.size), False)
    mo = orc.setup_eval_model(X, y, hp.theta, 'SqExp', True, 'precon', GP._etaK, np.zeros(y.size), np.atleast_1d(r.hp_beta), hp.varK)
    idx = rng.integers(0, 90000, 60)
    mu_o, sig_o = orc.eval_model(mo, xq[idx])
    np.testing.assert_allclose(pieces[90000][0][idx], mu_o, rtol=1e-6, atol=1e-7 * max(1.0, np.abs(mu_o).max()))
    np.testing.assert_allclose(pieces[90000][1][idx], sig_o, rtol=1e-4, atol=1e-6 * np.sqrt(hp.varK))
    assert GP.factor_fallbacks() == 0


def test_overlapped_inverse_equals_sequential(monkeypatch):
    """Value + gradient of one small matrix: W = L^-T runs on the context's second stream behind the factorisation, gated by the
    factorisation's diagonal-tile flags.  Same arithmetic in the same order as the sequential schedule: bit-identical gradients; and
    the progress argument must hold with any number of resident workgroups (gpg_set_max_workgroups)."""
    import gpgradpy_amd
    from oracle import gp_oracle as orc
    # 2, 16 and 50 tile columns (W on 64 x 64 tiles), 78 (64-tile factorisation, W on 128 x 128 tiles)
    for n, d, kernel in ((30, 3, 'SqExp'), (200, 4, 'Ma5f2'), (450, 6, 'SqExp'), (700, 6, 'SqExp')):
        X, f, g = orc.synthetic_design(n, d, seed=n)
        res = {}
        for mode in ('0', '1', '2'):                               # sequential | W behind L | -(W W^T) behind W as well (the default)
            monkeypatch.setenv('GPG_OVERLAP_INVERSE', mode)
            GP = gpgradpy_amd.GaussianProcess(d, True, kernel, 'precon')
            GP.set_data(X, f, np.full(n, 1e-3), g, np.full((n, d), 1e-2))
            hp = GP.make_hp_class(theta=np.linspace(0.05, 0.3, d), varK=1.3)
            out = []
            for cap in ((0, 1, 3, 100) if n < 700 else (0, 7, 100)):
                GP.set_max_workgroups(cap)
                info, ok = GP.calc_lkd_all(hp, calc_grad=True)
                assert ok and GP.factor_fallbacks() == 0
                out.append((info.ln_lkd, info.ln_lkd_grad.copy()))
            for ln, gr in out[1:]:
                assert ln == out[0][0] and np.array_equal(gr, out[0][1])
            res[mode] = out[0]
            GP.close()
        for mode in ('1', '2'):
            assert res['0'][0] == res[mode][0]
            np.testing.assert_array_equal(res['0'][1], res[mode][1])


def test_overlapped_inverse_does_not_read_a_previous_calls_flags():
    """The overlapped launches keep their flags in a buffer that is reused across calls; after a call its content reads "every tile of W
    is finished".  -(W W^T) of the next call runs on the main stream while W's flags are cleared on the other one: it must wait for that
    clear (an event between the streams), else it contracts the PREVIOUS call's W.  Alternating hyperparameters on one context, every
    gradient bit-identical to the sequential schedule's; no fallback of any kind."""
    import gpgradpy_amd
    from oracle import gp_oracle as orc
    for n, d in ((200, 4), (700, 6)):
        X, f, g = orc.synthetic_design(n, d, seed=3 * n)
        thetas = [np.linspace(0.05, 0.3, d), np.linspace(0.4, 0.02, d), np.full(d, 0.11)]
        ref = []
        GPs = gpgradpy_amd.GaussianProcess(d, True, 'SqExp', 'precon')
        GPs.set_data(X, f, np.zeros(n), g, np.zeros((n, d)))
        import os
        os.environ['GPG_OVERLAP_INVERSE'] = '0'
        try:
            GP0 = gpgradpy_amd.GaussianProcess(d, True, 'SqExp', 'precon')
        finally:
            del os.environ['GPG_OVERLAP_INVERSE']
        GP0.set_data(X, f, np.zeros(n), g, np.zeros((n, d)))
        for th in thetas:
            info, ok = GP0.calc_lkd_all(GP0.make_hp_class(theta=th), calc_grad=True)
            assert ok
            ref.append((info.ln_lkd, info.ln_lkd_grad.copy()))
        for rep in range(4):
            for k, th in enumerate(thetas):
                info, ok = GPs.calc_lkd_all(GPs.make_hp_class(theta=th), calc_grad=True)
                assert ok and info.ln_lkd == ref[k][0]
                np.testing.assert_array_equal(info.ln_lkd_grad, ref[k][1])
        assert GPs.factor_fallbacks() == 0 and GPs.overlap_fallbacks() == 0 and GPs.solve_fallbacks() == 0
        GPs.close(); GP0.close()


def test_alpha_of_the_last_gradient_call_is_refused_after_other_calls():
    """gpg_lkd_alpha hands out p * alpha of the last gpg_lkd_grad from workspace set 0; any call that rewrites that set's vectors in
    between (here gpg_abs_rowsum, gpg_set_data) must make it fail instead of returning alpha scaled by the wrong preconditioner."""
    import ctypes as C
    import gpgradpy_amd
    from gpgradpy_amd import _lib
    from oracle import gp_oracle as orc
    n, d = 40, 3
    X, f, g = orc.synthetic_design(n, d, seed=5)
    GP = gpgradpy_amd.GaussianProcess(d, True, 'SqExp', 'precon')
    GP.set_data(X, f, np.zeros(n), g, np.zeros((n, d)))
    hp = GP.make_hp_class(theta=np.array([0.05, 0.2, 0.01]))
    info, ok = GP.calc_lkd_all(hp, calc_grad=True)
    assert ok
    lib, ctx = GP._lib, GP._ctx
    a = np.zeros(GP.n_data)
    assert lib.gpg_lkd_alpha(ctx, _lib.as_dp(a)) == 0 and np.all(np.isfinite(a)) and np.abs(a).max() > 0
    rs = np.zeros(GP.n_data)
    ghp, keep = GP._make_hp(GP.make_hp_class(theta=np.array([0.5, 0.02, 0.3])), 1.0, True)
    assert lib.gpg_abs_rowsum(ctx, C.byref(ghp), _lib.as_dp(rs)) == 0
    b = np.zeros(GP.n_data)
    assert lib.gpg_lkd_alpha(ctx, _lib.as_dp(b)) != 0 and b"gpg_lkd_grad" in lib.gpg_last_error(ctx)
    info, ok = GP.calc_lkd_all(hp, calc_grad=True)
    assert ok and lib.gpg_lkd_alpha(ctx, _lib.as_dp(b)) == 0
    np.testing.assert_array_equal(a, b)
    GP.set_data(X, f + 1.0, np.zeros(n), g, np.zeros((n, d)))
    assert GP._lib.gpg_lkd_alpha(GP._ctx, _lib.as_dp(b)) != 0


def test_overlapped_inverse_is_cancelled_when_a_large_factorisation_fails():
    """Above 12288 columns the host reads the factorisation's info word before it goes on (a failed factor should not cost two more
    N^3/3 sweeps); W = L^-T is already running on the second stream by then, and is told to drain through its abort word.  The call must
    report the failure, and the context must work afterwards (same value as the value-only call, gradient against central differences)."""
    import time
    import gpgradpy_amd
    n, d = 1400, 8                                             # N = 12600 -> 12672 padded columns: 128-tile factorisation
    X, f, g = _design(n, d)[:3]
    GP = gpgradpy_amd.GaussianProcess(d, True, 'SqExp', 'precon')
    GP.set_data(X, f, np.zeros(n), g, np.zeros((n, d)))
    assert GP.n_data > 12288
    hp_good = GP.make_hp_class(theta=np.full(d, 0.05))
    info_g, ok = GP.calc_lkd_all(hp_good, calc_grad=True)
    assert ok and GP.last_factor()[0] == 'tile128'
    eta = GP._etaK
    GP._etaK = GP._eta_Kgrad = 0.0
    hp_bad = GP.make_hp_class(theta=np.full(d, 1e-9))          # all points alike, no nugget: not positive definite in fp64
    t0 = time.perf_counter()
    info_b, ok_b = GP.calc_lkd_all(hp_bad, calc_grad=True)
    t_bad = time.perf_counter() - t0
    assert not ok_b and info_b.ln_lkd is None
    GP._etaK = GP._eta_Kgrad = eta
    t0 = time.perf_counter()
    info_2, ok2 = GP.calc_lkd_all(hp_good, calc_grad=True)
    t_good = time.perf_counter() - t0
    assert ok2 and info_2.ln_lkd == info_g.ln_lkd and np.array_equal(info_2.ln_lkd_grad, info_g.ln_lkd_grad)
    assert GP.calc_lkd_all(hp_good)[0].ln_lkd == info_g.ln_lkd
    assert t_bad < 0.8 * t_good, (t_bad, t_good)               # the two inverse sweeps were not run to the end
    assert GP.factor_fallbacks() == 0
