"""The remaining outputs of the reference's likelihood-gradient code against reference runs (tests/golden/gen_golden_direct.py):
the direct (forward) method of calc_lkd_all (CalcLkd.py:69-86, 237-242) with hp_beta_grad / hp_varK_grad / ln_det_Kmat_grad, the
hp_beta_grad the noisy adjoint path fills as well, and calc_Kern_precon in the design of the reference's
unit_test/test_precon_grad.py (analytic d pvec / d theta against finite differences)."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
import tolerances as tol

CASES = sorted(glob.glob(os.path.join(GOLDEN_DIR, "direct_*.npz")))


def _load(path):
    with np.load(path, allow_pickle=False) as z:
        return {k: (z[k].item() if z[k].ndim == 0 else z[k]) for k in z.files}


@pytest.mark.parametrize("kernel", ["SqExp", "Ma5f2", "RatQu"])
def test_kern_precon_matches_reference_and_finite_differences(kernel):
    import gpgradpy_amd
    z = _load(os.path.join(GOLDEN_DIR, "precon_table.npz"))
    theta, n_eval, n_grad = z["theta"], int(z["n_eval"]), int(z["n_grad"])
    GP = gpgradpy_amd.GaussianProcess(3, True, kernel, 'precon')
    pvec, pinv, gvec = GP.calc_Kern_precon(n_eval, n_grad, theta, calc_grad=True, b_return_vec=True)
    P, Pinv, gmat = GP.calc_Kern_precon(n_eval, n_grad, theta, calc_grad=True, b_return_vec=False)
    for got, key in ((pvec, "pvec"), (pinv, "pinv"), (gvec, "gvec"), (P, "P"), (Pinv, "Pinv"), (gmat, "gmat")):
        np.testing.assert_allclose(got, z[f"{kernel}_{key}"], rtol=1e-14, atol=0)
    # unit_test/test_precon_grad.py:39-66: forward differences in theta, eps 1e-6, tolerance 1e-4 there
    eps = 1e-6
    fd = np.zeros_like(gvec)
    for i in range(3):
        th = theta.copy()
        th[i] += eps
        fd[:, i] = (GP.calc_Kern_precon(n_eval, n_grad, th, b_return_vec=True)[0] - pvec) / eps
    np.testing.assert_allclose(gvec, fd, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(np.diagonal(gmat, axis1=1, axis2=2).T, gvec, rtol=0, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("path", CASES, ids=lambda p: os.path.basename(p)[:-4])
def test_direct_method_matches_reference(path):
    import gpgradpy_amd
    c = _load(path)
    n, d = int(c["n"]), int(c["d"])
    GP = gpgradpy_amd.GaussianProcess(d, True, str(c["kernel"]), str(c["wellcond"]))
    sf, sg = (None, None) if c["noise"] == "unknown" else (c["std_f"], c["std_g"])
    GP.set_data(c["x"], c["f"], sf, c["g"], sg)
    nanv = lambda v: None if np.isnan(v) else float(v)           # noqa: E731
    hp = GP.make_hp_class(theta=c["theta"], kernel=nanv(c["hp_kernel"]), varK=nanv(c["varK_in"]), var_fval=nanv(c["var_fval_in"]),
                          var_fgrad=nanv(c["var_fgrad_in"]))
    noisy = GP.b_has_noisy_data
    adj = GP.calc_lkd_all(hp, calc_grad=True, lkd_use_adj_mtd=True)[0]
    dr = GP.calc_lkd_all(hp, calc_grad=True, lkd_use_adj_mtd=False)[0]

    def close(got, ref, rtol):
        ref = np.asarray(ref, dtype=float)
        np.testing.assert_allclose(got, ref, rtol=rtol, atol=rtol * np.abs(ref).max())

    np.testing.assert_allclose(adj.ln_lkd, c["adj_ln_lkd"], rtol=tol.LN_LKD_RTOL)
    np.testing.assert_allclose(dr.ln_lkd, c["dir_ln_lkd"], rtol=tol.LN_LKD_RTOL)
    close(adj.ln_lkd_grad, c["adj_ln_lkd_grad"], tol.LKD_GRAD_RTOL)
    close(dr.ln_lkd_grad, c["dir_ln_lkd_grad"], tol.LKD_GRAD_RTOL)
    close(dr.ln_det_Kmat_grad, c["dir_ln_det_grad"], tol.LKD_GRAD_RTOL)
    close(dr.hp_beta_grad, c["dir_hp_beta_grad"], 1e-5)
    assert dr.hp_beta_grad.shape == (1, GP.hp_info_optz_lkd.n_hp)
    if noisy:
        assert dr.hp_varK_grad is None and adj.ln_det_Kmat_grad is None
        close(adj.hp_beta_grad, c["adj_hp_beta_grad"], 1e-5)       # CalcLkd.py:216-217: filled by the adjoint path too
    else:
        assert adj.hp_beta_grad is None and adj.hp_varK_grad is None and adj.ln_det_Kmat_grad is None      # CalcLkd.py:59-63
        close(dr.hp_varK_grad, c["dir_hp_varK_grad"], tol.LKD_GRAD_RTOL)
        np.testing.assert_allclose(dr.hp_varK, c["dir_hp_varK"], rtol=tol.VARK_RTOL)
    # the optimiser's objective does not pay for hp_beta_grad
    GP._last_hp_vec = None
    hi = GP.hp_info_optz_lkd
    vec = np.zeros(hi.n_hp)
    vec[hi.idx_theta] = np.log10(c["theta"])
    if hi.has_kernel:
        vec[hi.idx_kernel] = np.log10(c["hp_kernel"])
    if hi.has_varK:
        vec[hi.idx_varK] = np.log10(c["varK_in"])
    if hi.has_var_fval:
        vec[hi.idx_var_fval] = np.log10(c["var_fval_in"])
    if hi.has_var_fgrad:
        vec[hi.idx_var_fgrad] = np.log10(c["var_fgrad_in"])
    val = GP.calc_store_likelihood(vec)[0]
    np.testing.assert_allclose(val, c["adj_ln_lkd"], rtol=tol.LN_LKD_RTOL)
    assert GP._skip_beta_grad is False
