"""The data-rescaling well-conditioning methods ('rescale_origin', 'rescale_eta_vary', 'dflt_vmin', 'dflt_vmax') against
vectors captured from the reference (tests/golden/gen_golden_rescale.py): shift / scale of the parameter space and of the
objective, nuggets and minimum distances (host logic, CPU); likelihood + adjoint gradient + condition number with its
gradient, posterior with derivatives back in the caller's coordinates, the row-sum nugget of 'rescale_eta_vary' (device,
through the C ABI); and the oracle on the scaled data against the same vectors."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
import tolerances as tol

CASES = sorted(glob.glob(os.path.join(GOLDEN_DIR, "rescale_*.npz")))
IDS = [os.path.basename(p)[:-4] for p in CASES]


def _load(path):
    with np.load(path, allow_pickle=False) as z:
        return {k: (z[k].item() if z[k].ndim == 0 else z[k]) for k in z.files}


def _noise_args(c):
    if c["noise"] == "unknown":
        return None, None
    return c["std_f"], c["std_g"]


def _gp(c, need_device):
    import gpgradpy_amd
    from gpgradpy_amd import _lib
    GP = gpgradpy_amd.GaussianProcess(int(c["d"]), True, str(c["kernel"]), str(c["wellcond"]))
    sf, sg = _noise_args(c)
    try:
        GP.set_data(c["x"], c["f"], sf, c["g"], sg)
        if not np.isnan(c["aniso"][0]):
            GP.DataScl.set_xscale_data(xvec_scale_in=c["aniso"])
    except _lib.GpgError:
        if need_device:
            raise
        if not np.isnan(c["aniso"][0]):                    # CPU box: the scaling itself is host logic
            GP.DataScl.on_change = None
            GP.DataScl.set_xscale_data(xvec_scale_in=c["aniso"])
    return GP


def test_fixtures_present():
    assert len(CASES) >= 8


@pytest.mark.parametrize("path", CASES, ids=IDS)
def test_scaling_and_nuggets_match_reference(path):
    c = _load(path)
    GP = _gp(c, need_device=False)
    S = GP.DataScl
    assert GP.b_use_data_scl and GP.cond_eta_is_const == bool(c["cond_eta_is_const"]) and GP.b_use_cond_cstr == bool(c["b_use_cond_cstr"])
    np.testing.assert_allclose(S.x_shift, c["x_shift"], rtol=0, atol=0)
    np.testing.assert_allclose(S.xvec_scale, c["xvec_scale"], rtol=1e-13)
    np.testing.assert_allclose(S.obj_shift, c["obj_shift"], rtol=1e-15)
    np.testing.assert_allclose(S.obj_scale, c["obj_scale"], rtol=1e-13)
    np.testing.assert_allclose(S.x_scl, c["x_scl"], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(S.obj_scl, c["obj_scl"], rtol=1e-13, atol=1e-12)
    np.testing.assert_allclose(S.grad_scl, c["grad_scl"], rtol=1e-13, atol=1e-12)
    np.testing.assert_allclose(GP._etaK, c["etaK"], rtol=1e-14)
    np.testing.assert_allclose(GP._eta_Kbase, c["eta_Kbase"], rtol=1e-15)
    np.testing.assert_allclose(GP._vmin_init, c["vmin_init"], rtol=1e-14)
    if np.isnan(c["vmin_req_grad"]):
        assert np.isnan(GP._vmin_req_grad)
    else:
        np.testing.assert_allclose(GP._vmin_req_grad, c["vmin_req_grad"], rtol=1e-15)
    # scaled points honour the method's distance target (vmin / vmax of the SCALED set)
    from gpgradpy_amd.rescaling import calc_dist_max, calc_dist_min
    if str(c["wellcond"]) == "dflt_vmax":
        assert np.isclose(calc_dist_max(S.x_scl), GP.cond_dist_max_dflt, rtol=1e-12)
    else:
        target = {"rescale_origin": c["vmin_req_grad"], "rescale_eta_vary": GP.vmin_rescale_eta_vary, "dflt_vmin": GP.cond_dist_min_dflt}[str(c["wellcond"])]
        assert np.isclose(calc_dist_min(S.x_scl), target, rtol=1e-12)
    # the proposal the rescale loop makes from an optimised theta (GpWellCond.py:43-76)
    th, dist2, scale = GP.rescaling_data_w_theta_sol(S.x_scl, S.xvec_scale, np.log10(c["theta"]))
    np.testing.assert_allclose(th, c["prop_theta"], rtol=1e-12)
    np.testing.assert_allclose(dist2, c["prop_dist2"], rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(scale, c["prop_scale"], rtol=1e-12)
    # round trips
    np.testing.assert_allclose(GP.x_scl_2_init(GP.x_init_2_scl(c["xq"])), c["xq"], rtol=1e-13, atol=1e-13)
    back = GP.data_scl_2_init(*GP.data_init_2_scl(c["f"], None, c["g"]))
    np.testing.assert_allclose(back[0], c["f"], rtol=1e-12, atol=1e-10)
    np.testing.assert_allclose(back[2], c["g"], rtol=1e-12, atol=1e-10)


def _oracle_eval(c):
    """The oracle on the SCALED data with the nugget the reference used."""
    from oracle import gp_oracle as orc
    n, d = int(c["n"]), int(c["d"])
    kern = (str(c["kernel"]), float(c["hp_kernel"])) if str(c["kernel"]) == "RatQu" else str(c["kernel"])
    sc = c["obj_scale"]
    y = orc.make_data_vec(c["obj_scl"], c["grad_scl"])
    if c["noise"] == "known":
        std_f, std_g = c["std_f"] * sc, c["std_g"] * sc / c["xvec_scale"][None, :]
        nv = orc.calc_noise_vec(n, d, True, std_f, std_g)
    elif c["noise"] == "unknown":
        nv = orc.calc_noise_vec(n, d, True, None, None, c["var_fval_in"], c["var_fgrad_in"])
    else:
        nv = np.zeros(n * (d + 1))
    noisy = c["noise"] != "none"
    return orc, kern, y, nv, noisy


@pytest.mark.parametrize("path", CASES, ids=IDS)
def test_oracle_on_scaled_data_matches_reference(path):
    c = _load(path)
    orc, kern, y, nv, noisy = _oracle_eval(c)
    r = orc.calc_lkd(c["x_scl"], y, c["theta"], kern, True, "base", float(c["eta_used"]), nv, noisy,
                     varK=float(c["varK_in"]) if noisy else None)
    assert r.ok
    np.testing.assert_allclose(r.ln_lkd, c["ln_lkd"], rtol=tol.LN_LKD_RTOL)
    np.testing.assert_allclose(r.hp_beta, np.ravel(c["beta"])[0], rtol=tol.BETA_RTOL)
    if str(c["wellcond"]) == "rescale_eta_vary":                   # the row-sum nugget (Kernel.py:272-274) from the oracle's own matrix
        K = orc.kern_grad(c["x_scl"], c["x_scl"], c["theta"], kern)
        rs = np.sum(np.abs(K), axis=1)
        assert int(np.argmax(rs)) == int(c["idx_eta"])
        np.testing.assert_allclose(rs.max() / (1e10 - 1), c["eta_used"], rtol=1e-12)
    m = orc.setup_eval_model(c["x_scl"], y, c["theta"], kern, True, "base", float(c["eta_used"]), nv, np.atleast_1d(c["beta"]),
                             float(c["varK_model"]))
    xq_scl = (c["xq"] - c["x_shift"][None, :]) * c["xvec_scale"][None, :]
    mu_s, sig_s = orc.eval_model(m, xq_scl)
    mu, sig = mu_s / c["obj_scale"] + c["obj_shift"], sig_s / c["obj_scale"]
    np.testing.assert_allclose(mu, c["mu"], rtol=tol.MU_RTOL, atol=tol.MU_ATOL_SCALE * max(1.0, np.abs(c["mu"]).max()))
    np.testing.assert_allclose(sig, c["sig"], rtol=tol.SIG_RTOL, atol=tol.SIG_ATOL_SCALE * np.sqrt(c["varK_model"]) / c["obj_scale"])


def _hp(GP, c):
    return GP.make_hp_class(theta=c["theta"], kernel=None if np.isnan(c["hp_kernel"]) else c["hp_kernel"],
                            varK=None if np.isnan(c["varK_in"]) else float(c["varK_in"]),
                            var_fval=None if np.isnan(c["var_fval_in"]) else float(c["var_fval_in"]),
                            var_fgrad=None if np.isnan(c["var_fgrad_in"]) else float(c["var_fgrad_in"]))


@pytest.mark.gpu
@pytest.mark.parametrize("path", CASES, ids=IDS)
def test_device_likelihood_and_posterior_match_reference(path):
    c = _load(path)
    GP = _gp(c, need_device=True)
    hp = _hp(GP, c)
    info, good = GP.calc_lkd_all(hp, calc_cond=True, calc_grad=True)
    assert good
    np.testing.assert_allclose(GP._etaK_last, c["eta_used"], rtol=1e-12)
    if int(c["idx_eta"]) >= 0:
        assert GP._idx_etaK_argmax_last == int(c["idx_eta"])
    N = GP.n_data
    np.testing.assert_allclose(info.ln_lkd, c["ln_lkd"], rtol=tol.LN_LKD_RTOL)
    np.testing.assert_allclose(info.ln_det_Kmat, c["ln_det"], rtol=0, atol=tol.LN_DET_ATOL + tol.LN_DET_ATOL_PER_N * N)
    np.testing.assert_allclose(info.hp_beta[0], np.ravel(c["beta"])[0], rtol=tol.BETA_RTOL)
    if c["noise"] == "none":
        np.testing.assert_allclose(info.hp_varK, c["varK"], rtol=tol.VARK_RTOL)
    g_ref = c["ln_lkd_grad"]
    np.testing.assert_allclose(info.ln_lkd_grad, g_ref, rtol=tol.LKD_GRAD_RTOL, atol=tol.LKD_GRAD_RTOL * np.abs(g_ref).max())
    np.testing.assert_allclose(info.cond, c["cond"], rtol=1e-6)
    cg = c["cond_grad"]
    np.testing.assert_allclose(info.cond_grad, cg, rtol=1e-4, atol=1e-5 * np.abs(cg).max())
    # the batched entry points take the same nugget row by row
    hi = GP.hp_info_optz_lkd
    row = np.zeros((2, hi.n_hp))
    row[:, hi.idx_theta] = np.log10(c["theta"])
    if hi.has_kernel:
        row[:, hi.idx_kernel] = np.log10(c["hp_kernel"])
    if hi.has_varK:
        row[:, hi.idx_varK] = np.log10(c["varK_in"])
    if hi.has_var_fval:
        row[:, hi.idx_var_fval] = np.log10(c["var_fval_in"])
    if hi.has_var_fgrad:
        row[:, hi.idx_var_fgrad] = np.log10(c["var_fgrad_in"])
    row[1, hi.idx_theta] += 0.05
    ln_b = GP.calc_lkd_batch(row)
    ln_g, grad_b, ok = GP.calc_lkd_grad_batch(row)
    assert ok.all()
    np.testing.assert_allclose(ln_b[0], c["ln_lkd"], rtol=tol.LN_LKD_RTOL)
    np.testing.assert_allclose(ln_g, ln_b, rtol=1e-11)
    np.testing.assert_allclose(grad_b[0], g_ref, rtol=tol.LKD_GRAD_RTOL, atol=tol.LKD_GRAD_RTOL * np.abs(g_ref).max())
    one = GP.calc_lkd_all(GP.hp_vec2dataclass(hi, row[1]))[0].ln_lkd
    np.testing.assert_allclose(ln_b[1], one, rtol=1e-11)
    # posterior, back in the caller's coordinates
    hp2 = GP.optz_closed_form_hp(hp)
    np.testing.assert_allclose(hp2.varK, c["varK_model"], rtol=tol.VARK_RTOL)
    GP.set_hpara('set', 0, hp_vals=hp2)
    mu, sig, dmu, dsig = GP.eval_model(c["xq"], calc_grad=True)[:4]
    sc = c["obj_scale"]
    np.testing.assert_allclose(mu, c["mu"], rtol=tol.MU_RTOL, atol=tol.MU_ATOL_SCALE * max(1.0, np.abs(c["mu"]).max()))
    np.testing.assert_allclose(sig, c["sig"], rtol=tol.SIG_RTOL, atol=tol.SIG_ATOL_SCALE * np.sqrt(c["varK_model"]) / sc)
    tol.check_post_grad(dmu, dsig, c)
    h = GP.eval_model(c["xq"][1], calc_grad=True, calc_hess=True, squeeze_nx=True)
    np.testing.assert_allclose(h[0], c["h_mu"], rtol=tol.MU_RTOL, atol=tol.MU_ATOL_SCALE * max(1.0, abs(c["h_mu"])))
    np.testing.assert_allclose(h[4], c["h_d2mudx2"], rtol=1e-6, atol=1e-6 * np.abs(c["h_d2mudx2"]).max())
    np.testing.assert_allclose(h[5], c["h_d2sigdx2"], rtol=1e-4, atol=1e-4 * np.abs(c["h_d2sigdx2"]).max())
    with pytest.raises(Exception, match="not setup for cases where data must be rescaled"):
        GP.eval_model_var(c["xq"])                                              # GpEvalModel.py:253-256
    assert GP.factor_fallbacks() == 0
