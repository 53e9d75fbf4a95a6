"""CPU: pins the oracle (oracle/gp_oracle.py) to the golden vectors captured from the reference."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, case_id, golden_case_paths, load_case
from oracle import gp_oracle as orc
import tolerances as tol

CASES = golden_case_paths()


def _inputs(c):
    y = orc.make_data_vec(c["f"], c["g"] if c["use_grad"] else None)
    vf = None if np.isnan(c["var_fval"]) else c["var_fval"]
    vg = None if np.isnan(c["var_fgrad"]) else c["var_fgrad"]
    nv = orc.calc_noise_vec(c["n"], c["d"], c["use_grad"], c["std_f"], c["std_g"] if c["use_grad"] else None, vf, vg,
                            n_grad=int(c["bvec_use_grad"].sum()))
    return y, nv


@pytest.mark.parametrize("path", CASES, ids=case_id)
def test_oracle_matches_reference(path):
    c = load_case(path)
    y, nv = _inputs(c)
    assert y.size == c["n_data"]
    # nugget (reference GpWellCond.py:116-154); the chofail case overrides it by hand
    if "chofail" not in c["name"]:
        eb, eg = orc.calc_nugget(c["n"], c["d"], c["kernel_o"], c["use_grad"], c["wellcond"])
        assert np.isclose(eg if c["use_grad"] else eb, c["etaK"], rtol=1e-14)
    noisy = c["b_has_noisy_data"]
    wc = c["wellcond"] if c["use_grad"] else "base"
    gm = None if c["bvec_use_grad"].all() else c["bvec_use_grad"]
    r = orc.calc_lkd(c["x"], y, c["theta"], c["kernel_o"], c["use_grad"], wc, c["etaK"], nv, noisy,
                     varK=c["varK_in"] if noisy else None, grad_mask=gm)
    assert r.ok == c["b_chofac_good"]
    if not r.ok:
        return
    np.testing.assert_allclose(nv, c["noise_vec"], rtol=0, atol=0)
    ln_lkd = r.ln_lkd
    if not np.isnan(c["pnlt"][0]):      # varK penalty, CalcLkd.py:118-133
        var_f = max(np.var(c["f"]), 0.1)
        ln_lkd = ln_lkd - c["pnlt"][0] * var_f * max(r.hp_varK - c["pnlt"][1] * var_f, 0.0) ** 2
        assert ln_lkd < r.ln_lkd
    tol.check_scalars(r.hp_beta[0], r.hp_varK, r.ln_det_Kmat, ln_lkd, c, y.size, noisy)
    if "pvec" in c:
        np.testing.assert_allclose(r.factor.pvec, c["pvec"], rtol=1e-15)
    np.testing.assert_allclose(np.diag(r.factor.chofac[0]), c["chofac_diag"], rtol=1e-6)
    if "Kern" in c:
        np.testing.assert_allclose(r.factor.Kern, c["Kern"], rtol=1e-14, atol=1e-16)
        np.testing.assert_allclose(r.factor.Kcov, c["Kcov"], rtol=1e-13, atol=1e-16)
    # as-written variant (dense diagonal products) gives the same values
    r2 = orc.calc_lkd(c["x"], y, c["theta"], c["kernel_o"], c["use_grad"], wc, c["etaK"], nv, noisy,
                      varK=c["varK_in"] if noisy else None, as_written=True, grad_mask=gm)
    if np.isnan(c["pnlt"][0]):
        np.testing.assert_allclose(r2.ln_lkd, c["ln_lkd"], rtol=tol.LN_LKD_RTOL)

    # adjoint gradient of the likelihood (reference CalcLkd.py:170-177 / 230-235, GpHparaGrad.py:13-155)
    if "ln_lkd_grad" in c and np.isnan(c["pnlt"][0]) and c["n_data"] <= 200:
        vf = None if np.isnan(c["var_fval"]) else c["var_fval"]
        vg = None if np.isnan(c["var_fgrad"]) else c["var_fgrad"]
        g = orc.calc_lkd_grad(c["x"], y, c["theta"], c["kernel_o"], c["use_grad"], wc, c["etaK"], c["std_f"],
                              c["std_g"] if c["use_grad"] else None, noisy, varK=c["varK_in"] if noisy else None,
                              var_fval=vf, var_fgrad=vg)
        sl = tol.lkd_grad_slots_to_check(c)
        tol.check_lkd_grad(g, c["ln_lkd_grad"], sl, cond=max(3e6, np.linalg.cond(r.factor.Kcov)))

    # posterior (reference GpEvalModel.py:17-198)
    beta = r.hp_beta
    varK_model = c["varK_in"] if noisy else r.hp_varK
    np.testing.assert_allclose(varK_model, c["varK_model"], rtol=tol.VARK_RTOL)
    m = orc.setup_eval_model(c["x"], y, c["theta"], c["kernel_o"], c["use_grad"], wc, c["etaK"], nv, beta, varK_model, gm)
    scale = np.linalg.norm(c["alpha"])
    assert np.linalg.norm(m.alpha - c["alpha"]) <= tol.ALPHA_NORMWISE * scale
    mu, sig = orc.eval_model(m, c["xq"])
    np.testing.assert_allclose(mu, c["mu"], rtol=tol.MU_RTOL, atol=tol.MU_ATOL_SCALE * max(1.0, np.abs(c["mu"]).max()))
    np.testing.assert_allclose(sig, c["sig"], rtol=tol.SIG_RTOL, atol=tol.SIG_ATOL_SCALE * np.sqrt(varK_model))
    mu2, sig2, dmudx, dsigdx = orc.eval_model_grad(m, c["xq"])
    np.testing.assert_allclose(mu2, mu, rtol=1e-12)
    tol.check_post_grad(dmudx, dsigdx, c)


def test_micro_example_survey_values():
    """Worked micro example recorded in SURVEY.md 8c."""
    c = load_case(os.path.join(GOLDEN_DIR, "micro_d1.npz"))
    assert np.isclose(c["etaK"], 2.3367333238305004e-10, rtol=1e-15)
    assert np.isclose(c["ln_lkd"], 1.325209576781126, rtol=1e-13)
    y, nv = _inputs(c)
    r = orc.calc_lkd(c["x"], y, c["theta"], "SqExp", True, "precon", c["etaK"], nv, False)
    assert np.isclose(r.hp_beta[0], 1.106530659570904, rtol=1e-10)
    assert np.isclose(r.hp_varK, 1.221743376624635, rtol=1e-9)
    assert np.isclose(r.ln_det_Kmat, -3.4515344972965383, rtol=1e-11)


@pytest.mark.parametrize("name,noisy,kernel", [("multistart_SqExp_n64_d4", False, "SqExp"),
                                               ("multistart_Ma5f2_noisy_n40_d6", True, "Ma5f2")])
def test_multistart_table(name, noisy, kernel):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    X, f, g = z["x"], z["f"], z["g"]
    y = orc.make_data_vec(f, g)
    nv = None
    if noisy:
        nv = orc.calc_noise_vec(X.shape[0], X.shape[1], True, z["std_f"], z["std_g"])
    ln, idx = orc.multistart_lkd(X, y, kernel, float(z["etaK"]), z["hp_x0"], nv, noisy)
    np.testing.assert_allclose(ln, z["ln_lkd_all"], rtol=tol.LN_LKD_RTOL)
    assert idx == int(z["idx_max"])


def test_nugget_table():
    rows = np.load(os.path.join(GOLDEN_DIR, "nugget_table.npz"))["rows"]
    for k, n, d, eb, eg in rows:
        b, gg = orc.calc_nugget(int(n), int(d), "SqExp" if k == 0 else "Ma5f2", True, "precon")
        assert np.isclose(b, eb, rtol=1e-15) and np.isclose(gg, eg, rtol=1e-14)
