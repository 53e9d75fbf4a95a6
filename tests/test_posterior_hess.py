"""Posterior Hessians d2mudx2, d2sigdx2 (reference GpEvalModel.py:355-380) against vectors
captured from the reference (tests/golden/gen_golden_hess.py): the NumPy oracle on the CPU, the HIP path on the GPU
(one query point per call, as in the reference), plus a central-difference check of the device Hessians against the
device gradients (the design of the reference's unit_test/test_grad_surr.py:184-244)."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR

HESS_CASES = sorted(glob.glob(os.path.join(GOLDEN_DIR, "hess_*.npz")))
ids = lambda p: os.path.basename(p)[:-4]  # noqa: E731


def _load(path):
    with np.load(path, allow_pickle=False) as z:
        return {k: (z[k].item() if z[k].ndim == 0 else z[k]) for k in z.files}


def _check(c, i, got, scale_mu=1e-7, scale_sig=1e-6):
    mu, sig, dmu, dsig, d2mu, d2sig = got
    assert np.isclose(mu, c["mu"][i], rtol=1e-7, atol=1e-9 * max(1.0, abs(c["mu"][i])))
    assert np.isclose(sig, c["sig"][i], rtol=1e-5, atol=1e-7 * np.sqrt(c["varK"]))
    np.testing.assert_allclose(d2mu, c["d2mudx2"][i], rtol=scale_mu * 10, atol=scale_mu * np.abs(c["d2mudx2"][i]).max())
    # d2sig divides by sig (GpEvalModel.py:378): its error scales with kappa * eps / sig
    ref = c["d2sigdx2"][i]
    np.testing.assert_allclose(d2sig, ref, rtol=1e-4, atol=scale_sig * max(1.0, np.abs(ref).max()) / min(1.0, c["sig"][i] / np.sqrt(c["varK"]) + 1e-12))


@pytest.mark.parametrize("path", HESS_CASES, ids=ids)
def test_oracle_hessians_match_reference(path):
    from oracle import gp_oracle as orc
    c = _load(path)
    n, d, ug = int(c["n"]), int(c["d"]), bool(c["use_grad"])
    y = orc.make_data_vec(c["f"], c["g"]) if ug else c["f"]
    nv = orc.calc_noise_vec(n, d, ug, c["std_f"], c["std_g"] if ug else None)
    kern = (str(c["kernel"]), float(c["hp_kernel"])) if "hp_kernel" in c else str(c["kernel"])
    m = orc.setup_eval_model(c["x"], y, c["theta"], kern, ug, "precon" if ug else "base", c["etaK"], nv,
                             c["beta"], c["varK"])
    for i in range(c["xq"].shape[0]):
        mu, sig, dmu, dsig, d2mu, d2sig = orc.eval_model_hess(m, c["xq"][i])
        _check(c, i, (mu[0], sig[0], dmu[0], dsig[0], d2mu[0], d2sig[0]), scale_mu=1e-9, scale_sig=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("path", HESS_CASES, ids=ids)
def test_device_hessians_match_reference(path):
    import gpgradpy_amd
    c = _load(path)
    d, ug = int(c["d"]), bool(c["use_grad"])
    GP = gpgradpy_amd.GaussianProcess(d, ug, str(c["kernel"]), "precon")
    if ug:
        GP.set_data(c["x"], c["f"], c["std_f"], c["g"], c["std_g"])
    else:
        GP.set_data(c["x"], c["f"], c["std_f"])
    hp = GP.make_hp_class(theta=c["theta"], kernel=float(c["hp_kernel"]) if "hp_kernel" in c else None,
                          varK=None if np.isnan(c["varK_in"]) else c["varK_in"])
    hp = GP.optz_closed_form_hp(hp)
    GP.set_hpara('set', 0, hp_vals=hp)
    for i in range(c["xq"].shape[0]):
        got = GP.eval_model(c["xq"][i], calc_grad=True, calc_hess=True, squeeze_nx=True)
        assert got[4].shape == (d, d) and got[5].shape == (d, d)
        np.testing.assert_allclose(got[4], got[4].T, rtol=1e-9, atol=1e-9 * np.abs(got[4]).max())
        _check(c, i, got)
    # central differences of the device gradients reproduce the device Hessians
    x0 = c["xq"][0]
    H_mu, H_sig = GP.eval_model(x0, calc_grad=True, calc_hess=True, squeeze_nx=True)[4:6]
    eps = 1e-5
    for k in range(d):
        xp, xm = x0.copy(), x0.copy()
        xp[k] += eps
        xm[k] -= eps
        gp_, gm_ = GP.eval_model(xp[None, :], calc_grad=True), GP.eval_model(xm[None, :], calc_grad=True)
        fd_mu = (gp_[2][0] - gm_[2][0]) / (2 * eps)
        fd_sig = (gp_[3][0] - gm_[3][0]) / (2 * eps)
        np.testing.assert_allclose(H_mu[k], fd_mu, rtol=1e-4, atol=1e-5 * np.abs(H_mu).max())
        np.testing.assert_allclose(H_sig[k], fd_sig, rtol=1e-3, atol=1e-4 * max(1.0, np.abs(H_sig).max()))
    with pytest.raises(AssertionError):
        GP.eval_model(c["xq"][:2], calc_grad=True, calc_hess=True)          # one point per call (GpEvalModel.py:358)
