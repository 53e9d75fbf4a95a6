"""Hyperparameter-derivative entries of the kernel table on the device (gpg_kern_rtensor_grad_hp) and the compositions of
optz/GpHparaGrad.py, against reference runs (tests/golden/gen_golden_kgrad.py) and, in the design of the reference's
unit_test/test_grad_Kmat.py, against finite differences of the device's own kernel matrices."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR

pytestmark = pytest.mark.gpu
CASES = sorted(p for p in glob.glob(os.path.join(GOLDEN_DIR, "kgrad_*.npz")) if "hessx" not in p)
HESSX = sorted(glob.glob(os.path.join(GOLDEN_DIR, "kgrad_hessx_*.npz")))


def _load(path):
    with np.load(path, allow_pickle=False) as z:
        return {k: (z[k].item() if z[k].ndim == 0 else z[k]) for k in z.files}


def _close(got, ref, rtol=1e-12):
    ref = np.asarray(ref, dtype=float)
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=rtol, atol=rtol * np.abs(ref).max())


@pytest.mark.parametrize("path", CASES, ids=lambda p: os.path.basename(p)[:-4])
def test_derivative_tensors_match_reference(path):
    import gpgradpy_amd
    c = _load(path)
    n, d, kernel = int(c["n"]), int(c["d"]), str(c["kernel"])
    GP = gpgradpy_amd.GaussianProcess(d, True, kernel, str(c["wellcond"]))
    sf, sg = (None, None) if c["noise"] == "unknown" else (c["std_f"], c["std_g"])
    GP.set_data(c["x"], c["f"], sf, c["g"], sg)
    nanv = lambda v: None if np.isnan(v) else float(v)           # noqa: E731
    hp = GP.make_hp_class(theta=c["theta"], kernel=nanv(c["hp_kernel"]), varK=nanv(c["varK_in"]), var_fval=nanv(c["var_fval_in"]),
                          var_fgrad=nanv(c["var_fgrad_in"]))
    Rt = GP.get_scl_x_w_dist()[1]
    _close(GP.calc_KernGrad_grad_th(Rt, c["theta"], hp.kernel), c["grad_grad_th"])
    _close(GP.calc_Kern_grad_theta(Rt, c["theta"], hp.kernel, None), c["grad_grad_th"])
    if kernel != "Ma5f2":          # the reference's gradient-free Matern derivative is not the derivative (tests/tolerances.py)
        _close(GP.calc_KernBase_grad_th(Rt, c["theta"], hp.kernel), c["base_grad_th"])
    if GP.kernel_has_hp:
        _close(GP.calc_KernBase_grad_alpha(Rt, c["theta"], hp.kernel), c["base_grad_alpha"], 1e-11)
        _close(GP.calc_KernGrad_grad_alpha(Rt, c["theta"], hp.kernel), c["grad_grad_alpha"], 1e-11)
    else:
        with pytest.raises(Exception, match="There are no kernel hyperparameters"):
            GP.calc_KernGrad_grad_alpha(Rt, c["theta"], hp.kernel)
    if "KernGrad_hp" in c:
        _close(GP.calc_KernGrad_hp(GP.hp_info_optz_lkd, hp, Rt), c["KernGrad_hp"], 1e-11)
    else:
        Kern = GP.calc_Kern(Rt, c["theta"], hp.kernel, None, None)
        _close(Kern, c["Kern"])
        _close(GP.calc_Kcov_grad_hp(GP.hp_info_optz_lkd, hp, Kern, Rt), c["Kcov_grad_hp"], 1e-11)
    with pytest.raises(NotImplementedError):
        GP.calc_KernGrad_grad_th(Rt, c["theta"], hp.kernel, np.arange(n) > 0)


@pytest.mark.parametrize("kernel", ["SqExp", "Ma5f2", "RatQu"])
@pytest.mark.parametrize("use_grad", [True, False])
def test_derivative_tensors_against_finite_differences(kernel, use_grad):
    """unit_test/test_grad_Kmat.py's design: analytic d K / d theta (and d K / d alpha) against central differences of the
    kernel matrix -- here both sides come from the device.  Covers the gradient-free Matern-5/2 derivative, which the
    reference's own value cannot pin."""
    import gpgradpy_amd
    rng = np.random.default_rng(5)
    n, d = 7, 3
    x = rng.uniform(-1.5, 1.5, (n, d))
    theta = 10.0 ** rng.uniform(-0.8, 0.3, d)
    GP = gpgradpy_amd.GaussianProcess(d, use_grad, kernel, 'precon')
    a = 1.5 if kernel == "RatQu" else None
    Rt = GP.calc_Rtensor(x, x, 1)
    an = GP.calc_Kern_grad_theta(Rt, theta, a) if use_grad else GP.calc_KernBase_grad_th(Rt, theta, a)
    eps = 1e-6
    for k in range(d):
        tp, tm = theta.copy(), theta.copy()
        tp[k] += eps
        tm[k] -= eps
        fd = (GP.calc_Kern(Rt, tp, a) - GP.calc_Kern(Rt, tm, a)) / (2 * eps)
        np.testing.assert_allclose(an[k], fd, rtol=1e-6, atol=1e-8)
    if kernel == "RatQu":
        an_a = GP.calc_Kern_grad_alpha(Rt, theta, a)
        fd = (GP.calc_Kern(Rt, theta, a + eps) - GP.calc_Kern(Rt, theta, a - eps)) / (2 * eps)
        np.testing.assert_allclose(an_a[0], fd, rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize("path", HESSX, ids=lambda p: os.path.basename(p)[:-4])
def test_x_derivative_entries_match_reference(path):
    import gpgradpy_amd
    c = _load(path)
    d = c["x1"].shape[1]
    GP = gpgradpy_amd.GaussianProcess(d, True, str(c["kernel"]), 'precon')
    a = None if np.isnan(c["hp_kernel"]) else float(c["hp_kernel"])
    Rt = GP.calc_Rtensor(c["x1"], c["x2"], 1)
    _close(GP.calc_KernBase_hess_x(Rt, c["theta"], a), c["base_hess_x"])
    _close(GP.calc_KernGrad_grad_x(Rt, c["theta"], a), c["grad_grad_x"])
    _close(GP.calc_Kern_hess_x(Rt, c["theta"], a), c["grad_grad_x"])


@pytest.mark.parametrize("kernel", ["SqExp", "Ma5f2", "RatQu"])
def test_x_derivative_entries_against_finite_differences(kernel):
    """d (d K / d x1_i) / d x1_k by central differences of the device's own gradient-enhanced cross kernel (rows of the x1-gradient
    blocks of calc_KernGrad), with a gradient mask on the second point set."""
    import gpgradpy_amd
    rng = np.random.default_rng(11)
    n1, n2, d = 4, 6, 3
    x1, x2 = rng.uniform(-1.5, 1.5, (n1, d)), rng.uniform(-1.5, 1.5, (n2, d))
    theta = 10.0 ** rng.uniform(-0.8, 0.3, d)
    a = 1.7 if kernel == "RatQu" else None
    mask2 = np.array([True, False, True, True, False, True])
    GP = gpgradpy_amd.GaussianProcess(d, True, kernel, 'precon')
    an = GP.calc_KernGrad_grad_x(GP.calc_Rtensor(x1, x2, 1), theta, a, mask2)
    assert an.shape == (d, n1 * d, n2 + int(mask2.sum()) * d)
    eps = 1e-6
    for k in range(d):
        xp, xm = x1.copy(), x1.copy()
        xp[:, k] += eps
        xm[:, k] -= eps
        Kp = GP.calc_KernGrad(GP.calc_Rtensor(xp, x2, 1), theta, a, None, mask2)
        Km = GP.calc_KernGrad(GP.calc_Rtensor(xm, x2, 1), theta, a, None, mask2)
        fd = (Kp[n1:] - Km[n1:]) / (2 * eps)                  # rows (i, a) of the x1-gradient blocks: shifting every x1 point at once is
        np.testing.assert_allclose(an[k], fd, rtol=2e-6, atol=2e-8)   # fine, entry (a, b) only depends on x1[a]
