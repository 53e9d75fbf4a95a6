"""fp64 parity tolerances for the likelihood hot path, used by CPU and GPU tests.

The covariance matrix is regularised to kappa(Kp) <~ cond_max_target = 1e10 (reference
GaussianProcess.py:101, GpWellCond.py:138), so two backward-stable evaluations of the same inputs may
differ by kappa * eps ~ 2e-6 relative in anything that depends on the smallest pivots.  Measured on
the golden set between the reference and the NumPy oracle (two CPU paths, same LAPACK):
|d ln_det| <= 2.2e-7, rel ln_lkd <= 3.5e-9, rel beta <= 1.5e-9, normwise alpha <= 2.7e-7.
"""
import numpy as np

LN_DET_ATOL = 2e-6        # + 1e-9 * N  : kappa*eps per near-singular pivot
LN_DET_ATOL_PER_N = 1e-9
LN_LKD_RTOL = 1e-8
BETA_RTOL = 1e-8
VARK_RTOL = 1e-8
ALPHA_NORMWISE = 1e-5     # kappa * eps budget
ALPHA_RESIDUAL = 1e-13    # ||Kcov alpha - r|| / (||Kcov|| ||alpha||)
MU_RTOL, MU_ATOL_SCALE = 1e-7, 1e-9
SIG_ATOL_SCALE = 1e-7     # abs tol = SIG_ATOL_SCALE * sqrt(varK)  (plus rtol 1e-5)
SIG_RTOL = 1e-5
DMU_RTOL, DSIG_RTOL = 1e-6, 1e-5      # posterior gradients (abs tol scaled by the largest entry)
KERN_RTOL, KERN_ATOL = 1e-13, 1e-15   # assembled matrix entries (exp/sqrt ulp differences only)


def check_scalars(got_beta, got_varK, got_ln_det, got_ln_lkd, c, N, noisy):
    np.testing.assert_allclose(got_beta, np.ravel(c["hp_beta"])[0], rtol=BETA_RTOL)
    np.testing.assert_allclose(got_ln_det, c["ln_det_Kmat"], rtol=0, atol=LN_DET_ATOL + LN_DET_ATOL_PER_N * N)
    np.testing.assert_allclose(got_ln_lkd, c["ln_lkd"], rtol=LN_LKD_RTOL)
    if not noisy:
        np.testing.assert_allclose(got_varK, c["hp_varK"], rtol=VARK_RTOL)


def check_post_grad(dmudx, dsigdx, c):
    np.testing.assert_allclose(dmudx, c["dmudx"], rtol=DMU_RTOL, atol=DMU_RTOL * max(1e-12, np.abs(c["dmudx"]).max()))
    np.testing.assert_allclose(dsigdx, c["dsigdx"], rtol=DSIG_RTOL, atol=DSIG_RTOL * max(1e-12, np.abs(c["dsigdx"]).max()))
