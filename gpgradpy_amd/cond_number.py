"""2-norm condition number of the matrix whose Cholesky factor sits in HBM.

The reference computes ``np.linalg.cond(Kcov_precon, 2)`` / ``np.linalg.cond(Kcov, 2)`` (Kernel.py:239-245,
279-285): an SVD of the N x N matrix on the host.  Here the matrix never leaves the device: for a symmetric
positive definite matrix cond_2 = lambda_max / lambda_min, and both ends come from a Lanczos iteration (full
reorthogonalisation, the small tridiagonal eigenproblem by LAPACK on the host) on two operators the C ABI offers
through the factor: v -> (L L^T) v and v -> (L L^T)^-1 v (``gpg_factor_apply``).  Each step moves one vector of
N doubles over PCIe.
"""
import numpy as np
from scipy.linalg import eigh_tridiagonal


def lanczos_largest(apply, n, k_max=120, rtol=1e-9, seed=0):
    """Largest eigenvalue of the symmetric positive definite operator `apply` (n -> n)."""
    k_max = min(k_max, n)
    Q = np.zeros((k_max + 1, n))
    q = np.random.default_rng(seed).standard_normal(n)
    Q[0] = q / np.linalg.norm(q)
    alpha, beta = [], []
    theta = np.nan
    for j in range(k_max):
        w = apply(Q[j])
        a = float(Q[j] @ w)
        alpha.append(a)
        w = w - a * Q[j] - (beta[-1] * Q[j - 1] if j > 0 else 0.0)
        for _ in range(2):                                   # full reorthogonalisation, twice
            w -= Q[:j + 1].T @ (Q[:j + 1] @ w)
        b = float(np.linalg.norm(w))
        if j >= 1:
            ev, evec = eigh_tridiagonal(np.array(alpha), np.array(beta))
            theta = ev[-1]
            if b * abs(evec[-1, -1]) <= rtol * abs(theta):   # residual bound of the largest Ritz pair
                return theta
        else:
            theta = a
        if b <= 1e-300 or j + 1 >= k_max:
            return theta
        beta.append(b)
        Q[j + 1] = w / b
    return theta


def cond_from_factor(apply_mat, apply_inv, n):
    """cond_2 = lambda_max(K) * lambda_max(K^-1), each from its own Lanczos run."""
    if n == 1:
        return 1.0
    lam_max = lanczos_largest(apply_mat, n)
    inv_lam_min = lanczos_largest(apply_inv, n, seed=1)
    return float(lam_max * inv_lam_min)
