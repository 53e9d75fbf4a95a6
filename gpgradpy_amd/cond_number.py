"""2-norm condition number of the matrix whose Cholesky factor sits in HBM.

The reference computes ``np.linalg.cond(Kcov_precon, 2)`` / ``np.linalg.cond(Kcov, 2)`` (Kernel.py:239-245,
279-285): an SVD of the N x N matrix on the host.  Here the matrix never leaves the device: for a symmetric
positive definite matrix cond_2 = lambda_max / lambda_min, and both ends come from a Lanczos iteration (full
reorthogonalisation, the small tridiagonal eigenproblem by LAPACK on the host) on two operators the C ABI offers
through the factor: v -> (L L^T) v and v -> (L L^T)^-1 v (``gpg_factor_apply``).  Each step moves one vector of
N doubles over PCIe.
"""
import warnings

import numpy as np
from scipy.linalg import eigh_tridiagonal


class LanczosNotConverged(RuntimeWarning):
    """The Ritz value returned by lanczos_largest did not meet its residual bound within k_max steps."""


def _tridiag_eigh(alpha, beta):
    """Largest eigenpair (value, unit vector) of the Lanczos tridiagonal matrix -- only that one is computed (bisection + inverse
    iteration: O(k) instead of the O(k^2) of all pairs, which at every step was most of a small problem's condition number).  LAPACK
    occasionally gives up on matrices with a very wide spectrum -- the operator here may be K^-1 with cond(K) ~ 1e10 and beyond -- ; the
    QR iteration (stev, all pairs) does not."""
    m = len(alpha)
    try:
        ev, evec = eigh_tridiagonal(alpha, beta, select='i', select_range=(m - 1, m - 1))
        if np.all(np.isfinite(evec)) and np.isfinite(ev[0]):
            return ev[0], evec[:, 0]
    except (np.linalg.LinAlgError, ValueError):
        pass
    ev, evec = eigh_tridiagonal(alpha, beta, lapack_driver='stev')
    return ev[-1], evec[:, -1]


def lanczos_largest(apply, n, k_max=120, rtol=1e-9, seed=0, want_vector=False, k_cap=None, max_restarts=12):
    """Largest eigenvalue (and, on request, its unit eigenvector) of the symmetric positive definite operator
    `apply` (n -> n).

    The Krylov basis starts with room for `k_max` vectors and GROWS (doubling, up to `k_cap`, default 8 k_max) while the
    largest Ritz pair has not met its residual bound; when the cap is reached the iteration RESTARTS from the Ritz vector
    (the Rayleigh quotient of the new start vector is the Ritz value reached so far, so the estimate never falls back),
    up to `max_restarts` times.  Only if all of that is exhausted the value is returned with a LanczosNotConverged warning
    (it is a lower bound then) -- the reference stores np.linalg.cond, an exact value, in this place (Kernel.py:239-245)."""
    k_max = max(1, min(k_max, n))
    k_cap = n if k_cap is None and 8 * k_max >= n else min(n, k_cap or 8 * k_max)
    k_cap = max(k_cap, k_max)
    q0 = np.random.default_rng(seed).standard_normal(n)
    theta, ritz, Q, res = np.nan, None, None, np.inf

    def done(theta, ritz, Q):
        if not want_vector:
            return theta
        v = Q[0].copy() if ritz is None else Q[:len(ritz)].T @ ritz
        return theta, v / np.linalg.norm(v)

    for restart in range(max_restarts + 1):
        rows = k_max + 1
        Q = np.zeros((rows, n))
        Q[0] = q0 / np.linalg.norm(q0)
        alpha, beta, hist = [], [], []
        ritz = None
        for j in range(k_cap):
            w = apply(Q[j])
            a = float(Q[j] @ w)
            alpha.append(a)
            w = w - a * Q[j] - (beta[-1] * Q[j - 1] if j > 0 else 0.0)
            for _ in range(2):                                   # full reorthogonalisation, twice
                w -= Q[:j + 1].T @ (Q[:j + 1] @ w)
            b = float(np.linalg.norm(w))
            if j >= 1 and (j < 24 or j % 4 == 3 or j + 1 >= k_cap):      # the convergence tests: every step at first, then every fourth
                theta, ritz = _tridiag_eigh(np.array(alpha), np.array(beta))
                res = b * abs(ritz[-1]) / abs(theta)
                if res <= rtol:                                  # residual bound of the largest Ritz pair
                    return done(theta, ritz, Q)
                # a cluster at the top of the spectrum (K^-1 of a matrix whose small eigenvalues sit on the nugget): the residual of ONE
                # Ritz vector stays large while the Ritz VALUE -- second order in that residual -- has long settled; any vector of the
                # cluster's invariant subspace serves the gradient formula as well as the one a dense eigensolver would pick
                old = [t for (jj, t) in hist if jj <= j - 6]
                hist.append((j, theta))
                if j >= 12 and old and abs(theta - old[-1]) <= 1e-13 * abs(theta):
                    return done(theta, ritz, Q)
            elif j == 0:
                theta = a
            if b <= 1e-300 or b <= 4e-16 * abs(theta) or not np.isfinite(b):   # invariant subspace (what is left of w is rounding): the Ritz
                return done(theta, ritz, Q)                                      # values are exact; nothing to normalise a next vector from
            if j + 1 >= k_cap:
                break
            if j + 2 > rows:                                     # grow the basis
                rows = min(k_cap + 1, 2 * rows)
                Q = np.vstack((Q, np.zeros((rows - Q.shape[0], n))))
            beta.append(b)
            Q[j + 1] = w / b
        if k_cap >= n:                                           # the full Krylov space: exact up to rounding
            return done(theta, ritz, Q)
        if ritz is None:
            break
        q0 = Q[:len(ritz)].T @ ritz                              # restart from the Ritz vector
    warnings.warn(f'Lanczos: largest Ritz value not converged after {max_restarts + 1} runs of {k_cap} steps '
                  f'(relative residual bound {res:.2e} > {rtol:.1e}); the condition number is a lower bound',
                  LanczosNotConverged, stacklevel=2)
    return done(theta, ritz, Q)


def cond_from_factor(apply_mat, apply_inv, n, want_vectors=False):
    """cond_2 = lambda_max(K) * lambda_max(K^-1), each from its own Lanczos run.  With want_vectors also
    (lambda_min, v_max, v_min): what the gradient of the condition number needs (GpHparaCon.py:175-197)."""
    if n == 1:
        return (1.0, 1.0, np.ones(1), np.ones(1)) if want_vectors else 1.0
    if not want_vectors:
        return float(lanczos_largest(apply_mat, n) * lanczos_largest(apply_inv, n, seed=1))
    lam_max, v_max = lanczos_largest(apply_mat, n, rtol=1e-12, want_vector=True)
    inv_lam_min, v_min = lanczos_largest(apply_inv, n, rtol=1e-12, seed=1, want_vector=True)
    return float(lam_max * inv_lam_min), 1.0 / float(inv_lam_min), v_max, v_min
