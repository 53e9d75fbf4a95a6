"""CPU oracle: a NumPy/SciPy restatement of the reference's gradient-enhanced GP likelihood hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under `gpgradpy_amd/` imports this module; only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` do, and only as the checker /
the timed CPU baseline.  The product path is the HIP library behind `include/gpgrad.h`.

Parity status: PINNED.  `tests/test_oracle_golden.py` checks every function below against the
fixtures in `tests/golden/*.npz`, which were produced by running the reference itself in the build
container (`tests/golden/gen_golden.py`; the reference's own unit tests hold no value fixtures for
this path).

All `file:line` citations are relative to the reference tree (marchildon/gpgradpy @ v2).
Third-party arithmetic on the path: `scipy.linalg.cho_factor/cho_solve` (LAPACK dpotrf/dpotrs); the
reference pins no SciPy version (setup.py:27), the oracle calls the same routines.

Matrix layout (reference KernelSqExp.py:381-408, CommonFun.py:151-173): derivative-major blocks.
Row/column index r = blk * n + a with blk 0 = function values, blk i+1 = d/dx_i at point a.
"""
from __future__ import annotations

from dataclasses import dataclass, field
import numpy as np
from scipy.linalg import cho_factor, cho_solve

SQEXP, MA5F2, RATQU = 0, 1, 2
_KERNEL_ID = {"SqExp": SQEXP, "Ma5f2": MA5F2, "RatQu": RATQU}


def _kern_split(kernel):
    """`kernel` is a name, or for the rational quadratic kernel the pair ("RatQu", alpha) -- alpha is
    HparaOptzVal.kernel of the reference (KernelRatQuad.py:468, default 2: :849)."""
    if isinstance(kernel, (tuple, list)):
        return kernel[0], float(kernel[1])
    return kernel, (2.0 if kernel == "RatQu" else None)


# --------------------------------------------------------------------------------------------
# scalars / small host pieces
# --------------------------------------------------------------------------------------------
def calc_nugget(n_eval: int, dim: int, kernel: str, use_grad: bool, wellcond: str = "precon",
                cond_max_target: float = 1e10):
    """(eta_Kbase, eta_Kgrad) -- reference GpWellCond.py:116-154 (cond_eta_set_mtd='Kbase_eta')."""
    eta_base = n_eval / (cond_max_target - 1.0)
    if not use_grad:
        return eta_base, np.nan
    if n_eval == 1:
        return eta_base, eta_base
    if wellcond == "precon":
        d = float(dim)
        kernel = _kern_split(kernel)[0]
        if kernel in ("SqExp", "RatQu"):                       # GpWellCond.py:129
            s = np.sqrt(1.0 + 4.0 * d)
            ub = 0.5 * (n_eval - 1) * (1.0 + s) * np.exp(-(1.0 + 2.0 * d - s) / (4.0 * d))
        elif kernel == "Ma5f2":
            r3 = np.sqrt(3.0 * d)
            al = (r3 - 1.0 + np.sqrt(15.0 * d + 2.0 * r3 + 1.0)) / (2.0 * (3.0 * d + r3))
            ub = (n_eval - 1) * (1.0 + (d + r3) * al + d * (1.0 + r3) * al ** 2) * np.exp(-r3 * al)
        else:
            raise ValueError(kernel)
        return eta_base, (1.0 + ub) / (cond_max_target - 1.0)
    return eta_base, eta_base  # wellcond == 'base'


def make_data_vec(fval, fgrad=None):
    """y = [f, d1 f (all points), d2 f (all points), ...] -- reference CommonFun.py:151-173."""
    fval = np.atleast_1d(np.asarray(fval, dtype=float))
    if fgrad is None:
        return fval
    return np.concatenate((fval, np.asarray(fgrad, dtype=float).T.ravel()))


def calc_noise_vec(n, d, use_grad, std_f, std_g, var_fval=None, var_fgrad=None, n_grad=None):
    """Noise variance per row of K -- reference Kernel.py:309-357.

    std_f / std_g None  => that variance is a hyperparameter (var_fval / var_fgrad).
    n_grad = number of points whose gradient is used (default: all, or the rows of std_g)."""
    if not use_grad:
        return np.asarray(std_f, dtype=float) ** 2 if std_f is not None else np.full(n, float(var_fval))
    if n_grad is None:
        n_grad = n if std_g is None else np.asarray(std_g).shape[0]
    out = np.zeros(n + n_grad * d)
    out[:n] = np.asarray(std_f, dtype=float) ** 2 if std_f is not None else float(var_fval)
    out[n:] = (np.asarray(std_g, dtype=float) ** 2).T.ravel() if std_g is not None else float(var_fgrad)
    return out


# --------------------------------------------------------------------------------------------
# kernels
# --------------------------------------------------------------------------------------------
def _rtensor(X, Y):
    """R[k,a,b] = X[a,k] - Y[b,k] -- reference CommonFun.py:56-84."""
    return np.transpose(X[:, None, :] - Y[None, :, :], (2, 0, 1))


def kern_base(X, Y, theta, kernel):
    """Gradient-free kernel matrix -- reference KernelSqExp.py:16-46, KernelMatern5f2.py:16-52."""
    R = _rtensor(X, Y)
    s = np.tensordot(theta, R ** 2, axes=1)
    kernel, alpha = _kern_split(kernel)
    if kernel == "SqExp":
        return np.exp(-s)
    if kernel == "RatQu":                                      # KernelRatQuad.py:16-48
        return (1.0 + s / alpha) ** (-alpha)
    nu = np.sqrt(s)
    r5 = np.sqrt(5.0)
    return (1.0 + r5 * nu + (5.0 / 3.0) * nu ** 2) * np.exp(-r5 * nu)


def _mask_index(n, d, mask):
    """Rows of the full [n (d+1)] ordering kept by a bvec_use_grad mask (KernelSqExp.py:349-377)."""
    g = np.flatnonzero(mask)
    return np.concatenate([np.arange(n)] + [n * (i + 1) + g for i in range(d)])


def kern_grad(X, Y, theta, kernel, grad_cols=True, mask1=None, mask2=None):
    """Gradient-enhanced kernel matrix [(n1 + n1g d) x (n2 + n2g d)].

    Reference KernelSqExp.py:320-410 (SqExp) and KernelMatern5f2.py:352-450 (Matern 5/2).
    grad_cols=False keeps only the first n2 columns (the cross matrix of GpEvalModel.py:135-139);
    mask1 / mask2 = bvec_use_grad1 / bvec_use_grad2 (points whose gradient rows / columns are kept)."""
    if mask1 is not None or mask2 is not None:
        K = kern_grad(X, Y, theta, kernel, grad_cols)
        n1, d = X.shape
        n2 = Y.shape[0]
        rows = slice(None) if mask1 is None else _mask_index(n1, d, mask1)
        cols = slice(None) if (mask2 is None or not grad_cols) else _mask_index(n2, d, mask2)
        return K[rows][:, cols]
    n1, d = X.shape
    n2 = Y.shape[0]
    R = _rtensor(X, Y)
    s = np.tensordot(theta, R ** 2, axes=1)
    kernel, alpha = _kern_split(kernel)
    if kernel == "RatQu":                                      # KernelRatQuad.py:439-554
        B = 1.0 + s / alpha
        K00 = B ** (-alpha)
        m1 = B ** (-alpha - 1.0)
        m2 = B ** (-alpha - 2.0)
        const = 4.0 * (1.0 + 1.0 / alpha)
        c1 = 2.0 * theta[:, None, None] * R * m1
        diag_add = [2.0 * theta[i] * m1 for i in range(d)]
        cross = lambda i, j: -const * theta[i] * theta[j] * R[i] * R[j] * m2
    elif kernel == "SqExp":
        E = np.exp(-s)
        K00 = E
        c1 = 2.0 * theta[:, None, None] * R * E                     # +2 th_i R_i E  (block [0, i+1])
        diag_add = [2.0 * theta[i] * E for i in range(d)]
        cross = lambda i, j: -4.0 * theta[i] * theta[j] * R[i] * R[j] * E
    elif kernel == "Ma5f2":
        nu = np.sqrt(s)
        r5 = np.sqrt(5.0)
        A = np.exp(-r5 * nu)
        m1 = (5.0 / 3.0) * (1.0 + r5 * nu) * A
        K00 = (1.0 + r5 * nu + (5.0 / 3.0) * nu ** 2) * A
        c1 = theta[:, None, None] * R * m1
        diag_add = [theta[i] * m1 for i in range(d)]
        cross = lambda i, j: -(25.0 / 3.0) * theta[i] * theta[j] * R[i] * R[j] * A
    else:
        raise ValueError(kernel)
    nc = n2 * (d + 1) if grad_cols else n2
    K = np.zeros((n1 * (d + 1), nc))
    K[:n1, :n2] = K00
    for i in range(d):
        ri = slice(n1 * (i + 1), n1 * (i + 2))
        K[ri, :n2] = -c1[i]
        if grad_cols:
            ci = slice(n2 * (i + 1), n2 * (i + 2))
            K[:n1, ci] = c1[i]
            for j in range(d):
                cj = slice(n2 * (j + 1), n2 * (j + 2))
                K[ri, cj] = cross(i, j) + (diag_add[i] if i == j else 0.0)
    return K


# --------------------------------------------------------------------------------------------
# regularise + factorise
# --------------------------------------------------------------------------------------------
@dataclass
class Factor:
    chofac: tuple | None          # scipy cho_factor tuple of Kcov (precon: (P L, True)); None on failure
    pvec: np.ndarray | None
    Kern: np.ndarray
    Kcov: np.ndarray
    etaK: float
    varK_mat: float


def calc_all_K_w_chofac(X, theta, kernel, use_grad, wellcond, etaK, noise_vec, varK=1.0,
                        as_written=False, grad_mask=None):
    """Kernel + noise + (preconditioner) + nugget + Cholesky -- reference Kernel.py:140-307.

    as_written=True keeps the reference's dense diagonal-matrix products (Kernel.py:224-227,237,252)
    so that the CPU baseline is timed on the algorithm the reference actually runs; False applies the
    same scaling element-wise (identical values up to rounding)."""
    Kern = kern_grad(X, X, theta, kernel, mask1=grad_mask, mask2=grad_mask) if use_grad \
        else kern_base(X, X, theta, kernel)
    N = Kern.shape[0]
    Kw = Kern + np.diag(noise_vec / varK)
    if wellcond == "precon":
        pvec = np.sqrt(np.diag(Kw))
        if as_written:
            P, Pinv = np.diag(pvec), np.diag(1.0 / pvec)
            Kcor = Pinv @ Kw @ Pinv
            Kp = varK * (Kcor + etaK * np.eye(N))
            Kcov = P @ Kp @ P
        else:
            Kp = varK * (Kw / np.outer(pvec, pvec))
            Kp[np.diag_indices(N)] += varK * etaK
            Kcov = Kp * np.outer(pvec, pvec)
        try:
            L, low = cho_factor(Kp, lower=True)
            PL = (P @ L) if as_written else pvec[:, None] * L
            chofac = (PL, low)
        except np.linalg.LinAlgError:
            chofac = None
        return Factor(chofac, pvec, Kern, Kcov, etaK, varK)
    Kcov = varK * (Kw + etaK * np.eye(N))
    try:
        chofac = cho_factor(Kcov)            # upper, as the reference's base branch (Kernel.py:291)
    except np.linalg.LinAlgError:
        chofac = None
    return Factor(chofac, None, Kern, Kcov, etaK, varK)


# --------------------------------------------------------------------------------------------
# likelihood
# --------------------------------------------------------------------------------------------
@dataclass
class LkdResult:
    ok: bool
    hp_beta: np.ndarray | None = None
    hp_varK: float | None = None
    ln_det_Kmat: float | None = None
    ln_lkd: float | None = None
    alpha: np.ndarray | None = None
    factor: Factor | None = field(default=None, repr=False)


def _vand_aug(n, d, use_grad, n_grad=None):
    """V = [1_n ; 0] -- reference GpMeanFun.py:172-191 with poly_ord_0 (:195-204)."""
    n_grad = n if n_grad is None else n_grad
    V = np.zeros((n + n_grad * d if use_grad else n, 1))
    V[:n, 0] = 1.0
    return V


def _gls_mean(chofac, V, y):
    """beta = (V' K^-1 V)^-1 (K^-1 V)' y -- reference GpMeanFun.py:69-122."""
    KiV = cho_solve(chofac, V)
    term1 = np.linalg.solve(V.T @ KiV, KiV.T)
    beta = term1 @ y
    return beta, V @ beta


def calc_lkd(X, y, theta, kernel, use_grad, wellcond, etaK, noise_vec, noisy, varK=None,
             as_written=False, grad_mask=None):
    """One marginal-log-likelihood evaluation (value only) -- reference CalcLkd.py:270-346.

    noisy=False: CalcLkd.py:30-95 + :149-181 (varK in closed form, matrix built with varK = 1).
    noisy=True : CalcLkd.py:185-251 (varK is a hyperparameter)."""
    n, d = X.shape
    fac = calc_all_K_w_chofac(X, theta, kernel, use_grad, wellcond, etaK, noise_vec,
                              varK=(varK if noisy else 1.0), as_written=as_written, grad_mask=grad_mask)
    if fac.chofac is None:
        return LkdResult(False, factor=fac)
    V = _vand_aug(n, d, use_grad, None if grad_mask is None else int(np.sum(grad_mask)))
    beta, mean_val = _gls_mean(fac.chofac, V, y)
    res = y - mean_val
    alpha = cho_solve(fac.chofac, res)
    N = y.size
    ln_det = 2.0 * np.sum(np.log(np.diag(fac.chofac[0])))
    if noisy:
        return LkdResult(True, beta, None, ln_det, -(ln_det + res @ alpha) / 2.0, alpha, fac)
    varK_opt = max(1e-32, (res @ alpha) / N)
    return LkdResult(True, beta, varK_opt, ln_det, -(N * np.log(varK_opt) + ln_det) / 2.0, alpha, fac)


def _kcov_only(X, theta, kernel, use_grad, wellcond, etaK, n, d, noise_fun, varK):
    """Regularised covariance as a function of the hyperparameters (no factorisation)."""
    Kern = kern_grad(X, X, theta, kernel) if use_grad else kern_base(X, X, theta, kernel)
    Kw = Kern + np.diag(noise_fun() / varK)
    if wellcond == "precon":
        return varK * (Kw + etaK * np.diag(np.diag(Kw)))       # = P (Kcor + eta I) P, Kernel.py:227-237
    return varK * (Kw + etaK * np.eye(Kw.shape[0]))


def calc_lkd_grad(X, y, theta, kernel, use_grad, wellcond, etaK, std_f, std_g, noisy, varK=None,
                  var_fval=None, var_fgrad=None, rel_step=1e-6):
    """d ln_lkd / d hp by the adjoint formula of the reference (CalcLkd.py:170-177 noise-free, :230-235 noisy)
    with the derivative tensor d Kcov / d hp_k (GpHparaGrad.py:13-155) replaced by central differences of the
    covariance matrix itself (rel. step 1e-6: truncation ~1e-12, rounding ~1e-10 relative per entry).
    Returns the vector ordered [theta(d), alpha? (RatQu), varK?, var_fval?, var_fgrad?] like HparaOptzInfo."""
    n, d = X.shape
    nv = lambda vf=var_fval, vg=var_fgrad: calc_noise_vec(n, d, use_grad, std_f, std_g, vf, vg)
    vK = varK if noisy else 1.0
    r0 = calc_lkd(X, y, theta, kernel, use_grad, wellcond, etaK, nv(), noisy, varK=varK)
    N = y.size
    Kinv = cho_solve(r0.factor.chofac, np.eye(N))
    al = r0.alpha
    Lam = (0.5 if noisy else 1.0 / (2.0 * r0.hp_varK)) * np.outer(al, al) - 0.5 * Kinv
    out = []

    def dcov(f_plus, f_minus, h):
        return (f_plus - f_minus) / (2.0 * h)
    for k in range(d):
        h = rel_step * theta[k]
        tp, tm = theta.copy(), theta.copy()
        tp[k] += h
        tm[k] -= h
        G = dcov(_kcov_only(X, tp, kernel, use_grad, wellcond, etaK, n, d, nv, vK),
                 _kcov_only(X, tm, kernel, use_grad, wellcond, etaK, n, d, nv, vK), h)
        out.append(np.sum(G * Lam))
    kname, alpha = _kern_split(kernel)
    if kname == "RatQu":                       # the kernel's own hyperparameter sits between theta and varK (GpHparaOptz.py:90-96)
        h = rel_step * alpha
        G = dcov(_kcov_only(X, theta, (kname, alpha + h), use_grad, wellcond, etaK, n, d, nv, vK),
                 _kcov_only(X, theta, (kname, alpha - h), use_grad, wellcond, etaK, n, d, nv, vK), h)
        out.append(np.sum(G * Lam))
    if noisy:
        h = rel_step * varK
        G = dcov(_kcov_only(X, theta, kernel, use_grad, wellcond, etaK, n, d, nv, varK + h),
                 _kcov_only(X, theta, kernel, use_grad, wellcond, etaK, n, d, nv, varK - h), h)
        out.append(np.sum(G * Lam))
        if std_f is None:
            h = rel_step * var_fval
            G = dcov(_kcov_only(X, theta, kernel, use_grad, wellcond, etaK, n, d, lambda: nv(var_fval + h, var_fgrad), varK),
                     _kcov_only(X, theta, kernel, use_grad, wellcond, etaK, n, d, lambda: nv(var_fval - h, var_fgrad), varK), h)
            out.append(np.sum(G * Lam))
        if use_grad and std_g is None:
            h = rel_step * var_fgrad
            G = dcov(_kcov_only(X, theta, kernel, use_grad, wellcond, etaK, n, d, lambda: nv(var_fval, var_fgrad + h), varK),
                     _kcov_only(X, theta, kernel, use_grad, wellcond, etaK, n, d, lambda: nv(var_fval, var_fgrad - h), varK), h)
            out.append(np.sum(G * Lam))
    return np.array(out)


# --------------------------------------------------------------------------------------------
# posterior
# --------------------------------------------------------------------------------------------
@dataclass
class EvalModel:
    X: np.ndarray
    theta: np.ndarray
    kernel: str
    use_grad: bool
    beta: np.ndarray
    varK: float
    chofac: tuple
    alpha: np.ndarray
    grad_mask: np.ndarray | None = None


def setup_eval_model(X, y, theta, kernel, use_grad, wellcond, etaK, noise_vec, beta, varK, grad_mask=None):
    """Reference GpEvalModel.py:17-57.  Built with b_normlz_w_varK=True: varK := 1 *before* the
    noise is divided by it (Kernel.py:196-197,218), so known noise is not scaled by the true varK."""
    n, d = X.shape
    fac = calc_all_K_w_chofac(X, theta, kernel, use_grad, wellcond, etaK, noise_vec, varK=1.0, grad_mask=grad_mask)
    if fac.chofac is None:
        return None
    V = _vand_aug(n, d, use_grad, None if grad_mask is None else int(np.sum(grad_mask)))
    alpha = cho_solve(fac.chofac, y - V @ beta)
    return EvalModel(X, np.asarray(theta, float), kernel, use_grad, np.asarray(beta, float), float(varK),
                     fac.chofac, alpha, grad_mask)


def eval_model_grad(m: EvalModel, xq):
    """(mu, sig, dmudx, dsigdx) -- reference GpEvalModel.py:59-198 with calc_grad=True (:133-140, :319-354)."""
    xq = np.atleast_2d(np.asarray(xq, dtype=float))
    nx, d = xq.shape
    Kg = kern_grad(m.X, xq, m.theta, m.kernel, grad_cols=True, mask1=m.grad_mask if m.use_grad else None)
    if not m.use_grad:
        Kg = Kg[:m.X.shape[0], :]                              # GpEvalModel.py:142-144
    Kyx, dKxy_dx = Kg[:, :nx], Kg[:, nx:].T
    sol = cho_solve(m.chofac, Kyx)
    sig2 = 1.0 - np.einsum("ij,ij->j", Kyx, sol)
    sig2[sig2 < 0] = 0.0
    sig = np.sqrt(sig2) * np.sqrt(m.varK)
    mu = m.beta[0] + Kyx.T @ m.alpha
    dmudx = np.reshape(dKxy_dx @ m.alpha, (nx, d), order="F")
    inv_sig = np.divide(1.0, sig, out=np.zeros_like(sig), where=sig != 0)
    term2 = np.sum(dKxy_dx * np.kron(np.ones((d, 1)), sol.T), axis=1) * m.varK
    dsigdx = -inv_sig[:, None] * term2.reshape((nx, d), order="F")
    return mu, sig, dmudx, dsigdx


def kern_hess_x(x, Y, theta, kernel, use_grad):
    """Second derivatives of the cross-covariance row K(x, [Y, dY]) with respect to the query point x:
    [d, d, N] -- reference KernelSqExp.py:48-88 (base columns), :412-468 (gradient columns),
    KernelMatern5f2.py:54-94, :453-530; r = x - y as in calc_Rtensor(x2model, x_eval)."""
    x = np.asarray(x, dtype=float).ravel()
    n, d = Y.shape
    R = x[None, :] - Y                                   # [n, d]
    N = n * (d + 1) if use_grad else n
    H = np.zeros((d, d, N))
    kernel, alpha = _kern_split(kernel)
    if kernel == "RatQu":                                # KernelRatQuad.py:51-136 (base columns), :556-632 (gradient columns)
        B = 1.0 + np.sum(theta * R ** 2, axis=1) / alpha
        f1, f2, f3 = B ** (-alpha - 1.0), B ** (-alpha - 2.0), B ** (-alpha - 3.0)
        s1, s2 = 1.0 + 1.0 / alpha, (1.0 + 1.0 / alpha) * (1.0 + 2.0 / alpha)
        for k in range(d):
            for i in range(d):
                H[k, i, :n] = -2 * theta[i] * (i == k) * f1 + 4 * s1 * theta[i] * theta[k] * R[:, i] * R[:, k] * f2
                if use_grad:
                    for j in range(d):
                        c0 = n + j * n
                        H[k, i, c0:c0 + n] = (-4 * s1 * theta[i] * theta[j] * ((i == k) * R[:, j] + (j == k) * R[:, i])
                                              - 4 * s1 * (i == j) * theta[i] * theta[k] * R[:, k]) * f2 \
                            + 8 * s2 * theta[i] * theta[j] * theta[k] * R[:, i] * R[:, j] * R[:, k] * f3
    elif kernel == "SqExp":
        K = np.exp(-np.sum(theta * R ** 2, axis=1))
        for k in range(d):
            for i in range(d):
                H[k, i, :n] = (-2 * theta[i] * (i == k) + 4 * theta[i] * theta[k] * R[:, i] * R[:, k]) * K
                if use_grad:
                    for j in range(d):
                        c0 = n + j * n
                        H[k, i, c0:c0 + n] = (-4 * theta[i] * theta[j] * ((i == k) * R[:, j] + (j == k) * R[:, i])
                                              - 4 * (i == j) * theta[i] * theta[k] * R[:, k]
                                              + 8 * theta[i] * theta[j] * theta[k] * R[:, i] * R[:, j] * R[:, k]) * K
    elif kernel == "Ma5f2":
        nu = np.sqrt(np.sum(theta * R ** 2, axis=1))
        base = np.exp(-np.sqrt(5) * nu)
        A = (5 / 3) * (1 + np.sqrt(5) * nu) * base
        inv_nu = 1 / np.maximum(nu, 1e-16)
        Rt = R * theta
        for k in range(d):
            for i in range(d):
                H[k, i, :n] = (25 / 3) * Rt[:, i] * Rt[:, k] * base - (i == k) * theta[k] * A
                if use_grad:
                    for j in range(d):
                        c0 = n + j * n
                        H[k, i, c0:c0 + n] = -(25 / 3) * (theta[i] * (i == k) * Rt[:, j] + theta[j] * (j == k) * Rt[:, i]
                                                        + theta[i] * (i == j) * Rt[:, k]
                                                        - np.sqrt(5) * inv_nu * Rt[:, i] * Rt[:, j] * Rt[:, k]) * base
    else:
        raise ValueError(kernel)
    return H


def eval_model_hess(m: EvalModel, xq):
    """(mu, sig, dmudx, dsigdx, d2mudx2, d2sigdx2) at ONE point -- reference GpEvalModel.py:59-198 with
    calc_hess=True (:148-150, :176-178), calc_d2mudx2 :355-363, calc_d2sigdx2 :365-380.  No gradient masks."""
    xq = np.atleast_2d(np.asarray(xq, dtype=float))
    assert xq.shape[0] == 1 and m.grad_mask is None
    mu, sig, dmudx, dsigdx = eval_model_grad(m, xq)
    d = xq.shape[1]
    Kg = kern_grad(m.X, xq, m.theta, m.kernel, grad_cols=True)
    if not m.use_grad:
        Kg = Kg[:m.X.shape[0], :]
    Kyx, dKxy_dx = Kg[:, :1], Kg[:, 1:].T
    H = kern_hess_x(xq[0], m.X, m.theta, m.kernel, m.use_grad)
    sol = cho_solve(m.chofac, Kyx)                                        # K^-1 Kyx  [N, 1]
    d2mudx2 = H @ m.alpha
    term1 = H @ sol[:, 0]
    term2 = dKxy_dx @ cho_solve(m.chofac, dKxy_dx.T)
    d2sig2 = -2 * m.varK * (term1 + term2)
    sig_mod = sig[0] if sig[0] != 0 else np.nan
    d2sigdx2 = (1 / (2 * sig_mod)) * (d2sig2 - 2 * np.outer(dsigdx[0], dsigdx[0]))
    return mu, sig, dmudx, dsigdx, d2mudx2[None], d2sigdx2[None]


def eval_model(m: EvalModel, xq):
    """Posterior mean and standard deviation -- reference GpEvalModel.py:59-198 (calc_grad=False)."""
    xq = np.atleast_2d(np.asarray(xq, dtype=float))
    if m.use_grad:
        Kyx = kern_grad(m.X, xq, m.theta, m.kernel, grad_cols=False, mask1=m.grad_mask)
    else:
        Kyx = kern_base(m.X, xq, m.theta, m.kernel)
    sol = cho_solve(m.chofac, Kyx)
    sig2 = 1.0 - np.einsum("ij,ij->j", Kyx, sol)
    sig2[sig2 < 0] = 0.0
    mu = m.beta[0] + Kyx.T @ m.alpha
    return mu, np.sqrt(sig2) * np.sqrt(m.varK)


# --------------------------------------------------------------------------------------------
# multi-start selection
# --------------------------------------------------------------------------------------------
def multistart_lkd(X, y, kernel, etaK, hp_rows, noise_vec=None, noisy=False, log10=True):
    """ln_lkd for every restart row + nanargmax -- reference GpHparaX0.py:33-59.

    Row layout (GpHparaOptz.py:44-138): [theta(d), varK?]; entries are log10 by default
    (GaussianProcess.py:41-43).  Failed factorizations contribute NaN."""
    n, d = X.shape
    hp_rows = np.atleast_2d(hp_rows)
    out = np.full(hp_rows.shape[0], np.nan)
    nv = np.zeros(n * (d + 1)) if noise_vec is None else noise_vec
    for i, row in enumerate(hp_rows):
        vals = 10.0 ** row if log10 else row
        r = calc_lkd(X, y, vals[:d], kernel, True, "precon", etaK, nv, noisy, varK=(vals[d] if noisy else None))
        if r.ok:
            out[i] = r.ln_lkd
    return out, int(np.nanargmax(out))


# --------------------------------------------------------------------------------------------
# synthetic workload of BASELINE.md section 3 / SURVEY.md 8(d)
# --------------------------------------------------------------------------------------------
def rosenbrock(x, a=10.0):
    n, d = x.shape
    if d == 1:
        return (1 - x[:, 0]) ** 2 + a * x[:, 0] ** 4, (-2 * (1 - x[:, 0]) + 4 * a * x[:, 0] ** 3)[:, None]
    f = np.zeros(n)
    g = np.zeros((n, d))
    for k in range(d - 1):
        t = x[:, k + 1] - x[:, k] ** 2
        f += a * t ** 2 + (1 - x[:, k]) ** 2
        g[:, k] += -4 * a * x[:, k] * t - 2 * (1 - x[:, k])
        g[:, k + 1] += 2 * a * t
    return f, g


def synthetic_design(n, d, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.uniform(-2.0, 2.0, (n, d))
    f, g = rosenbrock(X)
    return X, f, g
