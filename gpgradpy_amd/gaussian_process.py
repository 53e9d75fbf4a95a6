"""Host-side mirror of the reference's `GaussianProcess` facade for the likelihood hot path.

Same method names, argument meaning, return shapes and error behaviour as the reference
(gpgradpy/src/GaussianProcess.py, kernel/Kernel.py, optz/CalcLkd.py, eval/GpEvalModel.py; file:line
citations below are relative to gpgradpy/src/), but every O(N^2)/O(N^3) step runs in the HIP library
behind include/gpgrad.h.  What the path does not cover raises NotImplementedError -- there is no
NumPy/SciPy fallback for any part of the computation.
"""
import copy
import ctypes as C
import time
import weakref

import numpy as np

from . import _lib
from .hpara import HparaOptzInfo, HparaOptzVal, LkdInfo
from .rescaling import Rescaling, calc_dist_min, calc_dist_max
from .hpara_optz import HparaOptz


class DeviceChoFactor:
    """`KernEta_chofac` of a model whose factor lives in HBM: behaves like SciPy's `(c, lower)` pair --
    `scipy.linalg.cho_solve(GP.KernEta_chofac, b)` unpacks it -- and downloads the lower-triangular `P L`
    (Kernel.py:252) from the posterior's own device workspace on first use (gpg_get_matrix which = 4); the hot path
    (`eval_model`, `eval_model_var`) never triggers the download."""

    def __init__(self, gp):
        self._gp = weakref.ref(gp)       # no reference cycle: a dropped GaussianProcess frees its device memory at once
        self._fac = None

    def _array(self):
        if self._fac is None:
            gp = self._gp()
            if gp is None or gp.KernEta_chofac is not self:
                raise RuntimeError('this KernEta_chofac belongs to an earlier setup_eval_model(): its device factor has been replaced')
            fac = np.empty((gp.n_data, gp.n_data))
            rc = gp._lib.gpg_get_matrix(gp._ctx, None, 4, _lib.as_dp(fac))
            if rc != 0:
                raise _lib.GpgError(f'gpg_get_matrix failed ({rc}): {gp._err()}')
            self._fac = fac
        return self._fac

    def __iter__(self):
        return iter((self._array(), True))

    def __len__(self):
        return 2

    def __getitem__(self, i):
        return (self._array(), True)[i]


class GaussianProcess(HparaOptz):
    # ---- options read by the hot path (reference GaussianProcess.py:27-113) -------------------------
    optz_log_hp_theta = True
    optz_log_hp_var = True
    optz_log_hp_kernel = True
    lkd_use_adj_mtd = True
    lkd_optz_start_mtd = 'hp_best'
    lkd_hp_best_n_eval = 40
    lkd_varK_pnlt_use = False
    lkd_varK_pnlt_lb_var = 0.1
    lkd_varK_pnlt_c1 = 1.0
    lkd_varK_pnlt_c2 = 10.0
    hp_theta_init = 1e-2
    hp_varK_init = 1.0
    hp_var_fval_init = 0.0
    hp_var_fgrad_init = 0.0
    wellcond_mtd_avail = ['base', 'precon', 'rescale_origin', 'rescale_eta_vary', 'dflt_vmin', 'dflt_vmax']
    cond_eta_set_mtd = 'Kbase_eta'
    cond_eta_is_const = True
    cond_eta_dflt = 1e-8
    cond_max_target = 1e10
    cond_max = 1e10
    cond_max_abs = 1e16
    cond_norm = 2                 # GaussianProcess.py:104: 2 (Lanczos through the factor) or 'fro' (gpg_cond_fro)
    cond_dist_min_dflt = 1        # wellcond_mtd 'dflt_vmin' (GaussianProcess.py:106)
    cond_dist_max_dflt = 1        # wellcond_mtd 'dflt_vmax' (:107)
    cond_vreq_max_iter = 3        # rescale methods: re-optimisations with a theta-informed anisotropic scaling (:111)
    vmin_rescale_eta_vary = 1.0   # (:112)
    cond_vreq_iter_tol = 1e-1     # (:113)
    _vmin_req_grad = np.nan
    _vmin_init = np.nan
    DataScl = None

    b_optz_hp_kernel = True
    b_use_data_scl = False
    b_has_noisy_data = None
    b_optz_var_fval = None
    b_optz_var_fgrad = None
    bvec_use_grad = None
    hp_vals = None
    _time_chofac = 0
    _last_hp_vec = None

    kernel_types = ('SqExp', 'Ma5f2', 'RatQu')

    def __init__(self, dim, use_grad, kernel_type='SqExp', wellcond_mtd='precon', mean_fun_type='poly_ord_0',
                 path_data_surr='baye_data_surr', surr_name='obj_', device=0):
        # reference GaussianProcess.py:138-190
        assert isinstance(dim, int), 'dim must be an integer'
        assert isinstance(use_grad, bool), 'use_grad must be of type bool'
        assert isinstance(kernel_type, str), 'kernel_type must be of type str'
        self.dim = dim
        self.use_grad = use_grad
        self.set_wellcond_mtd(wellcond_mtd)
        self.path_data_surr = path_data_surr
        self.surr_name = surr_name
        if kernel_type not in self.kernel_types:
            raise Exception('Kernel type is not available')                       # Kernel.py:108-109
        self.kernel_type = kernel_type
        if kernel_type == 'RatQu':                                              # KernelRatQuad.py:849-850, Kernel.py:84-85
            self.hp_kernel_default = 2                                          # (value + posterior mean / std only here:
            self.hp_kernel_range = [1e-3, 10]                                   #  no likelihood / posterior derivatives)
        else:
            self.hp_kernel_default = None                                       # KernelSqExp.py:577, KernelMatern5f2.py:651
            self.hp_kernel_range = None
        self.kernel_has_hp = self.hp_kernel_default is not None                 # Kernel.py:111-112
        self.hp_kernel = self.hp_kernel_default
        if mean_fun_type != 'poly_ord_0':
            raise Exception(f'mean_fun_type = {mean_fun_type} not available')     # GpMeanFun.py:203-204
        self.mean_fun_type = mean_fun_type
        self.n_beta_coeff = 1
        self.beta_var_npara = 1
        self.device = int(device)
        self._lib = _lib.load()
        self._ctx = None
        self._ctx_shape = None
        self.KernEta_chofac = None
        self.invKernEta_fdiff = None

    def close(self):
        """Free the device context (every workspace, factor and buffer of this model) now; the object can take new data
        afterwards (set_data creates a fresh context)."""
        if getattr(self, '_ctx', None):
            self._lib.gpg_destroy(self._ctx)
            self._ctx = None
            self._ctx_shape = None
            self.KernEta_chofac = None
            self.invKernEta_fdiff = None
            self._eval_ready = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- small host pieces ---------------------------------------------------------------------------
    def set_wellcond_mtd(self, wellcond_mtd):
        # reference GaussianProcess.py:192-217
        assert wellcond_mtd in self.wellcond_mtd_avail, \
            f'Requested method not available, wellcond_mtd : {wellcond_mtd}'
        if wellcond_mtd == 'rescale_eta_vary':
            self.cond_eta_is_const = False                    # nugget from the largest absolute row sum of the matrix
        if not self.use_grad:
            wellcond_mtd = 'base'
        self.wellcond_mtd = wellcond_mtd
        self.b_use_cond_cstr = wellcond_mtd != 'precon'
        # the rescale / dflt_v methods work on shifted and scaled data (gpgradpy_amd/rescaling.py); what the device sees
        # then is the 'base' covariance of the scaled points
        self.b_use_data_scl = ('rescale' in wellcond_mtd) or ('dflt_v' in wellcond_mtd)

    def theta2gamma(self, theta):
        # KernelSqExp.py:580-583 / KernelMatern5f2.py:654-657
        # (RatQu as SqExp: KernelRatQuad.py:853-860)
        return np.sqrt((5.0 / 3.0) * theta) if self.kernel_type == 'Ma5f2' else np.sqrt(2 * theta)

    def gamma2theta(self, gamma):
        return (3.0 / 5.0) * gamma ** 2 if self.kernel_type == 'Ma5f2' else 0.5 * gamma ** 2

    def calc_nugget_Kbase(self, n_eval, cond_max=None):
        # GpWellCond.py:109-114
        if cond_max is None:
            cond_max = self.cond_max_target
        return n_eval / (cond_max - 1)

    def calc_nugget(self, n_eval):
        # GpWellCond.py:116-154
        if self.cond_eta_set_mtd == 'dflt_eta':
            return self.cond_eta_dflt, self.cond_eta_dflt
        eta_Kbase = self.calc_nugget_Kbase(n_eval)
        if not self.use_grad:
            return eta_Kbase, np.nan
        if n_eval == 1:
            return eta_Kbase, eta_Kbase
        if self.wellcond_mtd == 'precon':
            dim = self.dim
            if self.kernel_type == 'SqExp' or self.kernel_type == 'RatQu':        # GpWellCond.py:129
                ub = 0.5 * (n_eval - 1) * (1 + np.sqrt(1 + 4 * dim)) * np.exp(-(1 + 2 * dim - np.sqrt(1 + 4 * dim)) / (4 * dim))
            else:
                al = (np.sqrt(3 * dim) - 1 + np.sqrt(15 * dim + 2 * np.sqrt(3 * dim) + 1)) / (2 * (3 * dim + np.sqrt(3 * dim)))
                ub = (n_eval - 1) * (1 + (dim + np.sqrt(3 * dim)) * al + dim * (1 + np.sqrt(3 * dim)) * al ** 2) \
                    * np.exp(-np.sqrt(3 * dim) * al)
            return eta_Kbase, (1 + ub) / (self.cond_max_target - 1)
        if 'rescale' in self.wellcond_mtd:
            return eta_Kbase, self.calc_nugget_Kfull_vreq(n_eval)                 # GpWellCond.py:140-141
        if self.cond_eta_set_mtd == 'Kbase_eta':
            return eta_Kbase, eta_Kbase
        if self.cond_eta_set_mtd == 'Kbase_eta_w_dim':
            return eta_Kbase, eta_Kbase * (self.dim + 1)
        raise Exception(f'Uknown method for cond_eta_set_mtd = {self.cond_eta_set_mtd}')

    @staticmethod
    def calc_grad_precon_matrix(n_eval, n_grad, gamma_grad_theta, b_return_vec):
        """d pvec / d theta (KernelCommon.py:13-49): [n_data, dim] vectors or [dim, n_data, n_data] diagonal matrices; entry
        (n_eval + i n_grad + a, i) = d gamma_i / d theta_i."""
        dim = gamma_grad_theta.size
        n_data = n_eval + n_grad * dim
        cols = np.zeros((n_data, dim))
        for i in range(dim):
            cols[n_eval + i * n_grad:n_eval + (i + 1) * n_grad, i] = gamma_grad_theta[i]
        if b_return_vec:
            return cols
        return np.stack([np.diag(cols[:, i]) for i in range(dim)])

    def calc_Kern_precon(self, n_eval, n_grad, theta, calc_grad=False, b_return_vec=False):
        """The preconditioner of the noise-free kernel matrix and its derivative with respect to theta -- sq_exp_Kern_precon
        (KernelSqExp.py:590-605), matern_5f2_Kern_precon (KernelMatern5f2.py:664-679), rat_quad_Kern_precon
        (KernelRatQuad.py:862-877): pvec = [1_n, gamma_1 1_ng, ..., gamma_d 1_ng].  Host arithmetic on O(N) numbers; the
        device builds the same vector (with the noise of the data) inside its assembly."""
        gamma = self.theta2gamma(np.asarray(theta, dtype=float))
        pvec = np.hstack((np.ones(n_eval), np.kron(gamma, np.ones(n_grad))))
        gamma_grad_theta = 5.0 / (6.0 * gamma) if self.kernel_type == 'Ma5f2' else 1 / gamma
        grad_precon = self.calc_grad_precon_matrix(n_eval, n_grad, gamma_grad_theta, b_return_vec)
        if b_return_vec:
            return pvec, 1 / pvec, grad_precon
        return np.diag(pvec), np.diag(1 / pvec), grad_precon

    # ---- rescaling method (GpWellCond.py:18-100; "A Non-intrusive Solution to the Ill-Conditioning Problem of the
    #      Gradient-Enhanced Gaussian Covariance Matrix for Gaussian Processes") ---------------------------------------
    calc_dist_min = staticmethod(calc_dist_min)                                   # CommonFun.py:16-34
    calc_dist_max = staticmethod(calc_dist_max)                                   # CommonFun.py:36-54

    def calc_mtd_rescale_origin_vreq(self, n_eval, dim=None):
        """Minimum distance between scaled points that bounds the condition number with the constant nugget
        (GpWellCond.py:26-41): min(2 sqrt(d), (2 + sqrt(4 + 2 e^2 ln((n - 1)(1 + 2 sqrt(d)) / 2))) / e)."""
        if dim is None:
            dim = self.dim
        if n_eval == 1:
            return 1
        dist_star = 2 * np.sqrt(dim)
        root = np.sqrt(4 + 2 * np.exp(2) * np.log((n_eval - 1) * (1 + dist_star) / 2))
        return np.minimum((2 + root) / np.exp(1), dist_star)

    def calc_nugget_Kfull_vreq(self, n_eval, vmin=None):
        """Nugget that goes with that distance (GpWellCond.py:78-100)."""
        if vmin is None:
            vmin = self.calc_mtd_rescale_origin_vreq(n_eval)
        cond_max = self.cond_max_target
        if n_eval == 1:
            return n_eval / (cond_max - 1)
        assert vmin >= np.sqrt(2), f'This method requires that vmin = {vmin} >= sqrt(2)'
        v_frac = 2 * np.sqrt(self.dim) / vmin
        eta_Kgrad = (1 + (n_eval - 1) * v_frac * np.exp(1 / v_frac - 1)) / (cond_max - 1)
        eta_Kbase = self.calc_nugget_Kbase(n_eval, cond_max)
        assert v_frac >= 0.99, f'This term should be greater or equal to 1, v_frac = {v_frac}'
        assert eta_Kgrad >= 0.99 * eta_Kbase, \
            f'We expect that eta_Kgrad > eta_Kbase but eta_Kgrad = {eta_Kgrad}, eta_Kbase = {eta_Kbase}'
        return eta_Kgrad

    def rescaling_data_w_theta_sol(self, X_scl_v1, xvec_scale_v1, hp_theta, tol_min_dist_x=1e-15):
        """From an optimised theta to the next anisotropic scaling (GpWellCond.py:43-76): scale direction k by
        sqrt(theta_k / theta*) with theta* the geometric mean, restore the required minimum distance, and return the
        isotropic theta that solution corresponds to, its squared log-distance from the line theta_1 = ... = theta_d, and
        the new scale vector."""
        n_eval = X_scl_v1.shape[0]
        assert n_eval > 1, 'This method should only be called if n_eval > 1'
        if self.optz_log_hp_theta:
            theta_sol, log_theta = 10 ** hp_theta, hp_theta
        else:
            theta_sol, log_theta = hp_theta, np.log10(hp_theta)
        vreq = self.calc_mtd_rescale_origin_vreq(n_eval, self.dim)
        theta_star = 10 ** np.mean(log_theta)
        scale_v2 = np.sqrt(theta_sol / theta_star)
        correction = vreq / np.max((calc_dist_min(X_scl_v1 * scale_v2[None, :]), tol_min_dist_x))
        xvec_scale_new = xvec_scale_v1 * scale_v2 * correction
        dist2th_star = np.dot(log_theta, log_theta) - np.dot(log_theta, np.ones(self.dim)) ** 2 / self.dim
        theta_est = np.ones(self.dim) * theta_star / correction ** 2
        return (np.log10(theta_est) if self.optz_log_hp_theta else theta_est), dist2th_star, xvec_scale_new

    @staticmethod
    def make_data_vec(fval, fgrad=None):
        # CommonFun.py:151-173
        if fgrad is None:
            return np.atleast_1d(fval)
        return np.hstack((fval, fgrad.reshape(fgrad.size, order='f')))

    def make_hp_class(self, beta=None, theta=None, kernel=None, varK=None, var_fval=None, var_fgrad=None):
        return HparaOptzVal(beta, theta, kernel, varK, var_fval, var_fgrad)     # GpHpara.py:28-31

    def set_custom_hp(self, beta=None, theta=None, kernel=None, varK=None, var_fval=None, var_fgrad=None):
        if varK is not None:
            assert varK > 0, f'varK must be positive but it is {varK}'            # GpHpara.py:105-116
        self.hp_vals = self.make_hp_class(beta, theta, kernel, varK, var_fval, var_fgrad)

    def set_hp_optz_info(self, has_theta, has_kernel=False, has_varK=False, has_var_fval=False, has_var_fgrad=False):
        # GpHparaOptz.py:44-138
        n_hp = has_theta * self.dim + has_kernel + has_varK + has_var_fval + has_var_fgrad
        bvec = np.zeros(n_hp, dtype=bool)
        cnt = 0
        empty = np.array([], dtype=int)
        idx_theta = idx_kernel = idx_varK = idx_var_fval = idx_var_fgrad = empty
        if has_theta:
            idx_theta = np.arange(cnt, cnt + self.dim, dtype=int)
            cnt += self.dim
            if self.optz_log_hp_theta:
                bvec[idx_theta] = 1
        if has_kernel:
            assert self.kernel_has_hp, 'Kernel must have hyperaparamters if b_optz_hp_kernel is set to True'
            idx_kernel = np.array([cnt])                                        # GpHparaOptz.py:90-96
            cnt += 1
            if self.optz_log_hp_kernel:
                bvec[idx_kernel] = 1
        if has_varK:
            idx_varK = cnt
            cnt += 1
            if self.optz_log_hp_var:
                bvec[idx_varK] = 1
        if has_var_fval:
            assert self.known_eps_fval is False, 'var_fval should not be a hyperparameter if known_eps_fval is True'
            idx_var_fval = cnt
            cnt += 1
            if self.optz_log_hp_var:
                bvec[idx_var_fval] = 1
        if has_var_fgrad:
            assert self.known_eps_fgrad is False, 'var_fgrad should not be a hyperparameter if known_eps_fgrad is True'
            idx_var_fgrad = cnt
            cnt += 1
            if self.optz_log_hp_var:
                bvec[idx_var_fgrad] = 1
        return HparaOptzInfo(n_hp=n_hp, has_theta=has_theta, idx_theta=idx_theta, has_kernel=has_kernel,
                             idx_kernel=idx_kernel, has_varK=has_varK, idx_varK=idx_varK,
                             has_var_fval=has_var_fval, idx_var_fval=idx_var_fval, has_var_fgrad=has_var_fgrad,
                             idx_var_fgrad=idx_var_fgrad, bvec_log_optz=bvec)

    def setup_hp_idx4optz(self):
        # GpHparaOptz.py:187-196
        self.hp_info_optz_lkd = self.set_hp_optz_info(True, self.b_optz_hp_kernel and self.kernel_has_hp,
                                                      self.b_has_noisy_data, self.b_optz_var_fval,
                                                      self.b_optz_var_fgrad)

    def hp_vec2dataclass(self, hp_optz_info, hp_vec):
        # GpHpara.py:56-103
        v = np.copy(hp_vec)
        b = hp_optz_info.bvec_log_optz
        v[b] = 10 ** (v[b])
        theta = v[hp_optz_info.idx_theta] if hp_optz_info.has_theta else None
        hp_kernel = v[hp_optz_info.idx_kernel] if hp_optz_info.has_kernel else None     # GpHpara.py:80-83
        varK = var_fval = var_fgrad = None
        if hp_optz_info.has_varK:
            assert self.b_has_noisy_data
            varK = v[hp_optz_info.idx_varK]
        if hp_optz_info.has_var_fval:
            assert self.b_optz_var_fval
            var_fval = v[hp_optz_info.idx_var_fval]
        if hp_optz_info.has_var_fgrad:
            assert self.b_optz_var_fgrad
            var_fgrad = v[hp_optz_info.idx_var_fgrad]
        return self.make_hp_class(None, theta, hp_kernel, varK, var_fval, var_fgrad)

    # ---- data ingest ---------------------------------------------------------------------------------
    def set_data(self, x_eval, fval, std_fval, grad=None, std_grad=None, bvec_use_grad=None):
        # reference GaussianProcess.py:219-363
        n_eval = fval.size
        if self.use_grad:
            if bvec_use_grad is None:
                n_grad = n_eval
            else:
                bvec_use_grad = np.asarray(bvec_use_grad, dtype=bool)
                n_grad = int(np.sum(bvec_use_grad))
                assert bvec_use_grad.size == n_eval, \
                    f'Length of bvec_use_grad is {bvec_use_grad.size} but it should be n_eval = {n_eval}'
                assert grad.shape[0] == n_grad, \
                    f'No. of rows of grad is {grad.shape[0]} but it should be n_grad = {n_grad}'
        else:
            assert bvec_use_grad is None, 'bvec_use_grad must be None if grads are not used for the GP'
            n_grad = 0
        self.n_eval, self.n_grad = n_eval, n_grad
        self.n_data = n_eval + n_grad * self.dim
        fval = np.atleast_1d(fval).ravel()
        assert x_eval.ndim == 2, f'x_eval must be a 2 array but x_eval.ndim = {x_eval.ndim}'
        assert n_eval == fval.size, 'No. of points do not match with x_eval and fval'
        assert x_eval.shape == (n_eval, self.dim), 'Shape of x_eval does not match (n_eval, dim)'
        if (std_fval is None) or np.any(np.isnan(std_fval)):
            self.known_eps_fval = False
        else:
            self.known_eps_fval = True
            std_fval = np.atleast_1d(std_fval).ravel()
            assert n_eval == std_fval.size, f'Size of std_fval is {std_fval.size} while it should be {n_eval}'
        if grad is None:
            assert self.use_grad is False, 'No grad info provided but use_grad was set to True'
            self.has_grad_info = False
            self.known_eps_fgrad = False
        else:
            assert self.use_grad, 'Grad info provided but use_grad was set to False'
            self.has_grad_info = True
            assert grad.ndim == 2, f'grad must be a 2 array but grad.ndim = {grad.ndim}'
            assert grad.shape == (n_grad, self.dim), 'Shape of grad does not match x_eval'
            if (std_grad is None) or np.any(np.isnan(std_grad)):
                self.known_eps_fgrad = False
            else:
                self.known_eps_fgrad = True
                assert std_grad.ndim == 2, f'std_grad must be a 2 array but std_grad.ndim = {std_grad.ndim}'
                assert grad.shape == std_grad.shape, 'Shape of grad does not match std_grad'
        self._x_eval_in = x_eval
        self._fval_in = fval
        self._std_fval_in = std_fval if self.known_eps_fval else None
        self._grad_in = grad
        self._std_grad_in = std_grad if self.known_eps_fgrad else None
        self.bvec_use_grad = bvec_use_grad
        if self.known_eps_fval:
            self.b_optz_var_fval = False
            self.b_fval_zero = bool(np.max(std_fval) < 1e-10)
        else:
            self.b_optz_var_fval = True
            self.b_fval_zero = False
        if self.use_grad is False:
            self.b_optz_var_fgrad = False
            self.b_fgrad_zero = True
        elif self.known_eps_fgrad:
            self.b_optz_var_fgrad = False
            self.b_fgrad_zero = bool(np.max(std_grad) < 1e-10)
        else:
            self.b_optz_var_fgrad = True
            self.b_fgrad_zero = False
        self.b_has_noisy_data = not (self.b_fval_zero and self.b_fgrad_zero)
        self._eta_Kbase, self._eta_Kgrad = self.calc_nugget(self.n_eval)
        self._etaK = self._eta_Kgrad if self.use_grad else self._eta_Kbase
        self._vmin_init = calc_dist_min(x_eval)                                   # GaussianProcess.py:338
        self.setup_hp_idx4optz()
        self.DataScl = None
        if self.b_use_data_scl:                                                   # GaussianProcess.py:342-361
            if self.wellcond_mtd == 'rescale_origin':
                dist_set = self._vmin_req_grad = self.calc_mtd_rescale_origin_vreq(n_eval, self.dim)
                x_scl_method = 'set_vmin'
            elif self.wellcond_mtd == 'rescale_eta_vary':
                dist_set, x_scl_method = self.vmin_rescale_eta_vary, 'set_vmin'
            elif self.wellcond_mtd == 'dflt_vmin':
                dist_set, x_scl_method = self.cond_dist_min_dflt, 'set_vmin'
            elif self.wellcond_mtd == 'dflt_vmax':
                dist_set, x_scl_method = self.cond_dist_max_dflt, 'set_vmax'
            else:
                raise Exception(f'Unknown method wellcond_mtd = {self.wellcond_mtd}')
            self.DataScl = Rescaling(x_eval, x_scl_method=x_scl_method, dist_set=dist_set)
            self.DataScl.set_obj_data(fval, std_fval, grad, std_grad)
            me = weakref.ref(self)                              # (weak: DataScl must not keep the model and its device memory alive)
            self.DataScl.on_change = lambda: me() is not None and me()._on_rescale()   # a later set_xscale_data() re-sends the scaled data
        self._Rtensor_init = None       # the [d, n, n] tensor (GaussianProcess.py:363): not built by the hot path, see Rtensor_init
        self.KernEta_chofac = None
        self.invKernEta_fdiff = None
        self._eval_ready = False
        self._push_data()

    def _err(self):
        return self._lib.gpg_last_error(self._ctx).decode()

    def _push_data(self):
        shape = (self.n_eval, self.dim, bool(self.use_grad), self.kernel_type)
        if self._ctx is None or self._ctx_shape != shape:
            if self._ctx is not None:
                self._lib.gpg_destroy(self._ctx)
                self._ctx = None
            ctx = C.c_void_p()
            rc = self._lib.gpg_create(C.byref(ctx), self.device, self.n_eval, self.dim, int(self.use_grad),
                                      _lib.GPG_KERNEL[self.kernel_type])
            if rc != 0:
                raise _lib.GpgError(f'gpg_create failed ({rc}): {self._lib.gpg_last_error(None).decode()}')
            self._ctx, self._ctx_shape = ctx, shape
        if self.use_grad:
            mask = None if self.bvec_use_grad is None else np.ascontiguousarray(self.bvec_use_grad, dtype=np.uint8)
            rc = self._lib.gpg_set_grad_mask(self._ctx, None if mask is None else mask.ctypes.data_as(C.POINTER(C.c_ubyte)))
            if rc != 0:
                raise _lib.GpgError(f'gpg_set_grad_mask failed ({rc}): {self._err()}')
        x_scl = self.get_scl_x_w_dist(want_tensor=False)[0]
        fval_scl, std_fval_scl, grad_scl, std_grad_scl = self.get_scl_eval_data()
        self._fval_scl = fval_scl
        x = np.ascontiguousarray(x_scl, dtype=np.float64)
        y = np.ascontiguousarray(self.make_data_vec(fval_scl, grad_scl if self.use_grad else None), dtype=np.float64)
        noise = np.zeros(self.n_data)
        if self.b_has_noisy_data:                                                 # Kernel.py:324-353 (on the scaled data: :328)
            if self.known_eps_fval:
                noise[:self.n_eval] = std_fval_scl ** 2
            if self.use_grad and self.known_eps_fgrad:
                noise[self.n_eval:] = (std_grad_scl ** 2).reshape(std_grad_scl.size, order='f')
        self._noise_known = noise
        rc = self._lib.gpg_set_data(self._ctx, _lib.as_dp(x), _lib.as_dp(y), _lib.as_dp(noise))
        if rc != 0:
            raise _lib.GpgError(f'gpg_set_data failed ({rc}): {self._err()}')
        # the reference differentiates with the constant nugget even when the matrix gets the row-sum one (GpHparaGrad.py:43,107,126)
        self._lib.gpg_set_gradient_nugget(self._ctx, -1.0 if self.cond_eta_is_const else float(self._etaK))
        self._data_vec = y

    def _on_rescale(self):
        """DataScl.set_xscale_data / set_obj_scaling changed the scaled data (OptzLkd.py:176): the device copy and every
        factor made from the old one are stale."""
        self.KernEta_chofac = None
        self.invKernEta_fdiff = None
        self._eval_ready = False
        self._push_data()

    @property
    def Rtensor_init(self):
        """GaussianProcess.py:363: the difference tensor of the evaluation points with themselves.  The device path never
        needs it (256 MB at n = 2000, d = 8), so it is only built -- on the host, once per data set -- when a caller reads
        the attribute, e.g. to hand it to calc_KernGrad like the reference's own code does."""
        if self._Rtensor_init is None and getattr(self, '_x_eval_in', None) is not None:
            self._Rtensor_init = self.calc_Rtensor(self._x_eval_in, self._x_eval_in, 1)
        return self._Rtensor_init

    def get_scl_x_w_dist(self, want_tensor=True):
        """GaussianProcess.py:399-404: (scaled points, their [dim, n, n] difference tensor).  The tensor is only built when
        it is asked for (want_tensor=False is this package's own use)."""
        if self.b_use_data_scl:
            return (self.DataScl.x_scl, self.DataScl.Rtensor_scl if want_tensor else None)
        return self._x_eval_in, (self.Rtensor_init if want_tensor else None)

    def x_init_2_scl(self, x_init):
        return self.DataScl.x_init_2_scl(x_init) if self.b_use_data_scl else x_init      # GaussianProcess.py:406-411

    def x_scl_2_init(self, x_scl):
        return self.DataScl.x_scl_2_init(x_scl) if self.b_use_data_scl else x_scl        # GaussianProcess.py:413-418

    def get_init_eval_data(self):
        return self._fval_in, self._std_fval_in, self._grad_in, self._std_grad_in         # GaussianProcess.py:420-421

    def get_scl_eval_data(self):
        # GaussianProcess.py:423-433
        fval, std_fval, grad, std_grad = self.data_init_2_scl(*self.get_init_eval_data())[:4]
        return fval, (std_fval if self.known_eps_fval else None), grad, (std_grad if self.known_eps_fgrad else None)

    def data_init_2_scl(self, mu_in=None, sig_in=None, dmudx_in=None, dsigdx_in=None, d2mudx2_in=None, d2sigdx2_in=None):
        if self.b_use_data_scl:                                                           # GaussianProcess.py:435-445
            return self.DataScl.obj_init_2_scl(mu_in, sig_in, dmudx_in, dsigdx_in, d2mudx2_in, d2sigdx2_in)
        return mu_in, sig_in, dmudx_in, dsigdx_in, d2mudx2_in, d2sigdx2_in

    def data_scl_2_init(self, mu_scl=None, sig_scl=None, dmudx_scl=None, dsigdx_scl=None, d2mudx2_scl=None, d2sigdx2_scl=None):
        if self.b_use_data_scl:                                                           # GaussianProcess.py:447-457
            return self.DataScl.obj_scl_2_init(mu_scl, sig_scl, dmudx_scl, dsigdx_scl, d2mudx2_scl, d2sigdx2_scl)
        return mu_scl, sig_scl, dmudx_scl, dsigdx_scl, d2mudx2_scl, d2sigdx2_scl

    def calc_noise_vec(self, hp_vals):
        # Kernel.py:309-357 (host copy; the device builds the same vector from var_fval / var_fgrad)
        if self.b_fval_zero and self.b_fgrad_zero:
            return np.zeros(self.n_data)
        std_fval, _, std_fgrad = self.get_scl_eval_data()[1:]                    # Kernel.py:328
        out = np.zeros(self.n_data)
        if self.known_eps_fval:
            assert hp_vals.var_fval is None
            out[:self.n_eval] = std_fval ** 2
        else:
            out[:self.n_eval] = hp_vals.var_fval
        if self.use_grad:
            if self.known_eps_fgrad:
                assert hp_vals.var_fgrad is None
                out[self.n_eval:] = (std_fgrad ** 2).reshape(std_fgrad.size, order='f')
            else:
                out[self.n_eval:] = hp_vals.var_fgrad
        return out

    # ---- C-ABI argument packing ------------------------------------------------------------------------
    def _make_hp(self, hp_vals, varK_mat, closed_form):
        theta = np.ascontiguousarray(hp_vals.theta, dtype=np.float64)
        assert theta.size == self.dim, 'theta must have dim entries'
        assert np.sum(np.isnan(theta)) == 0, f'There are nan values theta = {theta}'     # Kernel.py:201
        hp = _lib.GpgHp()
        hp.theta = _lib.as_dp(theta)
        hp.varK_mat = float(varK_mat)
        hp.var_fval = -1.0 if (self.known_eps_fval or not self.b_has_noisy_data) else float(hp_vals.var_fval)
        if not self.use_grad:
            hp.var_fgrad = -1.0
        else:
            hp.var_fgrad = -1.0 if (self.known_eps_fgrad or not self.b_has_noisy_data) else float(hp_vals.var_fgrad)
        if self._noise_override:                                 # per-row vector on the device (calc_all_K_w_chofac(noise_vec=...))
            hp.var_fval = hp.var_fgrad = -1.0
        hp.eta = float(self._etaK)
        hp.wellcond = self._wellcond_code
        hp.closed_form_varK = int(closed_form)
        if self.kernel_has_hp:
            assert hp_vals.kernel is not None, 'hp_vals.kernel (alpha of RatQu) must be set'
            hp.hp_kernel = float(np.asarray(hp_vals.kernel).reshape(-1)[0])
        else:
            hp.hp_kernel = 0.0
        self._etaK_last, self._idx_etaK_argmax_last = self._etaK, None
        if not self.cond_eta_is_const:
            # Kernel.py:229-236 / 269-276: nugget from the largest absolute row sum of Kcor ('precon') or of the kernel matrix
            rowsum = np.empty(self.n_data)
            rc = self._lib.gpg_abs_rowsum(self._ctx, C.byref(hp), _lib.as_dp(rowsum))
            if rc != 0:
                raise _lib.GpgError(f'gpg_abs_rowsum failed ({rc}): {self._err()}')
            idx = int(np.argmax(rowsum))
            hp.eta = float(rowsum[idx] / (self.cond_max_target - 1))
            self._etaK_last, self._idx_etaK_argmax_last = hp.eta, idx
        return hp, theta   # keep theta alive

    @property
    def _wellcond_code(self):
        """The two matrix constructions of Kernel.py:220-302: 'precon', or -- for every other method -- nugget only."""
        return _lib.GPG_WELLCOND['precon' if self.wellcond_mtd == 'precon' else 'base']

    def _rows_with_eta(self, rows):
        """cond_eta_is_const = False: the batched calls take the nugget of every row in an extra column (include/gpgrad.h)."""
        if self.cond_eta_is_const:
            return rows
        out = np.empty((rows.shape[0], rows.shape[1] + 1))
        out[:, :-1] = rows
        d = self.dim
        for i, r in enumerate(rows):
            hp_vals = HparaOptzVal(None, r[:d], r[d + 3] if self.kernel_has_hp else None, r[d],
                                   r[d + 1] if r[d + 1] >= 0 else None, r[d + 2] if r[d + 2] >= 0 else None)
            out[i, -1] = self._make_hp(hp_vals, r[d], closed_form=not self.b_has_noisy_data)[0].eta
        return np.ascontiguousarray(out)

    # ---- kernel table (Kernel.py:27-126: bound per kernel type in the reference; same names and arguments here) ------
    @staticmethod
    def calc_Rtensor(X, Y, exp=1):
        """CommonFun.py:56-84: R[k, a, b] = X[a, k] - Y[b, k] (the reference's body does not use `exp` either)."""
        X, Y = np.asarray(X, dtype=np.float64), np.asarray(Y, dtype=np.float64)
        assert X.shape[1] == Y.shape[1], 'The dimensions of the arrays do not match'
        return np.ascontiguousarray(X.T[:, :, None] - Y.T[:, None, :])

    def _kern_from_rtensor(self, Rtensor, theta, hp_kernel, use_grad, bvec1=None, bvec2=None):
        Rtensor = np.ascontiguousarray(Rtensor, dtype=np.float64)
        assert Rtensor.ndim == 3, 'Rtensor must have the shape [dim, n1, n2]'
        dim, n1, n2 = Rtensor.shape
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        assert theta.size == dim, 'theta must have dim entries'
        m1 = m2 = None
        n1g, n2g = n1, n2
        if use_grad:
            if bvec1 is not None:
                m1 = np.ascontiguousarray(bvec1, dtype=np.uint8)
                assert m1.size == n1
                n1g = int(m1.sum())
            if bvec2 is not None:
                m2 = np.ascontiguousarray(bvec2, dtype=np.uint8)
                assert m2.size == n2
                n2g = int(m2.sum())
        shape = (n1 + n1g * dim, n2 + n2g * dim) if use_grad else (n1, n2)
        out = np.empty(shape)
        alpha = 0.0
        if self.kernel_has_hp:
            assert hp_kernel is not None and not np.isnan(hp_kernel), 'this kernel needs its hyperparameter hp_kernel'
            alpha = float(np.asarray(hp_kernel).reshape(-1)[0])
        ub = C.POINTER(C.c_ubyte)
        rc = self._lib.gpg_kern_rtensor(self.device, _lib.GPG_KERNEL[self.kernel_type], dim, n1, n2, _lib.as_dp(Rtensor),
                                        _lib.as_dp(theta), alpha, int(use_grad),
                                        None if m1 is None else m1.ctypes.data_as(ub), None if m2 is None else m2.ctypes.data_as(ub),
                                        _lib.as_dp(out))
        if rc != 0:
            raise _lib.GpgError(f'gpg_kern_rtensor failed ({rc}): {self._lib.gpg_last_error(None).decode()}')
        return out

    def calc_KernBase(self, Rtensor, theta, hp_kernel=None):
        """Gradient-free kernel matrix [n1, n2] from the difference tensor -- sq_exp_calc_KernBase (KernelSqExp.py:16-46),
        matern_5f2_calc_KernBase (KernelMatern5f2.py:16-52), rat_quad_calc_KernBase (KernelRatQuad.py:439-476), on the device."""
        return self._kern_from_rtensor(Rtensor, theta, hp_kernel, False)

    def calc_KernGrad(self, Rtensor, theta, hp_kernel=None, bvec_use_grad1=None, bvec_use_grad2=None):
        """Gradient-enhanced kernel matrix [n1 + n1g d, n2 + n2g d] -- sq_exp_calc_KernGrad (KernelSqExp.py:320-410),
        matern_5f2_calc_KernGrad (KernelMatern5f2.py:352-450), rat_quad_calc_KernGrad (KernelRatQuad.py:478-554), on the device."""
        return self._kern_from_rtensor(Rtensor, theta, hp_kernel, True, bvec_use_grad1, bvec_use_grad2)

    def calc_Kern(self, Rtensor, theta, hp_kernel=None, *masks):
        """Kernel.py:116,122: calc_KernGrad when the model uses gradients, calc_KernBase otherwise."""
        return self.calc_KernGrad(Rtensor, theta, hp_kernel, *masks) if self.use_grad else self.calc_KernBase(Rtensor, theta, hp_kernel)

    # ---- derivative tensors of the kernel table (Kernel.py:43-48, 69-75, 96-102, 119-125) and their compositions
    #      (optz/GpHparaGrad.py): materialised for callers that want them; the likelihood gradient never forms them -----------
    def _kern_grad_hp_tensors(self, Rtensor, theta, hp_kernel, use_grad, want_theta, want_alpha):
        Rtensor = np.ascontiguousarray(Rtensor, dtype=np.float64)
        assert Rtensor.ndim == 3, 'Rtensor must have the shape [dim, n1, n2]'
        dim, n1, n2 = Rtensor.shape
        assert n1 == n2, 'Incompatible shapes'                                   # KernelSqExp.py:500
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        assert theta.size == dim, 'theta must have dim entries'
        N = n1 * (dim + 1) if use_grad else n1
        out_th = np.empty((dim, N, N)) if want_theta else None
        out_al = np.empty((1, N, N)) if want_alpha else None
        alpha = 0.0
        if self.kernel_has_hp:
            assert hp_kernel is not None and not np.isnan(hp_kernel), 'this kernel needs its hyperparameter hp_kernel'
            alpha = float(np.asarray(hp_kernel).reshape(-1)[0])
        rc = self._lib.gpg_kern_rtensor_grad_hp(self.device, _lib.GPG_KERNEL[self.kernel_type], dim, n1, _lib.as_dp(Rtensor),
                                                _lib.as_dp(theta), alpha, int(use_grad),
                                                None if out_th is None else _lib.as_dp(out_th),
                                                None if out_al is None else _lib.as_dp(out_al))
        if rc != 0:
            raise _lib.GpgError(f'gpg_kern_rtensor_grad_hp failed ({rc}): {self._lib.gpg_last_error(None).decode()}')
        return out_th, out_al

    def _no_kernel_hp(self):
        name = {'SqExp': 'squared exponential', 'Ma5f2': 'Matern 5/2'}[self.kernel_type]
        raise Exception(f'There are no kernel hyperparameters for the {name} kernel')        # KernelSqExp.py:126-127, KernelMatern5f2.py:138-139

    @staticmethod
    def _full_mask_only(bvec_use_grad):
        if bvec_use_grad is not None and not np.all(bvec_use_grad):
            # the reference's own arrays do not fit together with a mask (KernelSqExp.py:552-554): nothing to reproduce
            raise NotImplementedError('hyperparameter derivatives of the kernel matrix with a bvec_use_grad mask are not supported')

    def calc_KernBase_grad_th(self, Rtensor, theta, hp_kernel=None, *args):
        """d KernBase / d theta_k, [dim, n, n] (sq_exp / matern_5f2 / rat_quad _calc_KernBase_grad_th), on the device.  Matern-5/2:
        the exact derivative (the reference's differentiates the exponential factor only, see tests/tolerances.py)."""
        return self._kern_grad_hp_tensors(Rtensor, theta, hp_kernel, False, True, False)[0]

    def calc_KernGrad_grad_th(self, Rtensor, theta, hp_kernel=None, bvec_use_grad=None):
        """d KernGrad / d theta_k, [dim, n (dim + 1), n (dim + 1)] (KernelSqExp.py:470-568, KernelMatern5f2.py:532-642,
        KernelRatQuad.py:640-750), on the device."""
        self._full_mask_only(bvec_use_grad)
        return self._kern_grad_hp_tensors(Rtensor, theta, hp_kernel, True, True, False)[0]

    def calc_KernBase_grad_alpha(self, Rtensor, theta, hp_kernel=None, *args):
        """d KernBase / d alpha, [1, n, n], for the rational quadratic kernel (KernelRatQuad.py:133-163); the others have none."""
        if not self.kernel_has_hp:
            self._no_kernel_hp()
        return self._kern_grad_hp_tensors(Rtensor, theta, hp_kernel, False, False, True)[1]

    def calc_KernGrad_grad_alpha(self, Rtensor, theta, hp_kernel=None, bvec_use_grad=None):
        """d KernGrad / d alpha, [1, N, N], for the rational quadratic kernel (KernelRatQuad.py:752-840)."""
        if not self.kernel_has_hp:
            self._no_kernel_hp()
        self._full_mask_only(bvec_use_grad)
        return self._kern_grad_hp_tensors(Rtensor, theta, hp_kernel, True, False, True)[1]

    def calc_Kern_grad_theta(self, Rtensor, theta, hp_kernel=None, *args):
        """Kernel.py:119-125: the gradient-enhanced or the gradient-free variant, as the model uses gradients or not."""
        return self.calc_KernGrad_grad_th(Rtensor, theta, hp_kernel, *args) if self.use_grad else self.calc_KernBase_grad_th(Rtensor, theta, hp_kernel)

    def calc_Kern_grad_alpha(self, Rtensor, theta, hp_kernel=None, *args):
        return self.calc_KernGrad_grad_alpha(Rtensor, theta, hp_kernel, *args) if self.use_grad else self.calc_KernBase_grad_alpha(Rtensor, theta, hp_kernel)

    def _kern_hess_x(self, Rtensor, theta, hp_kernel, use_grad, bvec_use_grad2=None):
        Rtensor = np.ascontiguousarray(Rtensor, dtype=np.float64)
        assert Rtensor.ndim == 3, 'Rtensor must have the shape [dim, n1, n2]'
        dim, n1, n2 = Rtensor.shape
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        assert theta.size == dim, 'theta must have dim entries'
        m2, n2g = None, n2
        if use_grad and bvec_use_grad2 is not None:
            m2 = np.ascontiguousarray(bvec_use_grad2, dtype=np.uint8)
            assert m2.size == n2
            n2g = int(m2.sum())
        out = np.empty((dim, n1 * dim, n2 + n2g * dim if use_grad else n2))
        alpha = 0.0
        if self.kernel_has_hp:
            assert hp_kernel is not None and not np.isnan(hp_kernel), 'this kernel needs its hyperparameter hp_kernel'
            alpha = float(np.asarray(hp_kernel).reshape(-1)[0])
        rc = self._lib.gpg_kern_rtensor_hess_x(self.device, _lib.GPG_KERNEL[self.kernel_type], dim, n1, n2, _lib.as_dp(Rtensor),
                                               _lib.as_dp(theta), alpha, int(use_grad),
                                               None if m2 is None else m2.ctypes.data_as(C.POINTER(C.c_ubyte)), _lib.as_dp(out))
        if rc != 0:
            raise _lib.GpgError(f'gpg_kern_rtensor_hess_x failed ({rc}): {self._lib.gpg_last_error(None).decode()}')
        return out

    def calc_KernBase_hess_x(self, Rtensor, theta, hp_kernel=None, *args):
        """d2 KernBase / d x1 d x1, [dim, n1 dim, n2] (KernelSqExp.py:48-88, KernelMatern5f2.py:53-97, KernelRatQuad.py:51-131)."""
        return self._kern_hess_x(Rtensor, theta, hp_kernel, False)

    def calc_KernGrad_grad_x(self, Rtensor, theta, hp_kernel=None, bvec_use_grad2=None):
        """Derivative of the gradient-enhanced cross kernel with respect to the first point set, [dim, n1 dim, n2 + n2g dim]
        (KernelSqExp.py:412-468, KernelMatern5f2.py:452-530, KernelRatQuad.py:556-638)."""
        return self._kern_hess_x(Rtensor, theta, hp_kernel, True, bvec_use_grad2)

    def calc_Kern_hess_x(self, Rtensor, theta, hp_kernel=None, *args):
        """Kernel.py:117,123: what eval_model's Hessians are made of (GpEvalModel.py:148-150)."""
        return self.calc_KernGrad_grad_x(Rtensor, theta, hp_kernel, *args) if self.use_grad else self.calc_KernBase_hess_x(Rtensor, theta, hp_kernel)

    def calc_KernGrad_hp(self, hp_optz_info, hp_vals, Rtensor, etaK=None):
        """GpHparaGrad.py:13-68: derivative of the regularised kernel matrix of the noise-free path with respect to the optimised
        hyperparameters, [n_hp, N, N]; 'precon' adds 2 eta gamma_i d gamma_i / d theta_i on the diagonal of the gradient blocks."""
        assert hp_optz_info.has_varK is False, 'If has_varK is True, use, calc_Kcov_grad_hp()'
        assert hp_optz_info.has_var_fval is False, 'If has_var_fval is True, use, calc_Kcov_grad_hp()'
        assert hp_optz_info.has_var_fgrad is False, 'If has_var_fgrad is True, use, calc_Kcov_grad_hp()'
        out = np.zeros((hp_optz_info.n_hp, self.n_data, self.n_data))
        if hp_optz_info.has_theta:
            out[hp_optz_info.idx_theta] = self.calc_Kern_grad_theta(Rtensor, hp_vals.theta, hp_vals.kernel, self.bvec_use_grad)
            if self.wellcond_mtd == 'precon':
                if etaK is None:
                    etaK = self._etaK
                pvec, _, grad_precon = self.calc_Kern_precon(self.n_eval, self.n_grad, hp_vals.theta, calc_grad=True, b_return_vec=True)
                add = (2 * etaK) * pvec[self.n_eval:, None] * grad_precon[self.n_eval:, :]
                idx = np.arange(self.n_eval, self.n_data)
                for i in range(self.dim):
                    out[hp_optz_info.idx_theta[i], idx, idx] += add[:, i]
        if hp_optz_info.has_kernel:
            out[hp_optz_info.idx_kernel] = self.calc_Kern_grad_alpha(Rtensor, hp_vals.theta, hp_vals.kernel, self.bvec_use_grad)
        return out

    def _plus_eta_diag(self, T):
        """'precon': K -> K + eta diag(diag(K)) for every slice of a [m, N, N] derivative (GpHparaGrad.py:105-110, 119-126)."""
        if self.wellcond_mtd == 'precon':
            idx = np.arange(T.shape[1])
            T[:, idx, idx] *= 1.0 + self._etaK
        return T

    def calc_Kcov_grad_theta(self, hp_vals, Rtensor):
        varK = self.hp_varK if hp_vals.varK is None else hp_vals.varK                    # GpHparaGrad.py:100-111
        return self._plus_eta_diag(varK * self.calc_Kern_grad_theta(Rtensor, hp_vals.theta, hp_vals.kernel, self.bvec_use_grad))

    def calc_Kcov_grad_alpha(self, hp_vals, Rtensor):
        varK = self.hp_varK if hp_vals.varK is None else hp_vals.varK                    # GpHparaGrad.py:113-126
        return self._plus_eta_diag(varK * self.calc_Kern_grad_alpha(Rtensor, hp_vals.theta, hp_vals.kernel, self.bvec_use_grad))

    def calc_Kcov_grad_varK(self, hp_vals, Kern):
        eta = self._etaK                                                                 # GpHparaGrad.py:128-137
        return Kern + eta * (np.diag(np.diag(Kern)) if self.wellcond_mtd == 'precon' else np.eye(Kern.shape[0]))

    def calc_Kcov_grad_var_fval(self, hp_vals):
        scalar = 1 + self._etaK if self.wellcond_mtd == 'precon' else 1                  # GpHparaGrad.py:139-146
        return scalar * np.diag(np.hstack((np.ones(self.n_eval), np.zeros(self.n_grad * self.dim))))

    def calc_Kcov_grad_var_fgrad(self, hp_vals):
        scalar = 1 + self._etaK if self.wellcond_mtd == 'precon' else 1                  # GpHparaGrad.py:148-155
        return scalar * np.diag(np.hstack((np.zeros(self.n_eval), np.ones(self.n_grad * self.dim))))

    def calc_Kcov_grad_hp(self, hp_optz_info, hp_vals, Kern, Rtensor):
        """GpHparaGrad.py:70-98: derivative of the covariance matrix of the noisy path, [n_hp, N, N]."""
        assert hp_vals.varK is not None, 'The method calc_Kcov_grad_hp() only needs to be called if varK is a hyperparameter'
        out = np.zeros((hp_optz_info.n_hp, self.n_data, self.n_data))
        if hp_optz_info.has_theta:
            out[hp_optz_info.idx_theta] = self.calc_Kcov_grad_theta(hp_vals, Rtensor)
        if hp_optz_info.has_kernel:
            out[hp_optz_info.idx_kernel] = self.calc_Kcov_grad_alpha(hp_vals, Rtensor)
        if hp_optz_info.has_varK:
            out[hp_optz_info.idx_varK] = self.calc_Kcov_grad_varK(hp_vals, Kern)
        if hp_optz_info.has_var_fval:
            out[hp_optz_info.idx_var_fval] = self.calc_Kcov_grad_var_fval(hp_vals)
        if hp_optz_info.has_var_fgrad:
            out[hp_optz_info.idx_var_fgrad] = self.calc_Kcov_grad_var_fgrad(hp_vals)
        return out

    # ---- kernel + factorisation (compat entry points; the hot path does not materialise N x N arrays) ----
    def calc_Kern_w_chofac(self, Rtensor, hp_vals, noise_vec=None, calc_chofac=True, calc_cond=False,
                           materialize=False):
        # Kernel.py:128-138
        assert self.b_has_noisy_data is False, 'This function should not be called if there is noisy data'
        return self.calc_all_K_w_chofac(Rtensor, hp_vals, noise_vec, calc_chofac, calc_cond, varK=1,
                                        materialize=materialize)

    def calc_all_K_w_chofac(self, Rtensor, hp_vals, noise_vec=None, calc_chofac=True, calc_cond=False,
                            varK=None, b_normlz_w_varK=False, materialize=False):
        """Kernel.py:140-307.  Returns the reference's 7-tuple (Kern, Kcor, Kcov, Kcov_chofac, condK, etaK,
        idx_etaK_argmax).  Kcov_chofac is the SciPy-compatible (P L, True) pair downloaded from the device
        (lower factor for both well-conditioning methods).  Kern / Kcov are downloaded only when
        materialize=True; Kcor is never formed (None)."""
        if noise_vec is not None:
            # Kernel.py:207-208, 218: the caller's per-row noise variances instead of the model's -- sent to the device for this
            # call, the model's own vector is put back afterwards
            noise_vec = np.ascontiguousarray(noise_vec, dtype=np.float64)
            assert noise_vec.size == self.n_data, 'noise_vec must have n_data entries'
            if self._lib.gpg_set_noise(self._ctx, _lib.as_dp(noise_vec)) != 0:
                raise _lib.GpgError(f'gpg_set_noise failed: {self._err()}')
            try:
                self._noise_override = True
                return self.calc_all_K_w_chofac(Rtensor, hp_vals, None, calc_chofac, calc_cond, varK, b_normlz_w_varK, materialize)
            finally:
                self._noise_override = False
                self._lib.gpg_set_noise(self._ctx, _lib.as_dp(self._noise_known))
        if varK is None:
            assert hp_vals.varK is not None, f'varK is not provided and hp_vals.varK is None, hp_vals = {hp_vals}'
            varK = hp_vals.varK
        if b_normlz_w_varK:
            varK = 1.0
        else:
            assert varK > 0, f'varK must be positive but varK = {varK}'
        hp, keep = self._make_hp(hp_vals, varK, closed_form=not self.b_has_noisy_data)
        N = self.n_data
        Kern = Kcov = None
        if materialize:
            Kern = np.empty((N, N))
            Kcov = np.empty((N, N))
            for which, out in ((0, Kern), (1, Kcov)):
                rc = self._lib.gpg_get_matrix(self._ctx, C.byref(hp), which, _lib.as_dp(out))
                if rc != 0:
                    raise _lib.GpgError(f'gpg_get_matrix failed ({rc}): {self._err()}')
        chofac = None
        t0 = time.time()
        if calc_chofac:
            out = _lib.GpgLkdOut()
            rc = self._lib.gpg_lkd(self._ctx, C.byref(hp), C.byref(out))
            if rc < 0:
                raise _lib.GpgError(f'gpg_lkd failed ({rc}): {self._err()}')
            if rc == 0:
                fac = np.empty((N, N))
                rc2 = self._lib.gpg_get_matrix(self._ctx, None, 3, _lib.as_dp(fac))
                if rc2 != 0:
                    raise _lib.GpgError(f'gpg_get_matrix failed ({rc2}): {self._err()}')
                chofac = (fac, True)
            else:
                print(f'Failure of the Cholesky decomposition, first non-positive pivot = {rc}')   # Kernel.py:255
        self._time_chofac += time.time() - t0
        condK = None
        if calc_cond:                                         # Kernel.py:239-245 / 279-285 (needs the factor here)
            if chofac is None:
                condK = np.nan
            elif self.cond_norm == 2:
                condK = self.calc_cond_device()
            else:
                condK = self.calc_cond_fro_device(hp)[0]
        return Kern, None, Kcov, chofac, condK, self._etaK_last, self._idx_etaK_argmax_last

    # ---- likelihood ------------------------------------------------------------------------------------
    def calc_lkd_all(self, hp_vals, calc_lkd=True, calc_cond=False, calc_grad=False, lkd_use_adj_mtd=None):
        """One marginal-log-likelihood evaluation -- reference CalcLkd.py:270-346 (value path)."""
        if calc_grad and self.bvec_use_grad is not None and not np.all(self.bvec_use_grad):
            # the reference itself fails here (shape bug KernelSqExp.py:552-554): nothing to pin against
            raise NotImplementedError('likelihood gradient with a bvec_use_grad mask is not supported')
        if lkd_use_adj_mtd is None:
            lkd_use_adj_mtd = self.lkd_use_adj_mtd                              # CalcLkd.py:292-293
        if calc_cond and self.cond_norm not in (2, 'fro'):
            raise Exception(f'cond_norm must be either 2 or "fro" but it is {self.cond_norm}')          # GpHparaCon.py:159
        noisy = self.b_has_noisy_data
        if noisy:
            assert hp_vals.varK is not None, f'varK is not provided and hp_vals.varK is None, hp_vals = {hp_vals}'
            varK_mat = hp_vals.varK
            assert varK_mat > 0, f'varK must be positive but varK = {varK_mat}'
        else:
            varK_mat = 1.0                                                           # Kernel.py:137-138
        hp, keep = self._make_hp(hp_vals, varK_mat, closed_form=not noisy)
        out = _lib.GpgLkdOut()
        t0 = time.time()
        if calc_grad:
            g_aa, g_inv = np.zeros(self.dim + 4), np.zeros(self.dim + 4)     # slots: theta(d), varK, var_fval, var_fgrad, hp_kernel
            rc = self._lib.gpg_lkd_grad(self._ctx, C.byref(hp), C.byref(out), _lib.as_dp(g_aa), _lib.as_dp(g_inv))
        else:
            rc = self._lib.gpg_lkd(self._ctx, C.byref(hp), C.byref(out))
        self._time_chofac += time.time() - t0
        if rc < 0:
            raise _lib.GpgError(f'gpg_lkd failed ({rc}): {self._err()}')
        if rc > 0:
            # CalcLkd.py:308-311 / 330-333: the reference reports np.linalg.cond(Kcov, 2) of the matrix that could not be
            # factorised.  No factor -> no Lanczos through it: the matrix is downloaded and the SVD done on the host, for
            # moderate sizes only (this branch is the failure path, not the hot path); NaN above that.
            cond_fail, cond_grad_fail = np.nan, None
            if self.n_data <= 4096:
                Kc = np.empty((self.n_data, self.n_data))
                if self._lib.gpg_get_matrix(self._ctx, C.byref(hp), 1, _lib.as_dp(Kc)) == 0:
                    cond_fail = float(np.linalg.cond(Kc, p=2))
                    if calc_grad and self.wellcond_mtd != 'precon':
                        # calc_cond_L2_w_grad of the failed matrix (GpHparaCon.py:163-197): extreme eigenpairs on the host
                        # (the matrix is already here), the quadratic forms v' (d Kcov / d hp_k) v on the device
                        ev, evec = np.linalg.eigh(Kc)
                        cond_grad_fail = self._cond_grad_from_vectors(hp, cond_fail, ev[0], evec[:, -1], evec[:, 0])
            return LkdInfo(cond=cond_fail, cond_grad=cond_grad_fail), False
        hp_beta_grad = None
        if calc_grad and (noisy or not lkd_use_adj_mtd) and not self._skip_beta_grad:
            # GpMeanFun.py:109-120 (the noisy path always passes K_grad_hp: CalcLkd.py:216-217); needs the state the gradient call
            # left on the device, so before anything else runs there
            hp_beta_grad = self._beta_grad_device(hp, hp_vals, varK_mat)
        cond = cond_grad = None
        if calc_cond and self.cond_norm == 'fro':                    # calc_cond_fronorm_w_grad (GpHparaCon.py:209-236)
            assert not (calc_grad and self.wellcond_mtd == 'precon'), \
                'Not setup to calculate the gradient of the condition number if wellcond_mtd = "precon" '
            cond, cond_grad = self.calc_cond_fro_device(hp, want_grad=calc_grad)
            if self.wellcond_mtd == 'precon':
                if cond > 1.1 * self.cond_max:                                   # Kernel.py:242-243
                    print(f'*** WARNING: condK = {cond:.2e} which is greater than cond_max = {self.cond_max:.2e} ***')
            elif cond > self.cond_max_abs:                                       # Kernel.py:282-283
                return LkdInfo(cond=cond), False
        elif calc_cond:
            if calc_grad and self.wellcond_mtd != 'precon':          # GpHparaCon.py:171-173: no gradient with 'precon'
                cond, cond_grad = self.calc_cond_device(want_grad=True, hp_struct=hp)
            else:
                cond = self.calc_cond_device()
                if calc_grad:
                    print('Not setup to calculate the gradient of the condition number if wellcond_mtd = "precon" ')
            if self.wellcond_mtd == 'precon':
                if cond > 1.1 * self.cond_max:                                   # Kernel.py:242-243
                    print(f'*** WARNING: condK = {cond:.2e} which is greater than cond_max = {self.cond_max:.2e} ***')
            elif cond > self.cond_max_abs:                                       # Kernel.py:282-283, CalcLkd.py:308-311
                return LkdInfo(cond=cond), False
        ln_lkd = out.ln_lkd
        if not noisy:
            ln_lkd -= self.calc_lkd_varK_pnlt(out.varK, self._fval_scl)[0]        # CalcLkd.py:162,168
        ln_lkd_grad = hp_varK_grad = ln_det_grad = None
        if calc_grad:
            # adjoint weights: CalcLkd.py:173-177 (noise-free), :233-235 (noisy).  The direct method (lkd_use_adj_mtd = False,
            # CalcLkd.py:69-86, 237-242) combines the same two contractions, alpha' G_k alpha and tr(Kcov^-1 G_k):
            # hp_varK_grad = -alpha' G alpha / N (V' alpha = 0), ln_det_Kmat_grad = tr(Kcov^-1 G), and its ln_lkd_grad is the
            # adjoint one term by term.
            if noisy:
                s_aa = 0.5
            else:
                s_aa = self.calc_lkd_varK_pnlt(out.varK, self._fval_scl)[1] / self.n_data + 1.0 / (2.0 * out.varK)
            if calc_lkd:
                ln_lkd_grad = self._slots_to_hp(s_aa * g_aa + g_inv)
            if not lkd_use_adj_mtd:
                ln_det_grad = self._slots_to_hp(-2.0 * g_inv) if calc_lkd else None
                if not noisy:
                    hp_varK_grad = self._slots_to_hp(-g_aa / self.n_data)
        info = LkdInfo(hp_beta=np.array([out.beta]), hp_beta_grad=hp_beta_grad, hp_varK=None if noisy else out.varK,
                       hp_varK_grad=hp_varK_grad, ln_det_Kmat=out.ln_det if calc_lkd or not noisy else None,
                       ln_det_Kmat_grad=ln_det_grad, ln_lkd=ln_lkd if calc_lkd else None, ln_lkd_grad=ln_lkd_grad,
                       data_vec=self._data_vec if noisy else None, cond=cond, cond_grad=cond_grad)
        return info, True

    _noise_override = False
    _skip_beta_grad = False       # the optimiser's objective sets it: OptzLkd.py reads ln_lkd_grad / cond_grad only

    def _slots_to_hp(self, g_all):
        """The C ABI's gradient slots [theta(d), varK, var_fval, var_fgrad, hp_kernel] in hp_info_optz_lkd order."""
        hi, d = self.hp_info_optz_lkd, self.dim
        out = np.zeros(hi.n_hp)
        out[hi.idx_theta] = g_all[:d]
        if hi.has_kernel:
            out[hi.idx_kernel] = g_all[d + 3]
        if hi.has_varK:
            out[hi.idx_varK] = g_all[d]
        if hi.has_var_fval:
            out[hi.idx_var_fval] = g_all[d + 1]
        if hi.has_var_fgrad:
            out[hi.idx_var_fgrad] = g_all[d + 2]
        return out

    def _beta_grad_device(self, hp, hp_vals, varK_mat):
        """hp_beta_grad [1, n_hp] of calc_model_max_lkd_poly (GpMeanFun.py:109-120): beta_grad_k = term1 G_k term2 with
        term1 = (V' K^-1 V)^-1 (K^-1 V)' =: u' and term2 = K^-1 V beta - K^-1 y = -alpha, i.e. -u' G_k alpha -- the polarisation of
        the device's quadratic form v' G_k v (gpg_dcov_quadform) at u + alpha and u - alpha; K^-1 V through the factor on the
        device (gpg_factor_apply), alpha from gpg_lkd_alpha.  Right after the gpg_lkd_grad call of calc_lkd_all."""
        N, n = self.n_data, self.n_eval
        alpha = np.empty(N)
        rc = self._lib.gpg_lkd_alpha(self._ctx, _lib.as_dp(alpha))
        if rc != 0:
            raise _lib.GpgError(f'gpg_lkd_alpha failed ({rc}): {self._err()}')
        pvec = np.ones(N)
        if self.wellcond_mtd == 'precon':                                        # Kernel.py:224: sqrt(diag(K + noise / varK))
            kd = np.ones(N)
            kd[n:] = np.kron(self.theta2gamma(np.asarray(hp_vals.theta, dtype=float)) ** 2, np.ones(self.n_grad))
            pvec = np.sqrt(kd + self.calc_noise_vec(hp_vals) / varK_mat)
        V = np.zeros(N)
        V[:n] = 1.0
        w = np.empty(N)
        rc = self._lib.gpg_factor_apply(self._ctx, 1, _lib.as_dp(np.ascontiguousarray(V / pvec)), _lib.as_dp(w))
        if rc != 0:
            raise _lib.GpgError(f'gpg_factor_apply failed ({rc}): {self._err()}')
        KinvV = w / pvec
        u = KinvV / (V @ KinvV)
        q = np.zeros((2, self.dim + 4))
        for i, v in enumerate((u + alpha, u - alpha)):
            rc = self._lib.gpg_dcov_quadform(self._ctx, C.byref(hp), _lib.as_dp(np.ascontiguousarray(v)), _lib.as_dp(q[i]))
            if rc != 0:
                raise _lib.GpgError(f'gpg_dcov_quadform failed ({rc}): {self._err()}')
        return -self._slots_to_hp((q[0] - q[1]) / 4.0)[None, :]

    def calc_lkd_varK_pnlt(self, varK, fval_vec):
        # CalcLkd.py:118-133 (noise-free path only, as in the reference: CalcLkd.py:204 leaves the noisy path out)
        if self.lkd_varK_pnlt_use:
            var_fval = np.max((np.var(fval_vec), self.lkd_varK_pnlt_lb_var))
            max_fun = np.max((varK - self.lkd_varK_pnlt_c2 * var_fval, 0))
            return self.lkd_varK_pnlt_c1 * var_fval * max_fun ** 2, 2 * self.lkd_varK_pnlt_c1 * var_fval * max_fun
        return 0, 0

    def _rows_from_hp_x0(self, hp_x0):
        """Decode optimiser rows (GpHpara.py:56-103) into the C ABI's [theta(d), varK_mat, var_fval, var_fgrad]."""
        info = self.hp_info_optz_lkd
        hp_x0 = np.atleast_2d(np.asarray(hp_x0, dtype=np.float64))
        assert hp_x0.shape[1] == info.n_hp, f'hp rows must have n_hp = {info.n_hp} entries'
        v = hp_x0.copy()
        v[:, info.bvec_log_optz] = 10.0 ** v[:, info.bvec_log_optz]
        m, d = v.shape[0], self.dim
        rows = np.empty((m, d + 4))
        rows[:, d + 3] = (v[:, info.idx_kernel[0]] if info.has_kernel else self.hp_kernel_default) if self.kernel_has_hp else 0.0
        rows[:, :d] = v[:, info.idx_theta]
        rows[:, d] = v[:, info.idx_varK] if info.has_varK else 1.0
        rows[:, d + 1] = v[:, info.idx_var_fval] if info.has_var_fval else -1.0
        rows[:, d + 2] = v[:, info.idx_var_fgrad] if info.has_var_fgrad else -1.0
        return rows

    def calc_lkd_batch(self, hp_x0, return_all=False):
        """ln_lkd of every restart row on this device, queued back-to-back -- the loop body of
        GpHparaX0.py:39-45.  Failed factorisations give NaN (GpHparaX0.py:34,43-45)."""
        rows = self._rows_with_eta(np.ascontiguousarray(self._rows_from_hp_x0(hp_x0)))
        m = rows.shape[0]
        outs = (_lib.GpgLkdOut * m)()
        t0 = time.time()
        rc = self._lib.gpg_lkd_batch(self._ctx, m, _lib.as_dp(rows), rows.shape[1], float(self._etaK),
                                     self._wellcond_code, int(not self.b_has_noisy_data), outs)
        self._time_chofac += time.time() - t0
        if rc != 0:
            raise _lib.GpgError(f'gpg_lkd_batch failed ({rc}): {self._err()}')
        ln = np.array([o.ln_lkd if o.info == 0 else np.nan for o in outs])
        if self.lkd_varK_pnlt_use and not self.b_has_noisy_data:
            ln = ln - np.array([self.calc_lkd_varK_pnlt(o.varK, self._fval_scl)[0] if o.info == 0 else 0.0 for o in outs])
        if return_all:
            return ln, outs
        return ln

    def calc_lkd_grad_batch(self, hp_x0):
        """Value and adjoint gradient (d ln_lkd / d hp in hp_info_optz_lkd order, with respect to the hyperparameter VALUES
        like calc_lkd_all(calc_grad=True), CalcLkd.py:170-177 / 230-235) of every row of hp_x0 in ONE device call
        (gpg_lkd_grad_batch).  Returns (ln_lkd [m], ln_lkd_grad [m, n_hp], ok [m]); rows whose factorisation failed have
        ok False and NaN entries (the caller decides what a failure means: OptzLkd.py:74-77)."""
        if self.bvec_use_grad is not None and not np.all(self.bvec_use_grad):
            raise NotImplementedError('likelihood gradient with a bvec_use_grad mask is not supported')
        rows = self._rows_with_eta(np.ascontiguousarray(self._rows_from_hp_x0(hp_x0)))
        m, d = rows.shape[0], self.dim
        outs = (_lib.GpgLkdOut * m)()
        g_aa, g_inv = np.zeros((m, d + 4)), np.zeros((m, d + 4))
        noisy = self.b_has_noisy_data
        t0 = time.time()
        rc = self._lib.gpg_lkd_grad_batch(self._ctx, m, _lib.as_dp(rows), rows.shape[1], float(self._etaK),
                                          self._wellcond_code, int(not noisy), outs, _lib.as_dp(g_aa), _lib.as_dp(g_inv))
        self._time_chofac += time.time() - t0
        if rc != 0:
            raise _lib.GpgError(f'gpg_lkd_grad_batch failed ({rc}): {self._err()}')
        hi = self.hp_info_optz_lkd
        ln, grad, ok = np.full(m, np.nan), np.full((m, hi.n_hp), np.nan), np.zeros(m, dtype=bool)
        for i, o in enumerate(outs):
            if o.info != 0:
                continue
            ok[i] = True
            if noisy:
                s_aa, pn = 0.5, 0.0
            else:
                pnlt = self.calc_lkd_varK_pnlt(o.varK, self._fval_scl)
                s_aa, pn = pnlt[1] / self.n_data + 1.0 / (2.0 * o.varK), pnlt[0]
            ln[i] = o.ln_lkd - pn
            g_all = s_aa * g_aa[i] + g_inv[i]
            grad[i, hi.idx_theta] = g_all[:d]
            if hi.has_kernel:
                grad[i, hi.idx_kernel] = g_all[d + 3]
            if hi.has_varK:
                grad[i, hi.idx_varK] = g_all[d]
            if hi.has_var_fval:
                grad[i, hi.idx_var_fval] = g_all[d + 1]
            if hi.has_var_fgrad:
                grad[i, hi.idx_var_fgrad] = g_all[d + 2]
        return ln, grad, ok

    def select_hp_best(self, hp_x0):
        """hp_best selection of GpHparaX0.py:33-59 for explicit start rows: the row of highest ln_lkd."""
        ln = self.calc_lkd_batch(hp_x0)
        idx = int(np.nanargmax(ln))
        return np.atleast_2d(hp_x0)[idx][None, :], ln, idx

    def optz_closed_form_hp(self, hp_vals):
        # GpHparaOptz.py:220-230
        lkd_info, b_chofac_good = self.calc_lkd_all(hp_vals, calc_lkd=False, calc_cond=False, calc_grad=False)
        hp_vals.beta = lkd_info.hp_beta
        if self.b_has_noisy_data is False:
            hp_vals.varK = lkd_info.hp_varK
        return hp_vals

    # ---- posterior -------------------------------------------------------------------------------------
    def set_hpara(self, method2set_hp, i_optz, hp_vals=None, calc_cond=False):
        # GaussianProcess.py:365-395
        assert type(method2set_hp) is str, 'method2set_hp must be a string'
        if method2set_hp == 'stored':
            assert i_optz >= 0
            self.set_hp_from_idx(i_optz)
        elif method2set_hp == 'optz':
            self.optz_hp(i_optz)                                                  # gpgradpy_amd/hpara_optz.py
        elif method2set_hp == 'current':
            assert i_optz > 0
            assert self.hp_vals is not None, 'Cannot use current hp_vals if they have not been set yet'
        elif method2set_hp == 'set':
            assert hp_vals is not None, 'If method2set_hp == "set", then the class hp_vals must be provided'
            self.hp_vals = hp_vals
        else:
            raise Exception(f'Unknown method to set GP hp: method2set_hp = {method2set_hp}')
        self.setup_eval_model(calc_cond=calc_cond)

    def setup_eval_model(self, calc_cond=False):
        """GpEvalModel.py:17-57: factor kept on the device, alpha = K^-1 (y - V beta) returned to the host."""
        self._hp_vals_model_setup = copy.copy(self.hp_vals)
        hp, keep = self._make_hp(self.hp_vals, 1.0, closed_form=not self.b_has_noisy_data)   # b_normlz_w_varK=True
        beta = float(np.ravel(self.hp_vals.beta)[0])
        alpha = np.empty(self.n_data)
        t0 = time.time()
        rc = self._lib.gpg_setup_eval(self._ctx, C.byref(hp), beta, _lib.as_dp(alpha))
        self._time_chofac += time.time() - t0
        if rc < 0:
            raise _lib.GpgError(f'gpg_setup_eval failed ({rc}): {self._err()}')
        self.data_vec = self._data_vec
        self.Kern = self.KernEta = None
        self.condK = None
        self.etaK_eval = self._etaK_last
        if rc > 0:
            self.KernEta_chofac = None
            self.invKernEta_fdiff = None
            self._eval_ready = False
        else:
            self.KernEta_chofac = DeviceChoFactor(self)   # the factor stays in HBM; downloaded if somebody unpacks it
            self.invKernEta_fdiff = alpha
            self._eval_ready = True
            if calc_cond:                                 # GpEvalModel.py:39-41 -> Kernel.py:239-245 / 279-285
                self.condK = self.calc_cond_device() if self.cond_norm == 2 else self.calc_cond_fro_device(hp)[0]

    def eval_model(self, x2model_in, calc_grad=False, calc_hess=False, squeeze_nx=False):
        """GpEvalModel.py:59-198: returns (mu, sig, dmudx, dsigdx, d2mudx2, d2sigdx2); the Hessians are
        evaluated one point per call, as in the reference (GpEvalModel.py:358,369)."""
        assert self.KernEta_chofac is not None, 'To evaluate the surr the Cholesky decomposition is required'
        if calc_hess:
            assert calc_grad, 'To return the hessian calc_grad must also be set to True'      # GpEvalModel.py:126-127
            if self.bvec_use_grad is not None and not np.all(self.bvec_use_grad):
                # reference shape bug with masks in calc_KernGrad_grad_x (KernelSqExp.py:451-452): nothing to pin against
                raise NotImplementedError('posterior Hessians with a bvec_use_grad mask are not supported')
        if x2model_in.ndim == 1:
            x2model = x2model_in[None, :]
        elif x2model_in.ndim == 2:
            x2model = x2model_in
        else:
            raise Exception(f'x2model_in should be a 2d array but it has shape {x2model_in.shape}')
        nx = x2model.shape[0]
        if squeeze_nx:
            assert nx == 1, 'If squeeze_nx is True, then x_acq must only have one point'
        if (self.hp_vals == self._hp_vals_model_setup) is False:
            raise Exception('Cannot change hp_vals between calling setup_eval_model() and eval_model()')
        if not self._eval_ready:
            raise Exception('setup_eval_model() must be called (again): set_data() replaced the data of the device model')
        if self.b_use_data_scl:
            x2model = self.DataScl.x_init_2_scl(x2model)                                   # GpEvalModel.py:121-122
        xq = np.ascontiguousarray(x2model, dtype=np.float64)
        mu, sig, s2 = np.empty(nx), np.empty(nx), np.empty(nx)
        dmudx = dsigdx = d2mudx2 = d2sigdx2 = None
        if calc_hess:
            assert nx == 1, 'calc_d2mudx2 can only be used on one point per call'          # GpEvalModel.py:358
            dmudx, dsigdx = np.empty((1, self.dim)), np.empty((1, self.dim))
            d2mudx2, d2sigdx2 = np.empty((1, self.dim, self.dim)), np.empty((1, self.dim, self.dim))
            rc = self._lib.gpg_predict_hess(self._ctx, _lib.as_dp(xq), float(self.hp_vals.varK), _lib.as_dp(mu),
                                            _lib.as_dp(sig), _lib.as_dp(dmudx), _lib.as_dp(dsigdx), _lib.as_dp(d2mudx2),
                                            _lib.as_dp(d2sigdx2))
            s2[:] = (sig / np.sqrt(self.hp_vals.varK)) ** 2
        elif calc_grad:
            dmudx, dsigdx = np.empty((nx, self.dim)), np.empty((nx, self.dim))
            rc = self._lib.gpg_predict_grad(self._ctx, nx, _lib.as_dp(xq), float(self.hp_vals.varK), _lib.as_dp(mu),
                                            _lib.as_dp(sig), _lib.as_dp(s2), _lib.as_dp(dmudx), _lib.as_dp(dsigdx))
        else:
            rc = self._lib.gpg_predict(self._ctx, nx, _lib.as_dp(xq), float(self.hp_vals.varK), _lib.as_dp(mu),
                                       _lib.as_dp(sig), _lib.as_dp(s2))
        if rc != 0:
            raise _lib.GpgError(f'gpg_predict failed ({rc}): {self._err()}')
        assert np.min(s2) >= 0, \
            f'The variance of the surr should be non-negative but min(sig2_wo_sigK) = {np.min(s2)}'   # GpEvalModel.py:163
        if self.b_use_data_scl:                                                            # GpEvalModel.py:181-183
            mu, sig, dmudx, dsigdx, d2mudx2, d2sigdx2 = self.data_scl_2_init(mu, sig, dmudx, dsigdx, d2mudx2, d2sigdx2)
        if squeeze_nx:
            if calc_hess:
                return mu[0], sig[0], dmudx[0, :], dsigdx[0, :], d2mudx2[0], d2sigdx2[0]  # GpEvalModel.py:186-196
            if calc_grad:
                return mu[0], sig[0], dmudx[0, :], dsigdx[0, :], None, None
            return mu[0], sig[0], None, None, None, None
        return mu, sig, dmudx, dsigdx, d2mudx2, d2sigdx2

    def eval_model_var(self, x2model_in, calc_grad=False, calc_hess=False, squeeze_nx=False):
        """GpEvalModel.py:200-317: variance form of the posterior, returns (sig2, dsig2dx, d2sig2dx2) with
        sig2 = varK (1 - diag(Kxy K^-1 Kyx)) and dsig2dx of calc_dsig2dx (:327-337)."""
        assert self.KernEta_chofac is not None, 'To evaluate the surr the Cholesky decomposition is required'
        if x2model_in.ndim == 1:
            x2model = x2model_in[None, :]
        elif x2model_in.ndim == 2:
            x2model = x2model_in
        else:
            raise Exception(f'x2model_in should be a 2d array but it has shape {x2model_in.shape}')
        nx = x2model.shape[0]
        if squeeze_nx:
            assert nx == 1, 'If squeeze_nx is True, then x_acq must only have one point'
        if (self.hp_vals == self._hp_vals_model_setup) is False:
            raise Exception('Cannot change hp_vals between calling setup_eval_model() and eval_model()')
        if self.b_use_data_scl:                                                            # GpEvalModel.py:253-256
            raise Exception('The method eval_model_var() is not setup for cases where data must be rescaled')
        if calc_hess:
            assert calc_grad, 'To return the hessian calc_grad must also be set to True'
        if not self._eval_ready:
            raise Exception('setup_eval_model() must be called (again): set_data() replaced the data of the device model')
        xq = np.ascontiguousarray(x2model, dtype=np.float64)
        sig2 = np.empty(nx)
        dsig2dx = np.empty((nx, self.dim)) if calc_grad else None
        rc = self._lib.gpg_predict_var(self._ctx, nx, _lib.as_dp(xq), float(self.hp_vals.varK), _lib.as_dp(sig2),
                                       None if dsig2dx is None else _lib.as_dp(dsig2dx))
        if rc != 0:
            raise _lib.GpgError(f'gpg_predict_var failed ({rc}): {self._err()}')
        assert np.min(sig2) >= 0, f'The variance of the surr should be non-negative but min(sig2) = {np.min(sig2)}'   # :298
        if calc_hess:
            raise Exception('Must add method to calculate d2sig2dx2')                                            # :306
        if squeeze_nx:
            return sig2[0], (dsig2dx[0, :] if calc_grad else None), None
        return sig2, dsig2dx, None

    def shard_restarts(self, group):
        """Opt in to sharding the restart rows of `set_hpara('optz')` (the hp_best pre-selection and the SLSQP
        multi-start) over the ranks of a torch.distributed process group, one process per GPU;
        None switches it off.  Every rank of the group must then make the same calls with the same data."""
        self.restart_group = group

    def _cond_grad_from_vectors(self, hp_struct, cond, lam_min, v_max, v_min):
        """d cond / d hp_k = (v_max' G_k v_max - cond v_min' G_k v_min) / max(lam_min, 1e-16), G_k = d Kmat / d hp_k
        (GpHparaCon.py:175-197), the quadratic forms on the device, laid out in hp_info_optz_lkd order."""
        q = []
        for v in (v_max, v_min):
            out = np.zeros(self.dim + 4)
            v = np.ascontiguousarray(v, dtype=np.float64)
            rc = self._lib.gpg_dcov_quadform(self._ctx, C.byref(hp_struct), _lib.as_dp(v), _lib.as_dp(out))
            if rc != 0:
                raise _lib.GpgError(f'gpg_dcov_quadform failed ({rc}): {self._err()}')
            q.append(out)
        g_all = (q[0] - cond * q[1]) / max(lam_min, 1e-16)                      # eig_min_mod, GpHparaCon.py:192
        hi, d = self.hp_info_optz_lkd, self.dim
        cond_grad = np.zeros(hi.n_hp)
        cond_grad[hi.idx_theta] = g_all[:d]
        if hi.has_kernel:
            cond_grad[hi.idx_kernel] = g_all[d + 3]
        if hi.has_varK:
            cond_grad[hi.idx_varK] = g_all[d]
        if hi.has_var_fval:
            cond_grad[hi.idx_var_fval] = g_all[d + 1]
        if hi.has_var_fgrad:
            cond_grad[hi.idx_var_fgrad] = g_all[d + 2]
        return cond_grad

    # ---- instrumentation -------------------------------------------------------------------------------
    def prof_enable(self, cats):
        mask = 0
        for c in cats:
            mask |= 1 << _lib.PROF_CATS.index(c)
        self._lib.gpg_prof_enable(self._ctx, mask)

    def prof_read(self):
        ms = (C.c_double * _lib.PROF_NCAT)()
        cnt = (C.c_longlong * _lib.PROF_NCAT)()
        work = (C.c_double * _lib.PROF_NCAT)()
        rc = self._lib.gpg_prof_read(self._ctx, ms, cnt, work)
        if rc != 0:
            raise _lib.GpgError(f'gpg_prof_read failed ({rc}): {self._err()}')
        return {c: dict(ms=ms[i], count=cnt[i], work=work[i]) for i, c in enumerate(_lib.PROF_CATS)}

    def set_panel(self, nb_outer):
        rc = self._lib.gpg_set_panel(self._ctx, int(nb_outer))
        if rc != 0:
            raise _lib.GpgError(f'gpg_set_panel failed ({rc}): {self._err()}')

    def set_lookahead(self, on):
        self._lib.gpg_set_lookahead(self._ctx, int(on))

    FACTOR_MODES = {'auto': 0, 'blocked': 1, 'tile64': 2, 'tile128': 3}

    def set_factor_mode(self, mode):
        """Factorisation schedule of the device Cholesky (include/gpgrad.h: gpg_factor_mode): 'auto' (default,
        one dataflow launch), 'blocked' (right-looking panels), 'tile64' / 'tile128' (force one dataflow kernel)."""
        rc = self._lib.gpg_set_factor_mode(self._ctx, self.FACTOR_MODES[mode] if isinstance(mode, str) else int(mode))
        if rc != 0:
            raise _lib.GpgError(f'gpg_set_factor_mode failed ({rc}): {self._err()}')

    def set_batch(self, max_matrices):
        """Restart rows factorised per launch by calc_lkd_batch on small matrices (include/gpgrad.h: gpg_set_batch)."""
        rc = self._lib.gpg_set_batch(self._ctx, int(max_matrices))
        if rc != 0:
            raise _lib.GpgError(f'gpg_set_batch failed ({rc}): {self._err()}')

    def set_max_workgroups(self, n):
        """Cap on the persistent workgroups of every dataflow launch (include/gpgrad.h: gpg_set_max_workgroups; 0 = default)."""
        rc = self._lib.gpg_set_max_workgroups(self._ctx, int(n))
        if rc != 0:
            raise _lib.GpgError(f'gpg_set_max_workgroups failed ({rc}): {self._err()}')

    def reserve_batch(self, rows):
        """Allocate the workspaces calc_lkd_batch(rows) will use (setup, like set_data)."""
        rc = self._lib.gpg_reserve_batch(self._ctx, int(rows))
        if rc != 0:
            raise _lib.GpgError(f'gpg_reserve_batch failed ({rc}): {self._err()}')

    def factor_fallbacks(self):
        """How often a dataflow factorisation timed out (device shared with another such launch) and the call was
        repeated with the blocked schedule; the context stays on 'blocked' until set_factor_mode is called."""
        return int(self._lib.gpg_factor_fallbacks(self._ctx)) if self._ctx else 0

    def overlap_fallbacks(self):
        """How often a value + gradient call with the overlapped inverse timed out and was repeated without the overlap."""
        return int(self._lib.gpg_overlap_fallbacks(self._ctx)) if self._ctx else 0

    def solve_fallbacks(self):
        """How often a dataflow triangular solve timed out and the call was repeated with the blocked sweeps."""
        return int(self._lib.gpg_solve_fallbacks(self._ctx)) if self._ctx else 0

    def _lkd_grad_central_differences(self, hp_vals, rel_step=1e-4):
        """d ln_lkd / d hp_k for the optimised hyperparameters (hp_info_optz_lkd order, derivatives with respect to the
        hyperparameter VALUES like CalcLkd.py:170-177, not their log10) by central differences of the device likelihood;
        the closed-form varK of the noise-free path is re-evaluated in every term, i.e. this is the total derivative the
        reference's adjoint formula gives.  Diagnostic cross-check of the adjoint gradient (tests); not on any product path."""
        hi = self.hp_info_optz_lkd
        x = np.zeros(hi.n_hp)
        x[hi.idx_theta] = hp_vals.theta
        if hi.has_kernel:
            x[hi.idx_kernel] = np.asarray(hp_vals.kernel, dtype=float).reshape(-1)[0]
        if hi.has_varK:
            x[hi.idx_varK] = hp_vals.varK
        if hi.has_var_fval:
            x[hi.idx_var_fval] = hp_vals.var_fval
        if hi.has_var_fgrad:
            x[hi.idx_var_fgrad] = hp_vals.var_fgrad
        b = hi.bvec_log_optz

        def to_optz(v):
            v = v.copy()
            v[b] = np.log10(v[b])
            return v
        rows, steps = [], []
        for k in range(hi.n_hp):
            h = rel_step * abs(x[k]) if x[k] != 0.0 else 1e-10
            xp, xm = x.copy(), x.copy()
            xp[k] += h
            xm[k] -= h
            rows += [to_optz(xp), to_optz(xm)]
            steps.append(h)
        ln = self.calc_lkd_batch(np.array(rows))
        return (ln[0::2] - ln[1::2]) / (2.0 * np.array(steps))

    def calc_cond_fro_device(self, hp_struct, want_grad=False):
        """Frobenius-norm condition number ||K||_F ||K^-1||_F of the matrix that is factorised for hp_struct -- what
        np.linalg.cond(., 'fro') gives in Kernel.py:239-245 / 279-285 -- and, on request, its gradient in hp_info_optz_lkd
        order (calc_cond_fronorm_w_grad, GpHparaCon.py:209-236), all on the device (gpg_cond_fro)."""
        cond = C.c_double(0.0)
        g_all = np.zeros(self.dim + 4) if want_grad else None
        rc = self._lib.gpg_cond_fro(self._ctx, C.byref(hp_struct), C.byref(cond), None if g_all is None else _lib.as_dp(g_all))
        if rc < 0:
            raise _lib.GpgError(f'gpg_cond_fro failed ({rc}): {self._err()}')
        if rc > 0:
            return np.nan, None
        if not want_grad:
            return cond.value, None
        hi, d = self.hp_info_optz_lkd, self.dim
        cond_grad = np.zeros(hi.n_hp)
        cond_grad[hi.idx_theta] = g_all[:d]
        if hi.has_kernel:
            cond_grad[hi.idx_kernel] = g_all[d + 3]
        if hi.has_varK:
            cond_grad[hi.idx_varK] = g_all[d]
        if hi.has_var_fval:
            cond_grad[hi.idx_var_fval] = g_all[d + 1]
        if hi.has_var_fgrad:
            cond_grad[hi.idx_var_fgrad] = g_all[d + 2]
        return cond.value, cond_grad

    def calc_cond_device(self, want_grad=False, hp_struct=None):
        """2-norm condition number of the matrix factorised last -- Kcov_precon = varK (Kcor + eta I) for 'precon',
        Kcov for 'base' (what Kernel.py:239-245, 279-285 pass to np.linalg.cond) -- from Lanczos runs on
        v -> (L L^T) v and v -> (L L^T)^-1 v through the factor in HBM (gpgradpy_amd/cond_number.py)."""
        from .cond_number import cond_from_factor
        N = self.n_data

        def op(which):
            def f(v):
                v = np.ascontiguousarray(v, dtype=np.float64)
                out = np.empty(N)
                rc = self._lib.gpg_factor_apply(self._ctx, which, _lib.as_dp(v), _lib.as_dp(out))
                if rc != 0:
                    raise _lib.GpgError(f'gpg_factor_apply failed ({rc}): {self._err()}')
                return out
            return f
        if not want_grad:
            return cond_from_factor(op(0), op(1), N)
        # gradient of the condition number (GpHparaCon.py:163-207): with the unit eigenvectors of lambda_max / lambda_min
        #   d cond / d hp_k = (v_max^T G_k v_max - cond v_min^T G_k v_min) / lambda_min,   G_k = d Kmat / d hp_k,
        # the quadratic forms on the device (gpg_dcov_quadform), the hyperparameter values in hp_info_optz_lkd order
        cond, lam_min, v_max, v_min = cond_from_factor(op(0), op(1), N, want_vectors=True)
        return cond, self._cond_grad_from_vectors(hp_struct, cond, lam_min, v_max, v_min)

    def last_factor(self):
        """(schedule, matrices) of the most recent factorisation launch: 'blocked' | 'tile64' | 'tile128' | 'pair128'."""
        import ctypes as C
        k, b = C.c_int(0), C.c_int(0)
        self._lib.gpg_last_factor(self._ctx, C.byref(k), C.byref(b))
        return ('blocked', 'tile64', 'tile128', 'pair128')[k.value], b.value

    def download_chofac(self):
        """(P L, True) of the factor currently on the device (Kernel.py:252) as a SciPy cho_factor pair."""
        fac = np.empty((self.n_data, self.n_data))
        rc = self._lib.gpg_get_matrix(self._ctx, None, 3, _lib.as_dp(fac))
        if rc != 0:
            raise _lib.GpgError(f'gpg_get_matrix failed ({rc}): {self._err()}')
        return fac, True
