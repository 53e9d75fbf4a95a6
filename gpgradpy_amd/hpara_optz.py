"""Hyperparameter optimisation on top of the device likelihood.

Host-side mirror of the reference's optimisation driver for the accelerated path:

* history of the optimised hyperparameters        -- base/GpParaDef.py:20-65 (init), :219-284 (store)
* start points + box constraints around the median of the recent history
                                                  -- optz/GpHparaX0.py:16-65, :67-193
* objective / gradient with the log10 chain rule   -- optz/OptzLkd.py:15-100
* SLSQP multi-start and selection of the best run  -- optz/OptzLkd.py:185-333
* `optz_hp` / `get_init_hp_vals`                    -- optz/GpHparaOptz.py:140-218

Every likelihood value and gradient comes from the device (`calc_lkd_all` -> `gpg_lkd` / `gpg_lkd_grad`); the
`hp_best` pre-selection (40 value-only evaluations in the reference's loop, GpHparaX0.py:39-45) is ONE
`gpg_lkd_batch` call, sharded over ranks when `torch.distributed` is initialised.

Differences from the reference: the Latin-hypercube source is SciPy's (`scipy.stats.qmc.LatinHypercube`, seed 1)
because `smt` is not installed here -- its sample sequence differs from smt's, the bounds and everything downstream are
the reference's; condition numbers come from the factor on the device (Lanczos / gpg_cond_fro) instead of a dense SVD.
The condition-number constraint of the non-'precon' methods and the rescale methods' outer loop
(`optz_hp_max_lkd_mtd_rescale`, OptzLkd.py:114-185) are here.
"""
import threading
import time

import numpy as np
from scipy.optimize import Bounds, NonlinearConstraint, minimize


def lhs_sample(xlimits, n, seed=1):
    """n Latin-hypercube points inside xlimits[:, 0] .. xlimits[:, 1] (stand-in for smt.sampling_methods.LHS
    with random_state=1, GpHparaX0.py:188-190)."""
    from scipy.stats import qmc
    xlimits = np.asarray(xlimits, dtype=float)
    u = qmc.LatinHypercube(d=xlimits.shape[0], seed=seed).random(n)
    return xlimits[:, 0] + u * (xlimits[:, 1] - xlimits[:, 0])


def _nanmedian(a, width=None):
    """np.median of a history slice; an empty slice gives NaN (per column) without NumPy's warnings."""
    a = np.asarray(a, dtype=float)
    if a.shape[0] == 0:
        return np.full(width, np.nan) if width is not None else np.nan
    return np.median(a, axis=0)


class _LockstepObjective:
    """The SLSQP runs of one multi-start advancing in lock step (the reference runs them one after the other,
    OptzLkd.py:249-290, one value + gradient evaluation per iterate).  Every run lives in a thread of its own and is
    SciPy's unmodified SLSQP; a run's objective request blocks until every run still alive has one pending, then ONE
    batched device call (`calc_lkd_grad_batch`) serves them all.  SciPy's SLSQP keeps its state in the arrays of each
    call, so the iterates of a run are exactly those of the sequential loop."""

    def __init__(self, evaluate_rows, n_runs):
        self._evaluate_rows = evaluate_rows                  # X [k, n_hp] -> list of k (value, gradient) pairs
        self._cv = threading.Condition()
        self._pending, self._results = {}, {}
        self._live = n_runs
        self._error = None
        self.n_batches = 0
        self.n_rows = 0

    def _fire_if_complete(self):                             # caller holds the lock
        if self._error is None and self._pending and len(self._pending) == self._live:
            runs = sorted(self._pending)
            X = np.array([self._pending[r] for r in runs])
            self._pending.clear()
            try:
                vals = self._evaluate_rows(X)
                self.n_batches += 1
                self.n_rows += len(runs)
                for r, v in zip(runs, vals):
                    self._results[r] = v
            except Exception as e:     # noqa: BLE001 -- every waiting run must be released with the error
                self._error = e
            self._cv.notify_all()

    def request(self, run, x):
        with self._cv:
            self._pending[run] = np.array(x, dtype=float, copy=True)
            self._fire_if_complete()
            while run not in self._results and self._error is None:
                self._cv.wait()
            if self._error is not None:
                raise self._error
            return self._results.pop(run)

    def finished(self, run):
        with self._cv:
            self._live -= 1
            self._pending.pop(run, None)
            self._fire_if_complete()


class HparaOptz:
    """Mixin for `gpgradpy_amd.GaussianProcess`."""

    # ---- options (reference GaussianProcess.py:33-83) ---------------------------------------------------------
    optz_mtd = 'SLSQP'
    optz_n_x0 = 5
    optz_iter_max = 250
    optz_tol_obj = 1e-12
    optz_tol_x = 1e-12
    hp_const_n_eval = 1
    hp_lhs_bound_factor = 1e3
    hp_box_bound_factor = 1e5
    hp_median_n_idx = 5
    hp_theta_range = [1e-18, 1e24]
    hp_varK_range = [1e-24, 1e14]
    hp_var_fval_range = [1e-8, 1e8]
    hp_var_fgrad_range = [1e-8, 1e8]
    b_use_cond_cstr = False
    optz_lockstep = True          # run the SLSQP starts of a rank in lock step: one batched value + gradient call per round
    restart_group = None          # torch.distributed process group the restarts are sharded over (None: no sharding)
    _last_chofac_good = True
    _save_data = False
    _lkd_val = _lkd_grad = None

    # ---- history (GpParaDef.py:20-65, 67-113, 219-284) -----------------------------------------------------------
    _HIST_1D = ('hp_varK_all', 'hp_var_fval_all', 'hp_var_fgrad_all', 'hp_kernel_all', 'min_nugget_all', 'Kcov_cond_all',
                'eta_Kbase_all', 'eta_Kgrad_all', 'vmin_init_all', 'vmin_req_grad_all', 'hp_optz_success', 'hp_optz_iter_mean', 'hp_optz_iter_max',
                'hp_optz_con_good', 'optz_n_cho_fail_all', 'optz_n_cond2big_all', 'optz_max_init_cond_all',
                'time_pick_hp0_all', 'time_hp_optz_all', 'time_chofac_all', 'var_fval', 'varK_var_fval')

    def init_optz_surr(self, n_optz_max):
        self._save_data = True
        self.n_optz_max = n_optz_max
        self.hp_beta_all = np.full((n_optz_max, self.n_beta_coeff), np.nan)
        self.hp_theta_all = np.full((n_optz_max, self.dim), np.nan)
        for name in self._HIST_1D:
            setattr(self, name, np.full(n_optz_max, np.nan))
        self.Kcov_cond_at_max_all = np.full(n_optz_max, False, dtype=bool)
        self.xvec_rescaling_all = np.full((n_optz_max, self.dim), np.nan)

    def finish_optz_surr(self, n_optz_final):
        assert self._save_data, 'If the method init_optz_surr has not been called, then finish_optz_surr cannot be used'
        idx = n_optz_final
        self.hp_beta_all = self.hp_beta_all[:idx, :]
        self.hp_theta_all = self.hp_theta_all[:idx, :]
        self.xvec_rescaling_all = self.xvec_rescaling_all[:idx]
        for name in self._HIST_1D + ('Kcov_cond_at_max_all',):
            setattr(self, name, getattr(self, name)[:idx])

    def _history_names(self):
        return ('hp_beta_all', 'hp_theta_all', 'xvec_rescaling_all', 'Kcov_cond_at_max_all') + self._HIST_1D

    def export_data_surr(self, save2file=True, file2save=None, file2save_old=None):
        """GpParaDef.py:171-217: the stored history as a dict keyed `surr_name + array name`; written as an .npz file when
        save2file (default file: path_data_surr + '.npz'; an existing file is kept as file2save_old, default '..._old.npz')."""
        import os
        assert self._save_data, 'If the method init_optz_surr has not been called, then export_data_surr cannot be used'
        data2save = {self.surr_name + k: getattr(self, k) for k in self._history_names()}
        if save2file:
            file2save = file2save or self.path_data_surr + '.npz'
            file2save_old = file2save_old or self.path_data_surr + '_old.npz'
            if os.path.isfile(file2save):
                os.replace(file2save, file2save_old)
            np.savez(file2save, **data2save)
        return data2save

    def load_data_surr(self, all_data=None):
        """GpParaDef.py:115-169: fill the first rows of the history arrays from a dict / NpzFile made by export_data_surr
        (default: path_data_surr + '.npz', nothing happens if it does not exist; numpy.load without pickle)."""
        import os
        assert self._save_data, 'If the method init_optz_surr has not been called, then load_data_surr cannot be used'
        if all_data is None:
            file2load = self.path_data_surr + '.npz'
            if not os.path.isfile(file2load):
                return
            all_data = np.load(file2load, allow_pickle=False)
        name = self.surr_name
        idx = all_data[name + 'hp_beta_all'].size                                # n_beta_coeff = 1: one entry per iteration
        for k in self._history_names():
            getattr(self, k)[:idx] = all_data[name + k]

    def store_new_para_surr(self, i_optz, hp_vals, surr_optz_info=None, cond_val=np.nan, time_hp_optz=np.nan,
                            time_chofac=np.nan, time_pick_hp0=np.nan):
        self.hp_vals = hp_vals
        if self._save_data is False:
            return
        idx = i_optz
        self.time_hp_optz_all[idx] = time_hp_optz
        self.time_chofac_all[idx] = time_chofac
        self.time_pick_hp0_all[idx] = time_pick_hp0
        self.hp_beta_all[idx, :] = hp_vals.beta
        self.hp_theta_all[idx, :] = hp_vals.theta
        self.hp_kernel_all[idx] = np.nan if hp_vals.kernel is None else float(np.asarray(hp_vals.kernel, dtype=float).reshape(-1)[0])
        self.hp_varK_all[idx] = np.nan if hp_vals.varK is None else hp_vals.varK
        self.hp_var_fval_all[idx] = np.nan if hp_vals.var_fval is None else hp_vals.var_fval
        self.hp_var_fgrad_all[idx] = np.nan if hp_vals.var_fgrad is None else hp_vals.var_fgrad
        self.min_nugget_all[idx] = self._eta_Kgrad if self.use_grad else self._eta_Kbase
        self.Kcov_cond_all[idx] = cond_val
        self.Kcov_cond_at_max_all[idx] = cond_val >= (0.99 * self.cond_max)
        self.eta_Kbase_all[idx] = self._eta_Kbase
        self.eta_Kgrad_all[idx] = self._eta_Kgrad
        self.vmin_init_all[idx] = self._vmin_init                                # GpParaDef.py:254-258
        self.vmin_req_grad_all[idx] = self._vmin_req_grad
        if self.b_use_data_scl:
            self.xvec_rescaling_all[idx] = self.DataScl.xvec_scale
        if surr_optz_info is not None:
            for key in ('hp_optz_success', 'hp_optz_iter_mean', 'hp_optz_iter_max', 'hp_optz_con_good'):
                if key in surr_optz_info:
                    getattr(self, key)[idx] = surr_optz_info[key]
            self.optz_n_cho_fail_all[idx] = surr_optz_info['optz_n_cho_fail']
            self.optz_n_cond2big_all[idx] = surr_optz_info['optz_n_cond2big']
            self.optz_max_init_cond_all[idx] = surr_optz_info['optz_max_init_cond']
        self.var_fval[idx] = np.var(self._fval_in)
        if self.var_fval[idx] > 0:
            self.varK_var_fval[idx] = self.hp_varK_all[idx] / self.var_fval[idx]

    def set_hp_from_idx(self, i_optz):
        # GpHpara.py:33-54
        assert self._save_data, 'no stored hyperparameters: call init_optz_surr first'
        if np.isnan(self.hp_var_fval_all[i_optz]):
            self.known_eps_fval, hp_var_fval = True, None
        else:
            self.known_eps_fval, hp_var_fval = False, self.hp_var_fval_all[i_optz]
        if np.isnan(self.hp_var_fgrad_all[i_optz]):
            self.known_eps_fgrad, hp_var_fgrad = True, None
        else:
            self.known_eps_fgrad, hp_var_fgrad = False, self.hp_var_fgrad_all[i_optz]
        kern = None if np.isnan(self.hp_kernel_all[i_optz]) else self.hp_kernel_all[i_optz]
        self.hp_vals = self.make_hp_class(self.hp_beta_all[i_optz, :], self.hp_theta_all[i_optz, :], kern,
                                          self.hp_varK_all[i_optz], hp_var_fval, hp_var_fgrad)

    # ---- start points and bounds (GpHparaX0.py) ------------------------------------------------------------------
    def get_hp_x0_lhs_median(self, i_optz, hp_optz_info, n_x0):
        idx_min = int(np.max((0, i_optz - self.hp_median_n_idx)))
        idx_max = i_optz
        lhs_factor, box_factor = self.hp_lhs_bound_factor, self.hp_box_bound_factor
        n_hp = hp_optz_info.n_hp
        lhs_lb, lhs_ub = np.full(n_hp, np.nan), np.full(n_hp, np.nan)
        box_lb, box_ub = np.full(n_hp, np.nan), np.full(n_hp, np.nan)

        def fill(idx, med, rng, init=None):
            # no stored history yet (i_optz = 0, or rows never filled): the reference would take the median of an empty
            # slice (NaN) and stop with 'Invalid bounds'; start from the initial hyperparameter instead
            if init is not None:
                med = np.where(np.isnan(med), init, med)
            med = np.minimum(np.maximum(med, rng[0]), rng[1])
            lhs_lb[idx] = np.maximum(med / lhs_factor, rng[0])
            lhs_ub[idx] = np.minimum(med * lhs_factor, rng[1])
            box_lb[idx] = np.maximum(med / box_factor, rng[0])
            box_ub[idx] = np.minimum(med * box_factor, rng[1])

        if hp_optz_info.has_theta:
            fill(hp_optz_info.idx_theta, _nanmedian(self.hp_theta_all[idx_min:idx_max, :], self.dim), self.hp_theta_range, self.hp_theta_init)
        if hp_optz_info.has_kernel:                                              # GpHparaX0.py:100-111
            fill(hp_optz_info.idx_kernel, _nanmedian(self.hp_kernel_all[idx_min:idx_max]), self.hp_kernel_range,
                 self.hp_kernel_default)
        if hp_optz_info.has_varK:
            fill(hp_optz_info.idx_varK, _nanmedian(self.hp_varK_all[idx_min:idx_max]), self.hp_varK_range, self.hp_varK_init)
        if hp_optz_info.has_var_fval:
            med = np.fmax(self.hp_var_fval_range[0], _nanmedian(self.hp_var_fval_all[idx_min:idx_max]))
            fill(hp_optz_info.idx_var_fval, med, self.hp_var_fval_range)
        if hp_optz_info.has_var_fgrad:
            med = np.fmax(self.hp_var_fgrad_range[0], _nanmedian(self.hp_var_fgrad_all[idx_min:idx_max]))
            fill(hp_optz_info.idx_var_fgrad, med, self.hp_var_fgrad_range)

        bvec = hp_optz_info.bvec_log_optz
        for v in (lhs_lb, lhs_ub, box_lb, box_ub):
            v[bvec] = np.log10(v[bvec])
        if np.any(lhs_lb > lhs_ub) or np.any(np.isnan(lhs_lb)):
            raise Exception(f'Invalid bounds for lhs: lhs_lb = {lhs_lb}, lhs_ub = {lhs_ub}')
        if np.any(box_lb > box_ub):
            raise Exception(f'Invalid bounds for box: box_lb = {box_lb}, box_ub = {box_ub}')
        hp_optz_bounds = Bounds(box_lb, box_ub, keep_feasible=True)
        if lhs_lb.size == 1:
            hp_x0 = np.linspace(lhs_lb[0], lhs_ub[0], n_x0 + 2)[1:-1, None]      # no nodes at the boundaries
        else:
            hp_x0 = lhs_sample(np.array([lhs_lb, lhs_ub]).T, n_x0, seed=1)
        return hp_x0, hp_optz_bounds

    def select_hp_optz_x0(self, i_optz, hp_optz_info):
        start_time = time.time()
        if self.lkd_optz_start_mtd == 'lhs':
            n_x0 = self.optz_n_x0
        elif self.lkd_optz_start_mtd == 'hp_best':
            n_x0 = self.lkd_hp_best_n_eval
        else:
            raise Exception(f'Unknown lkd_optz_start_mtd: {self.lkd_optz_start_mtd}')
        hp_x0, optz_bound = self.get_hp_x0_lhs_median(i_optz, hp_optz_info, n_x0)
        if self.lkd_optz_start_mtd == 'hp_best' and self.wellcond_mtd != 'precon':
            # GpHparaX0.py:33-59 with calc_cond: rows whose condition number exceeds 1.2 cond_max drop out (at least
            # the best-conditioned one is kept); the condition number needs each factor, so one row at a time
            ln_lkd_all, cond_all = np.full(n_x0, np.nan), np.full(n_x0, np.nan)
            for i in range(n_x0):
                lkd_info, ok = self.calc_lkd_all(self.hp_vec2dataclass(self.hp_info_optz_lkd, hp_x0[i, :]), calc_cond=True)
                if ok:
                    ln_lkd_all[i], cond_all[i] = lkd_info.ln_lkd, lkd_info.cond
            no_good = cond_all > (1.2 * self.cond_max)
            if np.sum(no_good) == no_good.size:
                no_good[np.nanargmin(cond_all)] = False
            ln_lkd_all[no_good] = np.nan
            hp_x0 = hp_x0[np.nanargmax(ln_lkd_all), :][None, :]
        elif self.lkd_optz_start_mtd == 'hp_best':
            # the reference's loop over calc_lkd_all (GpHparaX0.py:39-45) = one batched device call, sharded
            # over ranks when torch.distributed is up; failed factorisations are NaN and drop out of nanargmax
            from .multistart import select_best_restart
            hp_x0 = select_best_restart(hp_x0, self.calc_lkd_batch, group=self.restart_group)[0]
        return hp_x0, optz_bound, time.time() - start_time

    # ---- objective (OptzLkd.py:15-100) ------------------------------------------------------------------------
    def calc_store_likelihood(self, hp_vec, always_calc_cond=False, calc_grad=True):
        hp_vec = np.atleast_1d(hp_vec).ravel()
        if not np.array_equal(hp_vec, self._last_hp_vec):
            ln_lkd_val, ln_lkd_grad, cond_val, cond_grad, b_chofac_good = self._objective_at(hp_vec, always_calc_cond, calc_grad)
            self._last_hp_vec = hp_vec.copy()
            self._last_chofac_good = bool(b_chofac_good)
            self._lkd_val, self._lkd_grad = ln_lkd_val, ln_lkd_grad
            self._cond_val = np.nan if cond_val is None else cond_val
            self._cond_grad = cond_grad
        return self._lkd_val, self._lkd_grad, self._cond_val, self._cond_grad

    def _objective_at(self, hp_vec, always_calc_cond=False, calc_grad=True):
        """(ln_lkd, d ln_lkd / d hp_vec, cond, d cond / d hp_vec, Cholesky ok) at one optimiser vector -- the body of
        calc_store_likelihood (OptzLkd.py:47-83), without the memo."""
        hp_vals = self.hp_vec2dataclass(self.hp_info_optz_lkd, hp_vec)
        calc_cond = self.b_use_cond_cstr or always_calc_cond                   # OptzLkd.py:51
        self._skip_beta_grad = True                                            # hp_beta_grad is not read here
        try:
            lkd_info, b_chofac_good = self.calc_lkd_all(hp_vals, calc_lkd=True, calc_cond=calc_cond, calc_grad=calc_grad)
        finally:
            self._skip_beta_grad = False
        cond_val, cond_grad = lkd_info.cond, lkd_info.cond_grad
        if b_chofac_good:
            ln_lkd_val = lkd_info.ln_lkd
            ln_lkd_grad = lkd_info.ln_lkd_grad
            if calc_grad:
                bvec = self.hp_info_optz_lkd.bvec_log_optz
                transformation = 10 ** hp_vec[bvec] * np.log(10)             # log10 chain rule, OptzLkd.py:65-73
                ln_lkd_grad[bvec] *= transformation
                if self.b_use_cond_cstr and cond_grad is not None:
                    cond_grad[bvec] *= transformation
        else:
            # OptzLkd.py:74-77: a failed Cholesky makes minus the condition number the objective (and minus its
            # gradient, WITHOUT the log10 chain rule, the slope), so that SLSQP walks back into the region where
            # the matrix can be factorised.  calc_lkd_all computes both for the failed matrix up to N = 4096
            # (host SVD / eigenvectors of the downloaded matrix); above that, and for 'precon' (where the
            # reference has no cond_grad and stops with a TypeError), a large finite penalty / zero slope.
            n_hp = self.hp_info_optz_lkd.n_hp
            if cond_val is None or not np.isfinite(cond_val):
                cond_val = self.cond_max_abs
            cond_grad = np.zeros(n_hp) if cond_grad is None else cond_grad
            ln_lkd_val = -cond_val
            ln_lkd_grad = -cond_grad
        return ln_lkd_val, ln_lkd_grad, cond_val, cond_grad, b_chofac_good

    def _objective_rows(self, X):
        """Value and gradient (with respect to the optimiser vector: log10 chain rule, OptzLkd.py:65-70) at every row of X
        by ONE batched device call; rows whose Cholesky failed go through the one-row path (condition-number objective)."""
        X = np.atleast_2d(np.asarray(X, dtype=float))
        ln, grad, ok = self.calc_lkd_grad_batch(X)
        bvec = self.hp_info_optz_lkd.bvec_log_optz
        out = []
        for i in range(X.shape[0]):
            if ok[i]:
                g = grad[i].copy()
                g[bvec] *= 10 ** X[i, bvec] * np.log(10)
                out.append((ln[i], g))
            else:
                v, g = self._objective_at(X[i])[:2]
                out.append((v, np.asarray(g, dtype=float)))
        return out

    def return_optz_val(self, hp_vec):
        return -self.calc_store_likelihood(np.atleast_1d(hp_vec).ravel())[0]

    def return_optz_grad(self, hp_vec):
        return -self.calc_store_likelihood(np.atleast_1d(hp_vec).ravel())[1]

    # nonlinear condition-number constraint cond <= cond_max of the methods without the preconditioner (OptzLkd.py:102-114,
    # GaussianProcess.py:210-212)
    def return_cond_val(self, hp_vec):
        return self.calc_store_likelihood(np.atleast_1d(hp_vec).ravel())[2]

    def return_cond_grad(self, hp_vec):
        return self.calc_store_likelihood(np.atleast_1d(hp_vec).ravel())[3]

    # ---- multi-start SLSQP (OptzLkd.py:185-333) ------------------------------------------------------------------
    def optz_hp_max_lkd(self, hp_x0_all, optz_bound):
        if self.optz_mtd == 'SLSQP':
            optz_opt = {'ftol': self.optz_tol_obj, 'eps': self.optz_tol_x, 'maxiter': self.optz_iter_max, 'disp': False}
        elif self.optz_mtd == 'trust-constr':
            optz_opt = {'initial_tr_radius': 0.1, 'xtol': self.optz_tol_x, 'gtol': self.optz_tol_obj,
                        'maxiter': self.optz_iter_max, 'disp': False}
        else:
            raise Exception(f'Unknown optz_mtd: {self.optz_mtd}')
        hp_x0_all = np.atleast_2d(np.asarray(hp_x0_all, dtype=float))
        n_optz = hp_x0_all.shape[0]
        all_optz_success = np.full(n_optz, False, dtype=bool)
        all_total_fun_iter = np.full(n_optz, np.nan)
        optz_obj_all = np.full(n_optz, np.nan)
        optz_sol_all = np.full((n_optz, self.hp_info_optz_lkd.n_hp), np.nan)
        # the starts are independent: with torch.distributed up, rank r runs its contiguous block of them on its own
        # GPU and one all_gather shares (objective, success, iterations, solution) -- SURVEY.md 8e applied to the
        # optimiser's multi-start (the reference runs them one after the other, OptzLkd.py:252-290)
        from .multistart import _group_info, gather_rows, shard_rows
        group = self.restart_group                                # None: all starts on this process
        world, rank, _ = _group_info(group)
        lo, hi = shard_rows(n_optz, world, rank)
        all_con_good = np.full(n_optz, True, dtype=bool)
        optz_cond_all = np.full(n_optz, np.nan)
        # per-start counters of OptzLkd.py:255-259 (kept per start so that they travel with the gathered table)
        cho_fail_all, cond2big_all, init_cond_all = np.zeros(n_optz), np.zeros(n_optz), np.full(n_optz, np.nan)
        nlc = []
        if self.b_use_cond_cstr:                                                 # OptzLkd.py:245, GaussianProcess.py:210-212
            nlc = NonlinearConstraint(self.return_cond_val, -np.inf, self.cond_max, jac=self.return_cond_grad)
        err = None
        lockstep = self.optz_lockstep and not self.b_use_cond_cstr and (hi - lo) > 1 and self.optz_mtd == 'SLSQP'
        cls = type(self)
        if lockstep and (cls.return_optz_val is not HparaOptz.return_optz_val or cls.return_optz_grad is not HparaOptz.return_optz_grad) \
                and cls._objective_rows is HparaOptz._objective_rows:
            lockstep = False          # a subclass replaced the one-row objective only: keep calling that one
        if lockstep:
            err = self._run_starts_lockstep(hp_x0_all, lo, hi, optz_bound, optz_opt, optz_sol_all, optz_obj_all, all_optz_success,
                                            all_total_fun_iter)
        try:
            for i in range(lo, hi if not lockstep else lo):
                x0_i = hp_x0_all[i, :]
                if self.b_use_cond_cstr:                                         # OptzLkd.py:255-259
                    self._last_hp_vec = np.full((1, x0_i.size), np.nan)
                    self.calc_store_likelihood(x0_i)
                    init_cond_all[i] = self._cond_val
                    cho_fail_all[i] = float(not self._last_chofac_good)
                    cond2big_all[i] = float(self._cond_val > self.cond_max)
                self._last_hp_vec = np.full((1, x0_i.size), np.nan)
                res = minimize(self.return_optz_val, x0_i, method=self.optz_mtd, jac=self.return_optz_grad,
                               bounds=optz_bound, constraints=nlc, options=optz_opt)
                optz_sol_all[i, :] = res.x
                optz_obj_all[i] = res.fun
                all_optz_success[i] = res.success
                all_total_fun_iter[i] = res.nit
                if self.b_use_cond_cstr:                                         # OptzLkd.py:276-279
                    optz_cond_all[i] = self.return_cond_val(res.x)
                    all_con_good[i] = optz_cond_all[i] < 1.01 * self.cond_max
                if not (res.success and all_con_good[i]):
                    print(f"Surr hpara optz: Con {'GOOD' if all_con_good[i] else 'FAIL'}, Optimizer: {'GOOD' if res.success else res.message}")
        except Exception as e:          # noqa: BLE001 -- a failing rank still joins the collective below, then re-raises
            err = e
        if group is not None:
            table = gather_rows(np.column_stack((optz_obj_all, all_optz_success, all_total_fun_iter, all_con_good, optz_cond_all,
                                                 cho_fail_all, cond2big_all, init_cond_all, optz_sol_all))[lo:hi], n_optz,
                                group=group, error=err)
            optz_obj_all, all_optz_success, all_total_fun_iter = table[:, 0], table[:, 1] > 0.5, table[:, 2]
            all_con_good, optz_cond_all = table[:, 3] > 0.5, table[:, 4]
            cho_fail_all, cond2big_all, init_cond_all, optz_sol_all = table[:, 5], table[:, 6], table[:, 7], table[:, 8:]
        elif err is not None:
            raise err
        n_cho_fail, n_cond2big = int(np.nansum(cho_fail_all)), int(np.nansum(cond2big_all))
        max_init_cond = np.nan if np.all(np.isnan(init_cond_all)) else float(np.nanmax(init_cond_all))
        if np.any(all_con_good):                                                 # OptzLkd.py:294-305
            obj_ok, sol_ok = optz_obj_all[all_con_good], optz_sol_all[all_con_good, :]
        else:
            print('*** No solutions satisfy the constraints for the GP hyperparameter optimization ***')
            print(f'Cond = {optz_cond_all}')
            obj_ok, sol_ok = optz_obj_all, optz_sol_all
        if np.all(np.isnan(obj_ok)):
            raise RuntimeError('hyperparameter optimisation: every start ended with a NaN objective '
                               f'(objectives {optz_obj_all}); no solution can be selected')
        idx_min = np.nanargmin(obj_ok)
        best_hp = sol_ok[idx_min, :]
        surr_optz_info = {'hp_optz_success': np.mean(all_optz_success), 'hp_optz_iter_mean': np.mean(all_total_fun_iter),
                          'hp_optz_iter_max': np.max(all_total_fun_iter), 'hp_optz_con_good': np.mean(all_con_good),
                          'optz_n_cho_fail': n_cho_fail, 'optz_n_cond2big': n_cond2big, 'optz_max_init_cond': max_init_cond}
        self.optz_obj_all_last, self.optz_sol_all_last = optz_obj_all, optz_sol_all
        return best_hp, self._final_cond(best_hp), surr_optz_info

    optz_calc_final_cond = True   # False: skip it where nothing needs it ('precon': the value only goes into Kcov_cond_all)

    def _final_cond(self, best_hp):
        """Condition number of the matrix the selected hyperparameters give, for every well-conditioning method
        (OptzLkd.py:324-331).  Two Lanczos runs through the factor (~50 ms at N = 2500): with 'precon' the optimisation itself does not
        use it, optz_calc_final_cond = False then stores NaN instead."""
        if not self.optz_calc_final_cond and not self.b_use_cond_cstr:
            return np.nan
        return self.calc_lkd_all(self.hp_vec2dataclass(self.hp_info_optz_lkd, best_hp), calc_cond=True)[0].cond

    def _run_starts_lockstep(self, hp_x0_all, lo, hi, optz_bound, optz_opt, sol, obj, success, nit):
        """Starts [lo, hi) as concurrent SLSQP runs served by batched value + gradient calls (_LockstepObjective).
        Fills the result arrays in place; returns the first exception raised by a run (or None)."""
        ev = _LockstepObjective(self._objective_rows, hi - lo)
        errors = {}

        def run(i):
            memo = {'x': None, 'v': None, 'g': None}

            def at(x):
                x = np.atleast_1d(x).ravel()
                if memo['x'] is None or not np.array_equal(x, memo['x']):
                    memo['v'], memo['g'] = ev.request(i, x)
                    memo['x'] = x.copy()
                return memo
            try:
                res = minimize(lambda x: -at(x)['v'], hp_x0_all[i, :], method='SLSQP', jac=lambda x: -at(x)['g'],
                               bounds=optz_bound, constraints=[], options=optz_opt)
                sol[i, :], obj[i], success[i], nit[i] = res.x, res.fun, res.success, res.nit
                if not res.success:
                    print(f'Surr hpara optz: Con GOOD, Optimizer: {res.message}')
            except Exception as e:      # noqa: BLE001
                errors[i] = e
            finally:
                ev.finished(i)
        threads = [threading.Thread(target=run, args=(i,), name=f'slsqp-start-{i}') for i in range(lo, hi)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        self.optz_lockstep_batches, self.optz_lockstep_rows = ev.n_batches, ev.n_rows
        self._last_hp_vec = None
        return errors[min(errors)] if errors else None

    # ---- driver (GpHparaOptz.py:140-218) -------------------------------------------------------------------------
    def get_init_hp_vals(self):
        theta = self.hp_theta_init * np.ones(self.dim)
        beta = np.zeros(self.n_beta_coeff)
        if self.n_beta_coeff > 0:
            beta[0] = np.mean(self.get_scl_eval_data()[0])                    # GpHparaOptz.py:201,206
        hp_var_fval = None if self.known_eps_fval else self.hp_var_fval_init
        hp_var_fgrad = None if (self.use_grad is False) or self.known_eps_fgrad else self.hp_var_fgrad_init
        return self.make_hp_class(beta, theta, self.hp_kernel_default, self.hp_varK_init, hp_var_fval, hp_var_fgrad)

    def optz_hp_max_lkd_mtd_rescale(self, i_optz, hp_x0, optz_bound):
        """OptzLkd.py:114-185: the rescale methods' outer loop.  After the multi-start optimisation, up to cond_vreq_max_iter
        proposals of an anisotropic scaling are made from the optimised theta (rescaling_data_w_theta_sol); each but the last
        is followed by a re-optimisation from the isotropic theta it predicts.  The proposal closest to the line
        theta_1 = ... = theta_d wins: its scale vector goes into DataScl (the device data are re-sent) and its isotropic theta
        replaces the optimised one.  As in the reference, the scaled data stay as they are while the loop runs."""
        assert 'rescale' in self.wellcond_mtd, \
            f'Method should only be called for the rescale methods, but it is wellcond_mtd = {self.wellcond_mtd}'
        best_hp, cond_val, surr_optz_info = self.optz_hp_max_lkd(hp_x0, optz_bound)
        if self.n_eval <= 1:
            return best_hp, cond_val, surr_optz_info
        max_iter = self.cond_vreq_max_iter
        idx_theta = self.hp_info_optz_lkd.idx_theta
        theta_all = np.full((max_iter, self.dim), np.nan)
        dist2line_all = np.full(max_iter, np.nan)
        scale_vec_all = np.full((max_iter, self.dim), np.nan)
        theta_x0 = best_hp[idx_theta]
        for cnt in range(max_iter):
            theta_x0, dist2line, scale_new = self.rescaling_data_w_theta_sol(self.DataScl.x_scl, self.DataScl.xvec_scale, theta_x0)
            theta_all[cnt], dist2line_all[cnt], scale_vec_all[cnt] = theta_x0, dist2line, scale_new
            if cnt == max_iter - 1 or dist2line < self.cond_vreq_iter_tol:
                break
            x0 = np.copy(best_hp)
            x0[idx_theta] = theta_x0
            best_hp, cond_val = self.optz_hp_max_lkd(x0, optz_bound)[:2]
        pick = int(np.nanargmin(dist2line_all))
        self.DataScl.set_xscale_data(xvec_scale_in=scale_vec_all[pick])
        best_hp_final = np.copy(best_hp)
        best_hp_final[idx_theta] = theta_all[pick]
        return best_hp_final, cond_val, surr_optz_info

    def optz_hp(self, i_optz):
        if self.n_eval <= self.hp_const_n_eval:
            hp_vals = self.get_init_hp_vals()
            surr_optz_info = None
            # GpHparaOptz.py:155-156: the condition number of the matrix these fixed hyperparameters give (the device needs the
            # factor for it, which the reference's dense SVD does not)
            cond_val = self.calc_all_K_w_chofac(None, hp_vals, calc_chofac=True, calc_cond=True)[4]
            time_hp_optz = time_chofac = time_pick_hp0 = 0
        else:
            self._time_chofac = 0
            hp_x0, optz_bound, time_pick_hp0 = self.select_hp_optz_x0(i_optz, self.hp_info_optz_lkd)
            start_time = time.time()
            if ('rescale' in self.wellcond_mtd) and (self.cond_vreq_max_iter > 1):                    # GpHparaOptz.py:169-172
                hp_optz, cond_val, surr_optz_info = self.optz_hp_max_lkd_mtd_rescale(i_optz, hp_x0, optz_bound)
            else:
                hp_optz, cond_val, surr_optz_info = self.optz_hp_max_lkd(hp_x0, optz_bound)
            time_hp_optz = time.time() - start_time
            time_chofac = self._time_chofac
            hp_vals = self.hp_vec2dataclass(self.hp_info_optz_lkd, hp_optz)
            hp_vals = self.optz_closed_form_hp(hp_vals)
        self.store_new_para_surr(i_optz, hp_vals, surr_optz_info, cond_val, time_hp_optz, time_chofac, time_pick_hp0)
