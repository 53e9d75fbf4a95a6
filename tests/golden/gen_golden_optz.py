#!/usr/bin/env python3
"""Golden vectors for the hyperparameter-optimisation driver, captured by running the
*reference* (marchildon/gpgradpy @ v2) in the build container.  Same import recipe as gen_golden.py; the one
difference is the `smt.sampling_methods.LHS` stub, which here forwards to `gpgradpy_amd.hpara_optz.lhs_sample`
(SciPy's Latin hypercube, seed = random_state) so that the reference's `select_hp_optz_x0` runs end to end --
smt itself is not installable offline, its sample sequence is therefore NOT what these fixtures hold.

Stored per case: inputs, the history the start-point selection reads, the reference's start rows, box bounds,
chosen start row, optimised hyperparameter vector and the closed-form (beta, varK) at the optimum.

Usage:  python tests/golden/gen_golden_optz.py        # rewrites tests/golden/optz_*.npz
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("GPGRAD_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from gen_golden import rosenbrock  # noqa: E402


def _import_reference():
    nb = types.ModuleType("numba")

    def jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    nb.jit = jit
    sys.modules["numba"] = nb
    smt = types.ModuleType("smt")
    sm = types.ModuleType("smt.sampling_methods")

    class LHS:
        def __init__(self, xlimits, random_state=1, **k):
            self.xlimits, self.seed = np.asarray(xlimits, dtype=float), random_state

        def __call__(self, n):
            from gpgradpy_amd.hpara_optz import lhs_sample
            return lhs_sample(self.xlimits, n, seed=self.seed)

    sm.LHS = LHS
    smt.sampling_methods = sm
    sys.modules["smt"] = smt
    sys.modules["smt.sampling_methods"] = sm
    sys.path.insert(0, REF)
    from gpgradpy.src.GaussianProcess import GaussianProcess  # noqa: E402
    return GaussianProcess


def run_case(GaussianProcess, name, n, d, kernel, noise, seed, n_hist=2, start_mtd='hp_best', n_best=12, wellcond='precon'):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-2, 2, (n, d))
    f, g = rosenbrock(x)
    if noise == 'none':
        std_f, std_g = np.zeros(n), np.zeros((n, d))
    elif noise == 'known':
        std_f, std_g = np.full(n, 1e-2), np.full((n, d), 1e-1)
    else:
        std_f = std_g = None
    GP = GaussianProcess(d, True, kernel, wellcond)
    GP.lkd_optz_start_mtd = start_mtd
    GP.lkd_hp_best_n_eval = n_best
    GP.optz_n_x0 = 3
    GP.init_optz_surr(n_hist + 2)
    # iteration 0: a single evaluation -> the initial hyperparameters are stored (GpHparaOptz.py:152-157)
    GP.set_data(x[:1], f[:1], None if std_f is None else std_f[:1], g[:1], None if std_g is None else std_g[:1])
    GP.set_hpara('optz', 0)
    # iterations 1 .. n_hist: growing data set, each optimised from the median of the stored history
    sizes = np.linspace(max(4, n // 2), n, n_hist).astype(int)
    out = {}
    for it, ni in enumerate(sizes, start=1):
        GP.set_data(x[:ni], f[:ni], None if std_f is None else std_f[:ni], g[:ni], None if std_g is None else std_g[:ni])
        hp_x0_all, bounds = GP.get_hp_x0_lhs_median(it, GP.hp_info_optz_lkd, n_best if start_mtd == 'hp_best' else 3)
        hp_x0_sel, _, _ = GP.select_hp_optz_x0(it, GP.hp_info_optz_lkd)
        if GP.b_use_data_scl:                                  # objective, constraint and their gradients at the start rows
            vals = [GP.calc_store_likelihood(np.copy(r)) for r in hp_x0_sel]
            out[f'it{it}_x0_obj'] = np.array([v[0] for v in vals], dtype=float)
            out[f'it{it}_x0_grad'] = np.array([v[1] for v in vals], dtype=float)
            out[f'it{it}_x0_cond'] = np.array([v[2] for v in vals], dtype=float)
            out[f'it{it}_x0_cond_grad'] = np.array([np.real(v[3]) for v in vals], dtype=float)
        GP.set_hpara('optz', it)
        hv = GP.hp_vals
        vec = GP.hp_theta_all[it]
        out[f'it{it}_n'] = ni
        out[f'it{it}_hp_x0_all'] = hp_x0_all
        out[f'it{it}_box_lb'] = bounds.lb
        out[f'it{it}_box_ub'] = bounds.ub
        out[f'it{it}_hp_x0_sel'] = hp_x0_sel
        out[f'it{it}_theta'] = np.array(hv.theta, dtype=float)
        out[f'it{it}_varK'] = float(hv.varK)
        out[f'it{it}_beta'] = np.array(hv.beta, dtype=float)
        out[f'it{it}_var_fval'] = np.nan if hv.var_fval is None else float(hv.var_fval)
        out[f'it{it}_var_fgrad'] = np.nan if hv.var_fgrad is None else float(hv.var_fgrad)
        out[f'it{it}_kernel'] = np.nan if hv.kernel is None else float(np.asarray(hv.kernel, dtype=float).reshape(-1)[0])
        hp_final = GP.make_hp_class(theta=hv.theta, kernel=hv.kernel, varK=hv.varK if GP.b_has_noisy_data else None,
                                    var_fval=hv.var_fval, var_fgrad=hv.var_fgrad)
        out[f'it{it}_ln_lkd'] = GP.calc_lkd_all(hp_final)[0].ln_lkd
        out[f'it{it}_iter_max'] = GP.hp_optz_iter_max[it]
        out[f'it{it}_success'] = GP.hp_optz_success[it]
        if GP.b_use_data_scl:                                  # rescale methods: the scaling the outer loop ended with (OptzLkd.py:176)
            out[f'it{it}_xvec_scale'] = np.array(GP.DataScl.xvec_scale, dtype=float)
        print(f"{name} it{it}: n={ni} theta={hv.theta} varK={hv.varK:.6e} ln_lkd={out[f'it{it}_ln_lkd']:.10e} nit={GP.hp_optz_iter_max[it]}")
    out.update(wellcond=wellcond, cond_hist=GP.Kcov_cond_all[:n_hist + 1], con_good_hist=GP.hp_optz_con_good[:n_hist + 1])
    out.update(name=name, n=n, d=d, kernel=kernel, noise=noise, x=x, f=f, g=g, n_hist=n_hist, start_mtd=start_mtd, n_best=n_best,
               std_f=np.full(n, np.nan) if std_f is None else std_f, std_g=np.full((n, d), np.nan) if std_g is None else std_g,
               theta_hist0=GP.hp_theta_all[0], varK_hist0=GP.hp_varK_all[0], kernel_hist0=GP.hp_kernel_all[0])
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)


def main():
    GaussianProcess = _import_reference()
    if len(sys.argv) > 1 and sys.argv[1] == 'base':       # 'base' well-conditioning: SLSQP with the condition-number constraint
        run_case(GaussianProcess, 'optz_SqExp_none_n12_d2_base', 12, 2, 'SqExp', 'none', seed=37, wellcond='base')
        run_case(GaussianProcess, 'optz_Ma5f2_known_n10_d2_base_lhs', 10, 2, 'Ma5f2', 'known', seed=38, wellcond='base', start_mtd='lhs')
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'rescale':    # data-rescaling methods: constraint + the outer rescale loop (OptzLkd.py:114-185)
        run_case(GaussianProcess, 'optz_SqExp_none_n12_d2_rescale_origin', 12, 2, 'SqExp', 'none', seed=41, wellcond='rescale_origin', start_mtd='lhs')
        run_case(GaussianProcess, 'optz_Ma5f2_known_n10_d2_rescale_eta_vary', 10, 2, 'Ma5f2', 'known', seed=42, wellcond='rescale_eta_vary', start_mtd='lhs')
        run_case(GaussianProcess, 'optz_SqExp_none_n10_d3_dflt_vmin', 10, 3, 'SqExp', 'none', seed=43, wellcond='dflt_vmin', start_mtd='lhs')
        run_case(GaussianProcess, 'optz_SqExp_none_n12_d2_rescale_eta_vary', 12, 2, 'SqExp', 'none', seed=44, wellcond='rescale_eta_vary', start_mtd='lhs')
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'ratqu':      # only the rational quadratic case (kernel with its own hyperparameter)
        run_case(GaussianProcess, 'optz_RatQu_none_n12_d2', 12, 2, 'RatQu', 'none', seed=36)
        return
    run_case(GaussianProcess, 'optz_SqExp_none_n10_d1', 10, 1, 'SqExp', 'none', seed=31)
    run_case(GaussianProcess, 'optz_SqExp_none_n14_d3', 14, 3, 'SqExp', 'none', seed=32)
    run_case(GaussianProcess, 'optz_Ma5f2_known_n12_d2', 12, 2, 'Ma5f2', 'known', seed=33)
    run_case(GaussianProcess, 'optz_SqExp_unknown_n16_d2', 16, 2, 'SqExp', 'unknown', seed=34)
    run_case(GaussianProcess, 'optz_Ma5f2_none_n12_d2_lhs', 12, 2, 'Ma5f2', 'none', seed=35, start_mtd='lhs')


if __name__ == '__main__':
    main()
