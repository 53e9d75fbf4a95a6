#!/usr/bin/env python3
"""Golden vectors for the remaining outputs of the reference's likelihood-gradient code (marchildon/gpgradpy @ v2, run in the build
container with the import recipe of gen_golden.py): the direct (forward) method -- calc_lkd_all(..., lkd_use_adj_mtd=False),
CalcLkd.py:69-86, 237-242 -- with hp_beta_grad, hp_varK_grad, ln_det_Kmat_grad and its ln_lkd_grad, next to the adjoint method's
ln_lkd_grad (and hp_beta_grad, which the noisy path always fills: CalcLkd.py:216-217); and calc_Kern_precon
(KernelSqExp.py:590-605 etc.).

Usage:  python tests/golden/gen_golden_direct.py        # rewrites tests/golden/direct_*.npz, precon_table.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402


def case(GaussianProcess, name, n, d, kernel, noise, wellcond, seed):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-2, 2, (n, d))
    f, g = gg.rosenbrock(x)
    if noise == 'known':
        std_f, std_g = 1e-2 * (1 + rng.uniform(0, 1, n)), 1e-1 * (1 + rng.uniform(0, 1, (n, d)))
    elif noise == 'none':
        std_f, std_g = np.zeros(n), np.zeros((n, d))
    else:
        std_f = std_g = None
    GP = GaussianProcess(d, True, kernel, wellcond)
    GP.set_data(x, f, std_f, g, std_g)
    theta = 10.0 ** rng.uniform(-1.5, -0.3, d)
    noisy = GP.b_has_noisy_data
    hp = GP.make_hp_class(theta=theta, kernel=GP.hp_kernel_default, varK=2.0 if noisy else None,
                          var_fval=None if GP.known_eps_fval else 1e-3, var_fgrad=None if GP.known_eps_fgrad else 1e-2)
    adj = GP.calc_lkd_all(hp, calc_grad=True, lkd_use_adj_mtd=True)[0]
    dr = GP.calc_lkd_all(hp, calc_grad=True, lkd_use_adj_mtd=False)[0]
    nan = np.full(1, np.nan)
    out = dict(name=name, n=n, d=d, kernel=kernel, noise=noise, wellcond=GP.wellcond_mtd, x=x, f=f, g=g,
               std_f=np.full(n, np.nan) if std_f is None else std_f, std_g=np.full((n, d), np.nan) if std_g is None else std_g,
               theta=theta, varK_in=2.0 if noisy else np.nan, var_fval_in=np.nan if hp.var_fval is None else hp.var_fval,
               var_fgrad_in=np.nan if hp.var_fgrad is None else hp.var_fgrad, hp_kernel=np.nan if hp.kernel is None else float(hp.kernel),
               adj_ln_lkd=adj.ln_lkd, adj_ln_lkd_grad=adj.ln_lkd_grad,
               adj_hp_beta_grad=nan if adj.hp_beta_grad is None else np.asarray(adj.hp_beta_grad, dtype=float),
               dir_ln_lkd=dr.ln_lkd, dir_ln_lkd_grad=dr.ln_lkd_grad, dir_hp_beta_grad=np.asarray(dr.hp_beta_grad, dtype=float),
               dir_hp_varK_grad=nan if dr.hp_varK_grad is None else np.asarray(dr.hp_varK_grad, dtype=float),
               dir_ln_det_grad=np.asarray(dr.ln_det_Kmat_grad, dtype=float), dir_hp_varK=np.nan if dr.hp_varK is None else dr.hp_varK)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print(name, 'ln_lkd', adj.ln_lkd, dr.ln_lkd, '| grad diff', np.max(np.abs(adj.ln_lkd_grad - dr.ln_lkd_grad)), '| beta_grad', dr.hp_beta_grad)


def precon_table(GaussianProcess):
    rows = {}
    theta = np.linspace(2.5, 3, 3)
    for k in ('SqExp', 'Ma5f2', 'RatQu'):
        GP = GaussianProcess(3, True, k, 'precon')
        pvec, pinv, gvec = GP.calc_Kern_precon(4, 3, theta, calc_grad=True, b_return_vec=True)
        P, Pinv, gmat = GP.calc_Kern_precon(4, 3, theta, calc_grad=True, b_return_vec=False)
        rows.update({f'{k}_pvec': pvec, f'{k}_pinv': pinv, f'{k}_gvec': gvec, f'{k}_P': P, f'{k}_Pinv': Pinv, f'{k}_gmat': gmat})
    np.savez_compressed(os.path.join(HERE, 'precon_table.npz'), theta=theta, n_eval=4, n_grad=3, **rows)


def main():
    GaussianProcess = gg._import_reference()
    case(GaussianProcess, 'direct_SqExp_none_n12_d3', 12, 3, 'SqExp', 'none', 'precon', 61)
    case(GaussianProcess, 'direct_Ma5f2_known_n10_d2', 10, 2, 'Ma5f2', 'known', 'precon', 62)
    case(GaussianProcess, 'direct_RatQu_unknown_n10_d2', 10, 2, 'RatQu', 'unknown', 'precon', 63)
    case(GaussianProcess, 'direct_SqExp_known_n10_d2_base', 10, 2, 'SqExp', 'known', 'base', 64)
    case(GaussianProcess, 'direct_Ma5f2_none_n14_d2_base', 14, 2, 'Ma5f2', 'none', 'base', 65)
    precon_table(GaussianProcess)


if __name__ == '__main__':
    main()
