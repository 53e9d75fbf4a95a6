#!/usr/bin/env python3
"""Golden vectors for the hyperparameter-derivative entries of the reference's kernel table and their compositions
(marchildon/gpgradpy @ v2, import recipe of gen_golden.py): calc_KernBase_grad_th, calc_KernGrad_grad_th, calc_Kern*_grad_alpha
(RatQu), calc_KernGrad_hp (noise-free path, 'precon' and 'base'), calc_Kcov_grad_hp (noisy path).  Small cases: the tensors are
[n_hp, N, N].

Usage:  python tests/golden/gen_golden_kgrad.py        # rewrites tests/golden/kgrad_*.npz
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
    theta = 10.0 ** rng.uniform(-1.0, 0.2, d)
    noisy = GP.b_has_noisy_data
    hp = GP.make_hp_class(theta=theta, kernel=GP.hp_kernel_default, varK=1.7 if noisy else None,
                          var_fval=None if GP.known_eps_fval else 1e-3, var_fgrad=None if GP.known_eps_fgrad else 1e-2)
    Rt = GP.get_scl_x_w_dist()[1]
    out = dict(name=name, n=n, d=d, kernel=kernel, noise=noise, wellcond=GP.wellcond_mtd, x=x, f=f, g=g,
               std_f=np.full(n, np.nan) if std_f is None else std_f, std_g=np.full((n, d), np.nan) if std_g is None else std_g,
               theta=theta, varK_in=1.7 if noisy else np.nan, var_fval_in=np.nan if hp.var_fval is None else hp.var_fval,
               var_fgrad_in=np.nan if hp.var_fgrad is None else hp.var_fgrad, hp_kernel=np.nan if hp.kernel is None else float(hp.kernel),
               base_grad_th=GP.calc_KernBase_grad_th(Rt, theta, hp.kernel), grad_grad_th=GP.calc_KernGrad_grad_th(Rt, theta, hp.kernel))
    if GP.kernel_has_hp:
        out.update(base_grad_alpha=GP.calc_KernBase_grad_alpha(Rt, theta, hp.kernel), grad_grad_alpha=GP.calc_KernGrad_grad_alpha(Rt, theta, hp.kernel))
    if noisy:
        Kern = GP.calc_Kern(Rt, theta, hp.kernel, None, None)
        out.update(Kern=Kern, Kcov_grad_hp=GP.calc_Kcov_grad_hp(GP.hp_info_optz_lkd, hp, Kern, Rt))
    else:
        out.update(KernGrad_hp=GP.calc_KernGrad_hp(GP.hp_info_optz_lkd, hp, Rt))
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print(name, {k: np.shape(v) for k, v in out.items() if isinstance(v, np.ndarray) and v.ndim == 3})


def hessx_case(GaussianProcess, name, n1, n2, d, kernel, seed):
    """x-derivative entries: calc_KernBase_hess_x / calc_KernGrad_grad_x of two different point sets."""
    rng = np.random.default_rng(seed)
    x1, x2 = rng.uniform(-1.5, 1.5, (n1, d)), rng.uniform(-1.5, 1.5, (n2, d))
    theta = 10.0 ** rng.uniform(-0.8, 0.3, d)
    GP = GaussianProcess(d, True, kernel, 'precon')
    a = GP.hp_kernel_default
    Rt = GP.calc_Rtensor(x1, x2, 1)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), name=name, kernel=kernel, x1=x1, x2=x2, theta=theta,
                        hp_kernel=np.nan if a is None else float(a), base_hess_x=GP.calc_KernBase_hess_x(Rt, theta, a),
                        grad_grad_x=GP.calc_KernGrad_grad_x(Rt, theta, a))
    print(name, GP.calc_KernBase_hess_x(Rt, theta, a).shape, GP.calc_KernGrad_grad_x(Rt, theta, a).shape)


def main():
    GaussianProcess = gg._import_reference()
    hessx_case(GaussianProcess, 'kgrad_hessx_SqExp_d3', 4, 5, 3, 'SqExp', 81)
    hessx_case(GaussianProcess, 'kgrad_hessx_Ma5f2_d2', 5, 3, 2, 'Ma5f2', 82)
    hessx_case(GaussianProcess, 'kgrad_hessx_RatQu_d2', 3, 6, 2, 'RatQu', 83)
    case(GaussianProcess, 'kgrad_SqExp_none_n5_d2_precon', 5, 2, 'SqExp', 'none', 'precon', 71)
    case(GaussianProcess, 'kgrad_Ma5f2_none_n4_d3_base', 4, 3, 'Ma5f2', 'none', 'base', 72)
    case(GaussianProcess, 'kgrad_RatQu_none_n5_d2_precon', 5, 2, 'RatQu', 'none', 'precon', 73)
    case(GaussianProcess, 'kgrad_Ma5f2_known_n4_d2_precon', 4, 2, 'Ma5f2', 'known', 'precon', 74)
    case(GaussianProcess, 'kgrad_RatQu_unknown_n4_d2_base', 4, 2, 'RatQu', 'unknown', 'base', 75)
    case(GaussianProcess, 'kgrad_SqExp_unknown_n5_d1_precon', 5, 1, 'SqExp', 'unknown', 'precon', 76)


if __name__ == '__main__':
    main()
