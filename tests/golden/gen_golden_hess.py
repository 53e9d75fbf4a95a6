#!/usr/bin/env python3
"""Golden vectors for the posterior Hessians, captured by running the *reference* in the
build container (import recipe of gen_golden.py).  Per case: inputs, hyperparameters, and for a few query points
eval_model(calc_grad=True, calc_hess=True) one point per call: mu, sig, dmudx, dsigdx, d2mudx2, d2sigdx2.

Usage:  python tests/golden/gen_golden_hess.py       # rewrites tests/golden/hess_*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import _import_reference, rosenbrock  # noqa: E402


def run_case(GaussianProcess, name, n, d, kernel, noise, seed, use_grad=True, hp_kernel=None):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-2, 2, (n, d))
    f, g = rosenbrock(x)
    theta = 10.0 ** rng.uniform(-1.5, -0.3, d)
    std_f = np.zeros(n) if noise == 'none' else np.full(n, 1e-2)
    std_g = np.zeros((n, d)) if noise == 'none' else np.full((n, d), 1e-1)
    GP = GaussianProcess(d, use_grad, kernel, 'precon')
    if use_grad:
        GP.set_data(x, f, std_f, g, std_g)
    else:
        GP.set_data(x, f, std_f)
    varK_in = 2.5 if noise != 'none' else None
    hp = GP.make_hp_class(theta=theta, kernel=hp_kernel if hp_kernel is not None else GP.hp_kernel_default, varK=varK_in)
    hp = GP.optz_closed_form_hp(hp)
    GP.set_hpara('set', 0, hp_vals=hp)
    xq = rng.uniform(-2, 2, (4, d))
    xq[3] = x[0] + 1e-3                                        # close to a data point: small sig
    out = dict(name=name, n=n, d=d, kernel=kernel, noise=noise, use_grad=use_grad, x=x, f=f, g=g, std_f=std_f, std_g=std_g,
               theta=theta, varK_in=np.nan if varK_in is None else varK_in, varK=hp.varK, beta=hp.beta, xq=xq, etaK=GP._etaK)
    if hp.kernel is not None:
        out['hp_kernel'] = float(hp.kernel)
    res = [GP.eval_model(xq[i:i + 1], calc_grad=True, calc_hess=True) for i in range(xq.shape[0])]
    out['mu'] = np.array([r[0][0] for r in res])
    out['sig'] = np.array([r[1][0] for r in res])
    out['dmudx'] = np.array([r[2][0] for r in res])
    out['dsigdx'] = np.array([r[3][0] for r in res])
    out['d2mudx2'] = np.array([r[4][0] for r in res])
    out['d2sigdx2'] = np.array([r[5][0] for r in res])
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print(name, 'sig', out['sig'], '|d2mu|', np.abs(out['d2mudx2']).max(), '|d2sig|', np.abs(out['d2sigdx2']).max())


def main_ratqu(GaussianProcess):
    run_case(GaussianProcess, 'hess_RatQu_none_n14_d3', 14, 3, 'RatQu', 'none', 46, hp_kernel=1.7)
    run_case(GaussianProcess, 'hess_RatQu_known_n10_d2', 10, 2, 'RatQu', 'known', 47)
    run_case(GaussianProcess, 'hess_RatQu_none_n25_d2_nograd', 25, 2, 'RatQu', 'none', 48, use_grad=False, hp_kernel=0.6)


def main():
    GaussianProcess = _import_reference()
    if len(sys.argv) > 1 and sys.argv[1] == 'ratqu':      # only the rational quadratic cases
        main_ratqu(GaussianProcess)
        return
    run_case(GaussianProcess, 'hess_SqExp_none_n12_d2', 12, 2, 'SqExp', 'none', 41)
    run_case(GaussianProcess, 'hess_Ma5f2_none_n15_d3', 15, 3, 'Ma5f2', 'none', 42)
    run_case(GaussianProcess, 'hess_SqExp_known_n20_d4', 20, 4, 'SqExp', 'known', 43)
    run_case(GaussianProcess, 'hess_Ma5f2_known_n9_d1', 9, 1, 'Ma5f2', 'known', 44)
    run_case(GaussianProcess, 'hess_SqExp_none_n30_d2_nograd', 30, 2, 'SqExp', 'none', 45, use_grad=False)


if __name__ == '__main__':
    main()
