#!/usr/bin/env python3
"""Golden vectors at the BASELINE.json configuration sizes, from the *reference* (marchildon/gpgradpy @ v2), plus the
variance-form posterior and the failed-Cholesky objective.

Runs ONLY in the build container (reference mounted read-only at /root/reference, imported with the two stubs of
gen_golden.py); only the .npz files travel.  Inputs are exactly SURVEY.md 8(d): rng = default_rng(0),
X = rng.uniform(-2, 2, (n, d)), Rosenbrock(a = 10) values and gradients, theta = 0.5 * 1_d.

  cfg1_full.npz   BASELINE configs[0]: gradient-free SqExp, n = 200, d = 2 ('precon' coerced to 'base')
  cfg2_full.npz   BASELINE configs[1]: gradient-enhanced SqExp, n = 500, d = 4 (N = 2500), noise-free, precon
                  scalars + alpha + mu / sig / gradients at 8 points + ln_lkd_grad, no N x N arrays
  evar_*.npz      eval_model_var (GpEvalModel.py:200-317): sig2, dsig2dx
  failobj_*.npz   calc_store_likelihood on a matrix whose Cholesky fails (OptzLkd.py:74-77): -cond, -cond_grad

Usage:  python tests/golden/gen_golden_cfg.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402


def evar_case(GaussianProcess, name, n, d, kernel, noise, use_grad, seed, wellcond='precon', nq=6):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-2, 2, (n, d))
    f, g = gg.rosenbrock(x)
    theta = 10.0 ** rng.uniform(-2.0, -0.5, d)
    xq = rng.uniform(-2, 2, (nq, d))
    xq[0] = x[0]                                          # a training point: sig2 ~ 0 (noise-free), gradient still defined
    if noise == 'known':
        std_f = 1e-2 * (1 + rng.uniform(0, 1, n))
        std_g = 1e-1 * (1 + rng.uniform(0, 1, (n, d)))
        varK = 2.5
    else:
        std_f, std_g, varK = np.zeros(n), np.zeros((n, d)), None
    GP = GaussianProcess(d, use_grad, kernel, wellcond)
    if use_grad:
        GP.set_data(x, f, std_f, g, std_g)
    else:
        GP.set_data(x, f, std_f)
    hp = GP.make_hp_class(theta=theta, kernel=GP.hp_kernel_default, varK=varK if GP.b_has_noisy_data else None)
    hp = GP.optz_closed_form_hp(hp)
    GP.set_hpara('set', 0, hp_vals=hp)
    sig2, dsig2dx, _ = GP.eval_model_var(xq, calc_grad=True)
    mu, sig = GP.eval_model(xq)[:2]
    return dict(name=name, n=n, d=d, kernel=kernel, noise=noise, use_grad=use_grad, wellcond=GP.wellcond_mtd, x=x, f=f,
                g=g if use_grad else np.zeros((0, d)), std_f=std_f, std_g=std_g if use_grad else np.zeros((0, d)),
                theta=theta, varK=float(hp.varK), beta=np.asarray(hp.beta, dtype=float),
                hp_kernel=np.nan if hp.kernel is None else float(hp.kernel), xq=xq, sig2=sig2, dsig2dx=dsig2dx, mu=mu, sig=sig)


def failobj_case(GaussianProcess):
    """The chofail fixture's setup (gen_golden.py: 'base', near-duplicate points, nugget 1e-30): what the optimiser's
    objective returns there."""
    n, d = 20, 2
    rng = np.random.default_rng(16)
    x = rng.uniform(-2, 2, (n, d))
    x[1] = x[0] + 1e-6 * rng.standard_normal(d)
    f, g = gg.rosenbrock(x)
    GP = GaussianProcess(d, True, 'SqExp', 'base')
    GP.set_data(x, f, np.zeros(n), g, np.zeros((n, d)))
    GP._etaK = GP._eta_Kgrad = 1e-30
    hp_vec = np.log10(np.array([1e-4, 1e-4]))
    GP._last_hp_vec = np.full((1, 2), np.nan)
    val, grad, cond, cond_grad = GP.calc_store_likelihood(hp_vec)
    assert val == -cond
    return dict(name='failobj_SqExp_n20_d2', x=x, f=f, g=g, etaK=1e-30, hp_vec=hp_vec, obj=float(val), obj_grad=np.asarray(grad, dtype=float),
                cond=float(cond), cond_grad=np.asarray(cond_grad, dtype=float))


def main():
    GaussianProcess = gg._import_reference()
    c1 = gg.make_case(GaussianProcess, name='cfg1_full', n=200, d=2, kernel='SqExp', noise='none', use_grad=False, seed=0,
                      theta=0.5 * np.ones(2))
    c2 = gg.make_case(GaussianProcess, name='cfg2_full', n=500, d=4, kernel='SqExp', noise='none', seed=0,
                      theta=0.5 * np.ones(4))
    for c in (c1, c2):
        np.savez_compressed(os.path.join(HERE, c['name'] + '.npz'), **c)
        print(f"{c['name']:12s} ok={c['b_chofac_good']} ln_lkd={c['ln_lkd']:.12e}")
    for kw in (dict(name='evar_SqExp_none_n20_d3', n=20, d=3, kernel='SqExp', noise='none', use_grad=True, seed=31),
               dict(name='evar_Ma5f2_known_n25_d2', n=25, d=2, kernel='Ma5f2', noise='known', use_grad=True, seed=32),
               dict(name='evar_RatQu_none_n30_d2_nograd', n=30, d=2, kernel='RatQu', noise='none', use_grad=False, seed=33)):
        c = evar_case(GaussianProcess, **kw)
        np.savez_compressed(os.path.join(HERE, c['name'] + '.npz'), **c)
        print(f"{c['name']:32s} sig2 min {c['sig2'].min():.3e} max {c['sig2'].max():.3e}")
    c = failobj_case(GaussianProcess)
    np.savez_compressed(os.path.join(HERE, c['name'] + '.npz'), **c)
    print(f"{c['name']} obj={c['obj']:.6e} grad={c['obj_grad']}")


if __name__ == '__main__':
    main()
