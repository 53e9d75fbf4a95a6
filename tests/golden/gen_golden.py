#!/usr/bin/env python3
"""Generate golden input/output vectors by running the *reference* (marchildon/gpgradpy @ v2).

Runs ONLY in the build container, where the reference is mounted read-only at /root/reference.
Nothing from the reference travels: this script imports it in-process (with two stub modules for
the absent `numba` and `smt` packages, as recorded in SURVEY.md §8c), evaluates the hot path
(`set_data` -> `calc_lkd_all` -> `optz_closed_form_hp` -> `set_hpara('set')` -> `eval_model`) on
seeded synthetic inputs and stores inputs + outputs as small .npz fixtures next to this file.

Stub 1: `numba.jit` -> identity decorator (every jitted reference function is plain NumPy code).
Stub 2: `smt.sampling_methods.LHS` -> raises if called (never reached: restart tables are explicit).

Usage:  python tests/golden/gen_golden.py            # rewrites tests/golden/*.npz
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("GPGRAD_REFERENCE", "/root/reference")


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

    class LHS:  # pragma: no cover - must never be reached
        def __init__(self, **k):
            raise RuntimeError("smt.LHS is not available offline (SURVEY.md 8c)")

    sm.LHS = LHS
    smt.sampling_methods = sm
    sys.modules["smt"] = smt
    sys.modules["smt.sampling_methods"] = sm
    sys.path.insert(0, REF)
    from gpgradpy.src.GaussianProcess import GaussianProcess  # noqa: E402

    return GaussianProcess


def rosenbrock(x, a=10.0):
    """Rosenbrock value and gradient; formula as the reference's demo (plt/plt_cond.py:54-85)."""
    n, d = x.shape
    f = np.zeros(n)
    g = np.zeros((n, d))
    if d == 1:
        return (1 - x[:, 0]) ** 2 + a * x[:, 0] ** 4, (-2 * (1 - x[:, 0]) + 4 * a * x[:, 0] ** 3)[:, None]
    for k in range(d - 1):
        f += a * (x[:, k + 1] - x[:, k] ** 2) ** 2 + (1 - x[:, k]) ** 2
        g[:, k] += -4 * a * x[:, k] * (x[:, k + 1] - x[:, k] ** 2) - 2 * (1 - x[:, k])
        g[:, k + 1] += 2 * a * (x[:, k + 1] - x[:, k] ** 2)
    return f, g


def make_case(GaussianProcess, name, n, d, kernel, noise, use_grad=True, wellcond='precon', seed=0,
              theta=None, varK=None, var_fval=None, var_fgrad=None, nq=8, near_dup=False, etaK=None,
              store_mats=False, pnlt=None, mask=False, hp_kernel=None):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-2, 2, (n, d))
    if near_dup and n > 2:
        x[1] = x[0] + 1e-6 * rng.standard_normal(d)
    f, g = rosenbrock(x)
    if theta is None:
        theta = 10.0 ** rng.uniform(-2.5, -0.5, d)
    theta = np.asarray(theta, dtype=float)
    xq = rng.uniform(-2, 2, (nq, d))

    if noise == 'none':
        std_f = np.zeros(n)
        std_g = np.zeros((n, d))
    elif noise == 'known':
        std_f = 1e-2 * (1 + rng.uniform(0, 1, n))
        std_g = 1e-1 * (1 + rng.uniform(0, 1, (n, d)))
        f = f + std_f * rng.standard_normal(n)
        g = g + std_g * rng.standard_normal((n, d))
    elif noise == 'known_cfg5':   # BASELINE cfg5: constant known noise levels, data left as they are
        std_f = np.full(n, 1e-2)
        std_g = np.full((n, d), 1e-1)
        noise = 'known'
    elif noise == 'unknown':
        std_f = None
        std_g = None
        f = f + 1e-2 * rng.standard_normal(n)
        g = g + 1e-1 * rng.standard_normal((n, d))
    else:
        raise ValueError(noise)

    bvec = None
    if mask:                                  # bvec_use_grad: drop ~1/3 of the gradients, the last one for sure
        bvec = rng.random(n) > 0.35
        bvec[-1] = False
        bvec[0] = True
        g = g[bvec]
        if std_g is not None:
            std_g = std_g[bvec]

    GP = GaussianProcess(d, use_grad, kernel, wellcond)
    if use_grad:
        GP.set_data(x, f, std_f, g, std_g, bvec)
    else:
        GP.set_data(x, f, std_f)
    if etaK is not None:
        GP._etaK = etaK
        GP._eta_Kgrad = etaK
    if pnlt is not None:                      # varK penalty of CalcLkd.py:118-133 (off by default)
        GP.lkd_varK_pnlt_use = True
        GP.lkd_varK_pnlt_c1, GP.lkd_varK_pnlt_c2 = pnlt

    noisy = bool(GP.b_has_noisy_data)
    if noisy and varK is None:
        varK = 10.0 ** rng.uniform(-1, 1)
    if noise == 'unknown':
        var_fval = 1e-4 if var_fval is None else var_fval
        var_fgrad = (1e-2 if var_fgrad is None else var_fgrad) if use_grad else None
    hp = GP.make_hp_class(theta=theta, kernel=GP.hp_kernel_default if hp_kernel is None else hp_kernel,
                          varK=varK if noisy else None,
                          var_fval=var_fval, var_fgrad=var_fgrad)

    lkd, ok = GP.calc_lkd_all(hp, calc_lkd=True, calc_cond=False, calc_grad=False)
    out = dict(name=name, n=n, d=d, kernel=kernel, noise=noise, use_grad=use_grad, wellcond=GP.wellcond_mtd,
               x=x, f=f, g=g if use_grad else np.zeros((0, d)),
               std_f=np.array([]) if std_f is None else std_f,
               std_g=np.array([]) if (std_g is None or not use_grad) else std_g,
               theta=theta, varK_in=np.nan if varK is None else varK,
               var_fval=np.nan if var_fval is None else var_fval,
               var_fgrad=np.nan if var_fgrad is None else var_fgrad,
               etaK=GP._etaK, b_has_noisy_data=noisy, b_chofac_good=bool(ok), xq=xq,
               n_data=GP.n_data, bvec_use_grad=np.ones(n, dtype=bool) if bvec is None else bvec, pnlt=np.array([np.nan, np.nan] if pnlt is None else pnlt))
    if hp.kernel is not None:                 # kernels with a hyperparameter of their own (RatQu: alpha)
        out['hp_kernel'] = float(hp.kernel)
    if not ok:
        return out

    if not mask:
        lkd_g, ok_g = GP.calc_lkd_all(hp, calc_lkd=True, calc_cond=False, calc_grad=True)
        out['ln_lkd_grad'] = np.asarray(lkd_g.ln_lkd_grad, dtype=float)
    out.update(hp_beta=np.asarray(lkd.hp_beta, dtype=float),
               hp_varK=np.nan if lkd.hp_varK is None else float(lkd.hp_varK),
               ln_det_Kmat=float(lkd.ln_det_Kmat), ln_lkd=float(lkd.ln_lkd))

    # matrices of the evaluation (reference Kernel.py:140-307)
    x_scl, Rtensor = GP.get_scl_x_w_dist()
    if noisy:
        Kern, Kcor, Kcov, chofac, _, etaK_used, _ = GP.calc_all_K_w_chofac(Rtensor, hp)
        varK_mat = hp.varK
    else:
        Kern, Kcor, Kcov, chofac, _, etaK_used, _ = GP.calc_Kern_w_chofac(Rtensor, hp)
        varK_mat = 1.0
    noise_vec = GP.calc_noise_vec(hp)
    out['noise_vec'] = noise_vec
    lower = bool(chofac[1])
    out['chofac_lower'] = lower
    out['chofac_diag'] = np.diag(chofac[0]).copy()
    if GP.wellcond_mtd == 'precon':
        out['pvec'] = np.sqrt(np.diag(Kern + np.diag(noise_vec / varK_mat)))
    if store_mats:
        out['Kern'] = Kern
        out['Kcov'] = Kcov
        out['chofac'] = np.tril(chofac[0]) if lower else np.triu(chofac[0])

    # posterior: reference loop GpHparaOptz.py:220-230 then GaussianProcess.py:365-395
    hp2 = GP.optz_closed_form_hp(hp)
    GP.set_hpara('set', 0, hp_vals=hp2)
    out['alpha'] = np.asarray(GP.invKernEta_fdiff, dtype=float)
    out['varK_model'] = float(hp2.varK)
    mu, sig, dmudx, dsigdx = GP.eval_model(xq, calc_grad=True)[:4]
    out['mu'] = mu
    out['sig'] = sig
    out['dmudx'] = dmudx
    out['dsigdx'] = dsigdx
    return out


def main():
    GaussianProcess = _import_reference()
    cases = []
    add = lambda **k: cases.append(make_case(GaussianProcess, **k))

    # worked micro-example of SURVEY.md 8c (theta = 0.5 -> P = I)
    cases.append(_micro(GaussianProcess))

    idx = 0
    for kernel in ('SqExp', 'Ma5f2'):
        for noise in ('none', 'known', 'unknown'):
            for (n, d) in ((2, 1), (5, 2), (17, 4), (64, 8)):
                idx += 1
                add(name=f'{kernel}_{noise}_n{n}_d{d}', n=n, d=d, kernel=kernel, noise=noise,
                    seed=100 + idx, store_mats=(n * (d + 1) <= 90))
    # theta spread over decades (P far from I), odd sizes not multiple of any tile
    add(name='SqExp_none_n33_d3_spread', n=33, d=3, kernel='SqExp', noise='none', seed=7,
        theta=np.array([1e-3, 3e-1, 4.0]))
    add(name='Ma5f2_known_n33_d3_spread', n=33, d=3, kernel='Ma5f2', noise='known', seed=8,
        theta=np.array([2e-3, 1e-1, 2.0]), varK=3.7)
    # near-duplicate points: ill-conditioned, rescued by the preconditioner + nugget
    add(name='SqExp_none_n20_d2_neardup', n=20, d=2, kernel='SqExp', noise='none', seed=9, near_dup=True)
    add(name='Ma5f2_none_n20_d2_neardup', n=20, d=2, kernel='Ma5f2', noise='none', seed=10, near_dup=True)
    # larger single cases (N = 1000 / 1170) to exercise several Cholesky panels
    add(name='SqExp_none_n200_d4', n=200, d=4, kernel='SqExp', noise='none', seed=11)
    add(name='Ma5f2_known_n130_d8', n=130, d=8, kernel='Ma5f2', noise='known', seed=12)
    # gradient-free (BASELINE cfg1 shape, reduced n) : wellcond coerced to 'base', upper cho_factor
    add(name='SqExp_none_n50_d2_nograd', n=50, d=2, kernel='SqExp', noise='none', use_grad=False, seed=13,
        theta=np.array([0.5, 0.5]), store_mats=True)
    add(name='Ma5f2_known_n50_d2_nograd', n=50, d=2, kernel='Ma5f2', noise='known', use_grad=False, seed=14,
        theta=np.array([0.3, 0.7]), varK=2.0)
    # gradient-enhanced with wellcond_mtd='base' (no preconditioner)
    add(name='SqExp_none_n12_d2_base', n=12, d=2, kernel='SqExp', noise='none', wellcond='base', seed=15,
        theta=np.array([0.4, 0.9]), store_mats=True)
    # gradient masks (bvec_use_grad): value path of KernelSqExp.py:349-377 / KernelMatern5f2.py:380-417
    add(name='SqExp_none_n17_d4_mask', n=17, d=4, kernel='SqExp', noise='none', seed=21, mask=True, store_mats=True)
    add(name='Ma5f2_known_n23_d3_mask', n=23, d=3, kernel='Ma5f2', noise='known', seed=22, mask=True, store_mats=True)
    add(name='SqExp_unknown_n40_d5_mask', n=40, d=5, kernel='SqExp', noise='unknown', seed=23, mask=True)
    # varK penalty active (c2 small so that varK > c2 * var(f))
    add(name='SqExp_none_n17_d4_pnlt', n=17, d=4, kernel='SqExp', noise='none', seed=17, pnlt=(0.7, 1e-6))
    # Cholesky failure: 'base' method, near-duplicate points, tiny nugget -> cho_factor raises
    add(name='SqExp_none_n20_d2_chofail', n=20, d=2, kernel='SqExp', noise='none', wellcond='base', seed=16,
        near_dup=True, etaK=1e-30, theta=np.array([1e-4, 1e-4]))

    for c in cases:
        path = os.path.join(HERE, c['name'] + '.npz')
        np.savez_compressed(path, **c)
        print(f"{c['name']:36s} ok={c['b_chofac_good']} ln_lkd={c.get('ln_lkd', float('nan')):.12e}")

    # multi-start table (row a17, BASELINE cfg4 shape reduced): n=64, d=4, 64 restarts, noise-free SqExp
    rng = np.random.default_rng(0)
    n, d, m = 64, 4, 64
    x = rng.uniform(-2, 2, (n, d))
    f, g = rosenbrock(x)
    hp_x0 = np.random.default_rng(1).uniform(-2.5, -0.5, (m, d))  # log10 theta
    GP = GaussianProcess(d, True, 'SqExp', 'precon')
    GP.set_data(x, f, np.zeros(n), g, np.zeros((n, d)))
    ln_lkd_all = np.full(m, np.nan)
    for i in range(m):
        hp = GP.hp_vec2dataclass(GP.hp_info_optz_lkd, hp_x0[i])
        info, ok = GP.calc_lkd_all(hp)
        if ok:
            ln_lkd_all[i] = info.ln_lkd
    np.savez_compressed(os.path.join(HERE, 'multistart_SqExp_n64_d4.npz'), x=x, f=f, g=g, hp_x0=hp_x0,
                        ln_lkd_all=ln_lkd_all, idx_max=int(np.nanargmax(ln_lkd_all)), etaK=GP._etaK,
                        n_hp=GP.hp_info_optz_lkd.n_hp, bvec_log_optz=GP.hp_info_optz_lkd.bvec_log_optz)
    print('multistart: argmax', int(np.nanargmax(ln_lkd_all)), 'max', np.nanmax(ln_lkd_all))

    # noisy multi-start (cfg5 shape reduced): Ma5f2, known noise, rows [log10 theta, log10 varK]
    n, d, m = 40, 6, 16
    rng = np.random.default_rng(3)
    x = rng.uniform(-2, 2, (n, d))
    f, g = rosenbrock(x)
    std_f = np.full(n, 1e-2)
    std_g = np.full((n, d), 1e-1)
    hp_rows = np.hstack((np.random.default_rng(2).uniform(-3, -1, (m, d)),
                         np.random.default_rng(4).uniform(-1, 1, (m, 1))))
    GP = GaussianProcess(d, True, 'Ma5f2', 'precon')
    GP.set_data(x, f, std_f, g, std_g)
    ln_lkd_all = np.full(m, np.nan)
    for i in range(m):
        hp = GP.hp_vec2dataclass(GP.hp_info_optz_lkd, hp_rows[i])
        info, ok = GP.calc_lkd_all(hp)
        if ok:
            ln_lkd_all[i] = info.ln_lkd
    np.savez_compressed(os.path.join(HERE, 'multistart_Ma5f2_noisy_n40_d6.npz'), x=x, f=f, g=g, std_f=std_f,
                        std_g=std_g, hp_x0=hp_rows, ln_lkd_all=ln_lkd_all, idx_max=int(np.nanargmax(ln_lkd_all)),
                        etaK=GP._etaK, n_hp=GP.hp_info_optz_lkd.n_hp,
                        bvec_log_optz=GP.hp_info_optz_lkd.bvec_log_optz)
    print('multistart noisy: argmax', int(np.nanargmax(ln_lkd_all)))

    # calc_nugget table (row a4): etaK for a grid of (kernel, n, d)
    rows = []
    for kernel in ('SqExp', 'Ma5f2'):
        for n in (1, 2, 10, 500, 2000, 4000):
            for d in (1, 2, 4, 8, 16):
                GP = GaussianProcess(d, True, kernel, 'precon')
                eb, eg = GP.calc_nugget(n)
                rows.append((0 if kernel == 'SqExp' else 1, n, d, eb, eg))
    np.savez_compressed(os.path.join(HERE, 'nugget_table.npz'), rows=np.array(rows))


def _micro(GaussianProcess):
    GP = GaussianProcess(1, True, 'SqExp', 'precon')
    x = np.array([[0.0], [1.0]])
    f = np.array([0.0, 1.0])
    g = np.array([[0.0], [2.0]])
    GP.set_data(x, f, np.zeros(2), g, np.zeros((2, 1)))
    hp = GP.make_hp_class(theta=np.array([0.5]))
    lkd, ok = GP.calc_lkd_all(hp)
    lkd_g = GP.calc_lkd_all(hp, calc_grad=True)[0]
    Kern, Kcor, Kcov, chofac, _, _, _ = GP.calc_Kern_w_chofac(GP.Rtensor_init, hp)
    hp2 = GP.optz_closed_form_hp(hp)
    GP.set_hpara('set', 0, hp_vals=hp2)
    xq = np.array([[0.4]])
    mu, sig, dmudx, dsigdx = GP.eval_model(xq, calc_grad=True)[:4]
    return dict(name='micro_d1', n=2, d=1, kernel='SqExp', noise='none', use_grad=True, wellcond='precon',
                x=x, f=f, g=g, std_f=np.zeros(2), std_g=np.zeros((2, 1)), theta=np.array([0.5]),
                varK_in=np.nan, var_fval=np.nan, var_fgrad=np.nan, etaK=GP._etaK, b_has_noisy_data=False,
                b_chofac_good=True, xq=xq, n_data=4, bvec_use_grad=np.ones(2, dtype=bool), pnlt=np.array([np.nan, np.nan]), hp_beta=lkd.hp_beta, hp_varK=lkd.hp_varK,
                ln_det_Kmat=lkd.ln_det_Kmat, ln_lkd=lkd.ln_lkd, noise_vec=np.zeros(4), chofac_lower=True,
                chofac_diag=np.diag(chofac[0]).copy(), pvec=np.ones(4), Kern=Kern, Kcov=Kcov,
                chofac=np.tril(chofac[0]), alpha=GP.invKernEta_fdiff, varK_model=hp2.varK, mu=mu, sig=sig, dmudx=dmudx, dsigdx=dsigdx, ln_lkd_grad=np.asarray(lkd_g.ln_lkd_grad, dtype=float))


if __name__ == '__main__':
    main()
