"""Randomised stress run of the less-travelled paths (diagnostic; GPU box): gradient-free models, bvec_use_grad masks, posterior
gradients / Hessians against central differences of the posterior itself, the same object taking data sets of different sizes one after
the other, kernel tables on random point sets.    python tests/stress/stress_misc.py [seconds] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import gpgradpy_amd
from oracle import gp_oracle as orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
n_cases = n_checks = 0
GPs = {}
while time.time() < t_end:
    d = int(rng.integers(1, 5))
    kernel = ('SqExp', 'Ma5f2', 'RatQu')[int(rng.integers(0, 3))]
    use_grad = bool(rng.random() < 0.7)
    key = (d, kernel, use_grad)
    if key not in GPs or rng.random() < 0.2:                  # mostly REUSE the object: new data, new sizes, old state must not leak
        GPs[key] = gpgradpy_amd.GaussianProcess(d, use_grad, kernel, 'precon')
    GP = GPs[key]
    n = int(rng.integers(2, 90))
    noise = ('none', 'known')[int(rng.integers(0, 2))]
    X, f, g = orc.synthetic_design(n, d, seed=int(rng.integers(0, 10 ** 6)))
    std_f = np.zeros(n) if noise == 'none' else np.full(n, 1e-2)
    mask = None
    if use_grad and rng.random() < 0.4:
        mask = rng.random(n) < 0.6
        if not mask.any():
            mask[0] = True
    if use_grad:
        gg = g if mask is None else g[mask]
        std_g = np.zeros(gg.shape) if noise == 'none' else np.full(gg.shape, 1e-1)
        GP.set_data(X, f, std_f, gg, std_g, bvec_use_grad=mask)
    else:
        GP.set_data(X, f, std_f)
    theta = 10.0 ** rng.uniform(-1.8, -0.3, d)
    a = 10.0 ** rng.uniform(-0.3, 0.6) if kernel == 'RatQu' else None
    hp = GP.make_hp_class(theta=theta, kernel=a, varK=1.3 if GP.b_has_noisy_data else None)
    info, good = GP.calc_lkd_all(hp)
    assert good
    # oracle
    kern_o = (kernel, a) if kernel == 'RatQu' else kernel
    y = orc.make_data_vec(f, (g if mask is None else g[mask]) if use_grad else None)
    std_g_full = None
    if use_grad:
        std_g_full = np.zeros((int(mask.sum()) if mask is not None else n, d)) if noise == 'none' else np.full((int(mask.sum()) if mask is not None else n, d), 1e-1)
    nv = orc.calc_noise_vec(n, d, use_grad, std_f, std_g_full, n_grad=None if mask is None else int(mask.sum()))
    r = orc.calc_lkd(X, y, theta, kern_o, use_grad, GP.wellcond_mtd, GP._etaK, nv, GP.b_has_noisy_data, varK=hp.varK, grad_mask=mask)
    assert r.ok and abs(info.ln_lkd - r.ln_lkd) <= 1e-6 * max(1.0, abs(r.ln_lkd)) + 4e-6, (key, n, noise, mask is not None, info.ln_lkd, r.ln_lkd)
    hp_m = GP.optz_closed_form_hp(hp)
    GP.set_hpara('set', 0, hp_vals=hp_m)
    xq = rng.uniform(-1.5, 1.5, (3, d))
    mu, sig, dmu, dsig = GP.eval_model(xq, calc_grad=True)[:4]
    eps = 1e-4      # central differences of a posterior whose matrix has cond ~ 1e10: the rounding noise of mu / sig divided by eps is
                    # part of every tolerance below
    for k in range(d):
        xp, xm = xq.copy(), xq.copy()
        xp[:, k] += eps
        xm[:, k] -= eps
        mp, sp = GP.eval_model(xp)[:2]
        mm, sm = GP.eval_model(xm)[:2]
        assert np.allclose(dmu[:, k], (mp - mm) / (2 * eps), rtol=1e-3, atol=1e-3 * max(1.0, np.abs(dmu).max()) + 3e-7 * max(1.0, np.abs(mu).max()) / eps), (key, n, 'dmu', dmu[:, k], (mp - mm) / (2 * eps))
        fd_s = (sp - sm) / (2 * eps)
        # sig = sqrt(varK (1 - k' K^-1 k)): the difference of two numbers of size 1 -- its rounding noise kappa * eps / (2 sig) divided by
        # eps is what a central difference of sig can resolve
        noise_s = 1e-6 * hp_m.varK / np.maximum(sig, 1e-300) / eps
        assert np.all(np.abs(dsig[:, k] - fd_s) <= 2e-3 * np.abs(fd_s) + 1e-4 * max(1e-3, np.abs(dsig).max()) + noise_s), (key, n, 'dsig', dsig[:, k], fd_s, sig)
    n_checks += 2 + 2 * d
    if mask is None or not use_grad:
        h = GP.eval_model(xq[0], calc_grad=True, calc_hess=True, squeeze_nx=True)
        for k in range(d):
            xp, xm = xq[0].copy(), xq[0].copy()
            xp[k] += eps
            xm[k] -= eps
            gp_ = GP.eval_model(xp, calc_grad=True, squeeze_nx=True)
            gm_ = GP.eval_model(xm, calc_grad=True, squeeze_nx=True)
            assert np.allclose(h[4][k], (gp_[2] - gm_[2]) / (2 * eps), rtol=1e-3, atol=1e-3 * max(1.0, np.abs(h[4]).max()) + 3e-6 * max(1.0, np.abs(dmu).max()) / eps), (key, n, 'd2mu', h[4][k], (gp_[2] - gm_[2]) / (2 * eps))
        n_checks += d
    # kernel table on two random point sets
    x2 = rng.uniform(-1.5, 1.5, (int(rng.integers(1, 12)), d))
    Rt = GP.calc_Rtensor(X[:min(n, 9)], x2, 1)
    Kb = GP.calc_KernBase(Rt, theta, a)
    assert np.allclose(Kb, orc.kern_base(X[:min(n, 9)], x2, theta, kern_o), rtol=1e-12, atol=1e-14)
    n_checks += 1
    n_cases += 1
print(f'stress_misc: {n_cases} random cases, {n_checks} checks, all passed')
