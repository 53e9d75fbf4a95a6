"""Randomised stress run of the non-'precon' well-conditioning methods and the round-2 additions (diagnostic; GPU box): 'base', the
four data-rescaling methods, the row-sum nugget, caller-supplied noise vectors, direct against adjoint gradient, batched against
single calls, posterior before / after likelihood calls, and the oracle on the scaled data.    python tests/stress/stress_wellcond.py [seconds] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import gpgradpy_amd
from oracle import gp_oracle as orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
n_cases = n_checks = n_fail = 0
METHODS = ('base', 'rescale_origin', 'rescale_eta_vary', 'dflt_vmin', 'dflt_vmax', 'precon')
while time.time() < t_end:
    d = int(rng.integers(1, 6))
    n = int(rng.integers(3, 60))
    kernel = ('SqExp', 'Ma5f2', 'RatQu')[int(rng.integers(0, 3))]
    noise = ('none', 'known', 'unknown')[int(rng.integers(0, 3))]
    wellcond = METHODS[int(rng.integers(0, len(METHODS)))]
    X, f, g = orc.synthetic_design(n, d, seed=int(rng.integers(0, 10 ** 6)))
    std_f = std_g = None
    if noise == 'none':
        std_f, std_g = np.zeros(n), np.zeros((n, d))
    elif noise == 'known':
        std_f, std_g = np.full(n, 1e-2), np.full((n, d), 1e-1)
    GP = gpgradpy_amd.GaussianProcess(d, True, kernel, wellcond)
    if wellcond == 'precon' and rng.random() < 0.5:
        GP.cond_eta_is_const = False                      # row-sum nugget of Kcor (Kernel.py:229-236)
    GP.set_data(X, f, std_f, g, std_g)
    hi = GP.hp_info_optz_lkd
    m = int(rng.integers(1, 6))
    rows = np.zeros((m, hi.n_hp))
    # scaled coordinates: distances are O(1) .. O(5) after 'set_vmin', tiny after 'set_vmax' -> theta accordingly
    lo, hi_t = (-2.5, -0.8) if wellcond != 'dflt_vmax' else (-0.5, 1.0)
    rows[:, hi.idx_theta] = rng.uniform(lo, hi_t, (m, d))
    if hi.has_kernel:
        rows[:, hi.idx_kernel] = rng.uniform(-0.3, 0.6, (m, 1))
    if hi.has_varK:
        rows[:, hi.idx_varK] = rng.uniform(-0.5, 0.5, m)
    if hi.has_var_fval:
        rows[:, hi.idx_var_fval] = rng.uniform(-5, -3, m)
    if hi.has_var_fgrad:
        rows[:, hi.idx_var_fgrad] = rng.uniform(-3, -1, m)
    ln_b = GP.calc_lkd_batch(rows)
    ln_g, grad_b, ok = GP.calc_lkd_grad_batch(rows)
    assert np.array_equal(np.isnan(ln_b), ~ok)
    if not ok.all():
        n_fail += 1
        continue
    assert np.allclose(ln_g, ln_b, rtol=1e-11)
    i = int(rng.integers(0, m))
    hp = GP.hp_vec2dataclass(hi, rows[i])
    hp_m = GP.optz_closed_form_hp(GP.hp_vec2dataclass(hi, rows[i]))
    GP.set_hpara('set', 0, hp_vals=hp_m)
    xq = rng.uniform(-2, 2, (int(rng.integers(1, 20)), d))
    mu0, sig0, dmu0 = GP.eval_model(xq, calc_grad=True)[:3]
    adj, good = GP.calc_lkd_all(hp, calc_grad=True, calc_cond=wellcond != 'precon')
    assert good and abs(adj.ln_lkd - ln_b[i]) <= 1e-11 * abs(ln_b[i])
    assert np.allclose(adj.ln_lkd_grad, grad_b[i], rtol=1e-8, atol=1e-8 * np.abs(grad_b[i]).max())
    dr = GP.calc_lkd_all(hp, calc_grad=True, lkd_use_adj_mtd=False)[0]
    assert np.allclose(dr.ln_lkd_grad, adj.ln_lkd_grad, rtol=1e-10, atol=1e-10 * np.abs(adj.ln_lkd_grad).max())
    assert dr.hp_beta_grad.shape == (1, hi.n_hp) and np.all(np.isfinite(dr.hp_beta_grad))
    # a caller's noise vector for one call, then everything as before
    nv2 = rng.uniform(1e-4, 1e-2, GP.n_data)
    varK_arg = hp.varK if GP.b_has_noisy_data else 1.0
    GP.calc_all_K_w_chofac(None, hp, noise_vec=nv2, varK=varK_arg)
    again = GP.calc_lkd_all(hp)[0].ln_lkd
    assert again == adj.ln_lkd
    mu1, sig1, dmu1 = GP.eval_model(xq, calc_grad=True)[:3]
    assert np.array_equal(mu0, mu1) and np.array_equal(sig0, sig1) and np.array_equal(dmu0, dmu1)
    n_checks += 8
    # oracle on the scaled data with the nugget that was used
    if GP.b_use_data_scl:
        Xs = GP.DataScl.x_scl
        fs, sfs, gs, sgs = GP.get_scl_eval_data()
    else:
        Xs, fs, sfs, gs, sgs = X, f, std_f, g, std_g
    kern_o = (kernel, float(np.ravel(hp.kernel)[0])) if kernel == 'RatQu' else kernel
    y = orc.make_data_vec(fs, gs)
    nv = orc.calc_noise_vec(n, d, True, sfs, sgs, hp.var_fval, hp.var_fgrad)
    noisy = noise != 'none'
    wc = 'precon' if wellcond == 'precon' else 'base'
    GP.calc_lkd_all(hp)                                   # sets _etaK_last for this hp
    r = orc.calc_lkd(Xs, y, hp.theta, kern_o, True, wc, GP._etaK_last, nv, noisy, varK=hp.varK)
    # (the absolute term: ln_lkd near zero is a difference of terms of size ~N, each good to ~1e-8 at cond ~1e10 -- tests/tolerances.py gives
    # ln det 2e-6 + 1e-9 N; seed 63 met 1.1e-6 on a value of 0.78)
    assert r.ok and abs(adj.ln_lkd - r.ln_lkd) <= 1e-6 * max(1.0, abs(r.ln_lkd)) + 4e-6, (kernel, noise, wellcond, n, d, adj.ln_lkd, r.ln_lkd)
    n_checks += 1
    n_cases += 1
    del GP
print(f'stress_wellcond: {n_cases} random cases ({n_fail} skipped: a factorisation failed), {n_checks} checks, all passed')
