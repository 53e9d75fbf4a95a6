#!/usr/bin/env python3
"""Golden vectors for the data-rescaling well-conditioning methods ('rescale_origin', 'rescale_eta_vary', 'dflt_vmin',
'dflt_vmax': GaussianProcess.py:84-89,342-361, base/Rescaling.py, GpWellCond.py:18-100), from the *reference*
(marchildon/gpgradpy @ v2) run in the build container with the import recipe of gen_golden.py.  Only the .npz files travel.

Per case: the inputs; what set_data derives (shift / scale of the parameter space and of the objective, scaled data,
nuggets, vmin); one likelihood evaluation with gradient, condition number and its gradient at a given hyperparameter
vector (with the nugget the matrix got -- the row-sum one for 'rescale_eta_vary'); the posterior with first derivatives
at 6 points and with Hessians at one, all in the caller's (unscaled) coordinates; and the anisotropic-scaling proposal
rescaling_data_w_theta_sol makes from that theta.

Usage:  python tests/golden/gen_golden_rescale.py        # rewrites tests/golden/rescale_*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402


def case(GaussianProcess, name, wellcond, n, d, kernel, noise, seed, aniso=None):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-2, 2, (n, d))
    f, g = gg.rosenbrock(x)
    if noise == 'known':
        std_f = 1e-2 * (1 + rng.uniform(0, 1, n))
        std_g = 1e-1 * (1 + rng.uniform(0, 1, (n, d)))
    elif noise == 'none':
        std_f, std_g = np.zeros(n), np.zeros((n, d))
    else:
        std_f = std_g = None
    GP = GaussianProcess(d, True, kernel, wellcond)
    GP.set_data(x, f, std_f, g, std_g)
    if aniso is not None:                      # what the rescale loop does at its end (OptzLkd.py:176)
        GP.DataScl.set_xscale_data(xvec_scale_in=np.asarray(aniso, dtype=float))
    S = GP.DataScl
    out = dict(name=name, wellcond=wellcond, n=n, d=d, kernel=kernel, noise=noise, x=x, f=f, g=g,
               std_f=np.full(n, np.nan) if std_f is None else std_f, std_g=np.full((n, d), np.nan) if std_g is None else std_g,
               aniso=np.full(d, np.nan) if aniso is None else np.asarray(aniso, dtype=float),
               x_shift=S.x_shift, xvec_scale=S.xvec_scale, obj_shift=float(S.obj_shift), obj_scale=float(S.obj_scale),
               x_scl=S.x_scl, obj_scl=S.obj_scl, grad_scl=S.grad_scl, etaK=float(GP._etaK), eta_Kbase=float(GP._eta_Kbase),
               vmin_init=float(GP._vmin_init), vmin_req_grad=float(GP._vmin_req_grad), cond_eta_is_const=bool(GP.cond_eta_is_const),
               b_use_cond_cstr=bool(GP.b_use_cond_cstr))
    theta = 10.0 ** rng.uniform(-2.0, -0.7, d)
    noisy = GP.b_has_noisy_data
    varK_in = 3.0 if noisy else None
    hp = GP.make_hp_class(theta=theta, kernel=GP.hp_kernel_default, varK=varK_in,
                          var_fval=None if GP.known_eps_fval else 1e-3, var_fgrad=None if GP.known_eps_fgrad else 1e-2)
    info, good = GP.calc_lkd_all(hp, calc_cond=True, calc_grad=True)
    assert good
    x_scl, Rt = GP.get_scl_x_w_dist()
    eta_used, idx_eta = GP.calc_all_K_w_chofac(Rt, hp, calc_chofac=False, varK=1.0 if not noisy else None)[5:7]
    out.update(theta=theta, varK_in=np.nan if varK_in is None else varK_in, var_fval_in=np.nan if hp.var_fval is None else hp.var_fval,
               var_fgrad_in=np.nan if hp.var_fgrad is None else hp.var_fgrad,
               hp_kernel=np.nan if hp.kernel is None else float(hp.kernel), ln_lkd=float(info.ln_lkd), ln_det=float(info.ln_det_Kmat),
               beta=np.asarray(info.hp_beta, dtype=float), varK=np.nan if info.hp_varK is None else float(info.hp_varK),
               ln_lkd_grad=np.asarray(info.ln_lkd_grad, dtype=float), cond=float(info.cond), cond_grad=np.asarray(info.cond_grad, dtype=float),
               eta_used=float(eta_used), idx_eta=-1 if idx_eta is None else int(idx_eta))
    hp2 = GP.optz_closed_form_hp(hp)
    GP.set_hpara('set', 0, hp_vals=hp2)
    xq = rng.uniform(-2, 2, (6, d))
    xq[0] = x[0]
    mu, sig, dmu, dsig = GP.eval_model(xq, calc_grad=True)[:4]
    h = GP.eval_model(xq[1], calc_grad=True, calc_hess=True, squeeze_nx=True)
    th_new, dist2, scale_new = GP.rescaling_data_w_theta_sol(S.x_scl, S.xvec_scale, np.log10(theta))
    out.update(xq=xq, mu=mu, sig=sig, dmudx=dmu, dsigdx=dsig, h_mu=h[0], h_sig=h[1], h_dmudx=h[2], h_dsigdx=h[3], h_d2mudx2=h[4],
               h_d2sigdx2=h[5], alpha=GP.invKernEta_fdiff, varK_model=float(hp2.varK), prop_theta=th_new, prop_dist2=float(dist2),
               prop_scale=scale_new)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print(f"{name}: xvec_scale={S.xvec_scale} obj_scale={S.obj_scale:.6e} etaK={GP._etaK:.6e} eta_used={eta_used:.6e} "
          f"ln_lkd={info.ln_lkd:.12e} cond={info.cond:.6e}")


def main():
    GaussianProcess = gg._import_reference()
    case(GaussianProcess, 'rescale_origin_SqExp_none_n14_d2', 'rescale_origin', 14, 2, 'SqExp', 'none', 51)
    case(GaussianProcess, 'rescale_origin_Ma5f2_known_n12_d3', 'rescale_origin', 12, 3, 'Ma5f2', 'known', 52)
    case(GaussianProcess, 'rescale_origin_SqExp_none_n16_d3_aniso', 'rescale_origin', 16, 3, 'SqExp', 'none', 53, aniso=[1.0, 2.5, 0.4])
    case(GaussianProcess, 'rescale_eta_vary_SqExp_none_n14_d2', 'rescale_eta_vary', 14, 2, 'SqExp', 'none', 54)
    case(GaussianProcess, 'rescale_eta_vary_Ma5f2_known_n12_d2', 'rescale_eta_vary', 12, 2, 'Ma5f2', 'known', 55)
    case(GaussianProcess, 'rescale_eta_vary_RatQu_unknown_n12_d2', 'rescale_eta_vary', 12, 2, 'RatQu', 'unknown', 56)
    case(GaussianProcess, 'rescale_dflt_vmin_RatQu_none_n12_d2', 'dflt_vmin', 12, 2, 'RatQu', 'none', 57)
    case(GaussianProcess, 'rescale_dflt_vmax_Ma5f2_none_n15_d4', 'dflt_vmax', 15, 4, 'Ma5f2', 'none', 58)


if __name__ == '__main__':
    main()
