#!/usr/bin/env python3
"""Condition numbers of the golden cases from the *reference*: LkdInfo.cond of calc_lkd_all(..., calc_cond=True)
(Kernel.py:239-245 for 'precon': np.linalg.cond(Kcov_precon, 2); :279-285 for 'base': np.linalg.cond(Kcov, 2)),
on the inputs stored in the existing fixtures.  Writes tests/golden/cond_table.npz (names, cond).
Runs only in the build container (reference mounted at /root/reference).

Usage:  python tests/golden/gen_golden_cond.py
"""
import glob
import os

import numpy as np

import gen_golden as gg


def main():
    GaussianProcess = gg._import_reference()
    names, conds, grads = [], [], {}
    skip = ("multistart_", "nugget_", "optz_", "hess_", "cond_", "micro_")
    for path in sorted(glob.glob(os.path.join(gg.HERE, "*.npz"))):
        base = os.path.basename(path)
        if base.startswith(skip):
            continue
        z = np.load(path, allow_pickle=False)
        if int(z["n_data"]) > 700:
            continue
        failed_case = not bool(z["b_chofac_good"])
        d, use_grad, kernel = int(z["d"]), bool(z["use_grad"]), str(z["kernel"])
        GP = GaussianProcess(d, use_grad, kernel, str(z["wellcond"]) if use_grad else "base")
        std_f = None if z["std_f"].size == 0 else z["std_f"]
        std_g = None if z["std_g"].size == 0 else z["std_g"]
        bvec = z["bvec_use_grad"].astype(bool)
        if use_grad:
            GP.set_data(z["x"], z["f"], std_f, z["g"], std_g, None if bvec.all() else bvec)
        else:
            GP.set_data(z["x"], z["f"], std_f)
        if not np.isnan(z["pnlt"][0]):
            continue
        noisy = bool(z["b_has_noisy_data"])
        nanv = lambda k: None if np.isnan(float(z[k])) else float(z[k])
        hp = GP.make_hp_class(theta=z["theta"], kernel=float(z["hp_kernel"]) if "hp_kernel" in z.files else GP.hp_kernel_default,
                              varK=float(z["varK_in"]) if noisy else None, var_fval=nanv("var_fval"), var_fgrad=nanv("var_fgrad"))
        if failed_case:                           # Cholesky failure: the reference reports cond(Kcov) from the matrix (CalcLkd.py:330-333)
            GP._etaK = GP._eta_Kgrad = float(z["etaK"])
            lkd, ok = GP.calc_lkd_all(hp, calc_lkd=True, calc_cond=False, calc_grad=False)
            assert not ok
            names.append(base[:-4])
            conds.append(float(lkd.cond))
            print(f"{base[:-4]:36s} cond = {lkd.cond:.10e} (failed factorisation)")
            continue
        lkd, ok = GP.calc_lkd_all(hp, calc_lkd=True, calc_cond=True, calc_grad=False)
        assert ok and np.isclose(lkd.ln_lkd, float(z["ln_lkd"]), rtol=1e-10), base
        if GP.wellcond_mtd != 'precon':          # gradient of the condition number (GpHparaCon.py:163-207; not with 'precon')
            lkd_g, ok_g = GP.calc_lkd_all(hp, calc_lkd=True, calc_cond=True, calc_grad=True)
            assert ok_g and np.isclose(lkd_g.cond, lkd.cond, rtol=1e-12)
            grads[base[:-4]] = np.real(np.asarray(lkd_g.cond_grad, dtype=complex)).astype(float)
            print(f"{base[:-4]:36s} cond_grad = {grads[base[:-4]]}")
        names.append(base[:-4])
        conds.append(float(lkd.cond))
        print(f"{base[:-4]:36s} cond = {lkd.cond:.10e}")
    np.savez_compressed(os.path.join(gg.HERE, "cond_table.npz"), names=np.array(names), cond=np.array(conds),
                        **{"grad_" + k: v for k, v in grads.items()})
    fro_table(GaussianProcess)


def fro_table(GaussianProcess):
    """cond_norm = 'fro' (GaussianProcess.py:104): np.linalg.cond(., 'fro') of Kernel.py:239-245 / 279-285 for a subset of the
    cases, and calc_cond_fronorm_w_grad (GpHparaCon.py:209-236) for the 'base' ones -> tests/golden/cond_fro_table.npz."""
    names, conds, grads = [], [], {}
    for base in ("SqExp_none_n17_d4", "Ma5f2_known_n17_d4", "SqExp_unknown_n17_d4", "RatQu_known_n17_d4", "SqExp_none_n12_d2_base",
                 "RatQu_none_n12_d2_base", "SqExp_none_n50_d2_nograd", "Ma5f2_known_n50_d2_nograd", "SqExp_none_n64_d8", "Ma5f2_known_n130_d8"):
        z = np.load(os.path.join(gg.HERE, base + ".npz"), allow_pickle=False)
        d, use_grad, kernel = int(z["d"]), bool(z["use_grad"]), str(z["kernel"])
        GP = GaussianProcess(d, use_grad, kernel, str(z["wellcond"]) if use_grad else "base")
        GP.cond_norm = 'fro'
        std_f = None if z["std_f"].size == 0 else z["std_f"]
        std_g = None if z["std_g"].size == 0 else z["std_g"]
        if use_grad:
            GP.set_data(z["x"], z["f"], std_f, z["g"], std_g)
        else:
            GP.set_data(z["x"], z["f"], std_f)
        noisy = bool(z["b_has_noisy_data"])
        nanv = lambda k: None if np.isnan(float(z[k])) else float(z[k])
        hp = GP.make_hp_class(theta=z["theta"], kernel=float(z["hp_kernel"]) if "hp_kernel" in z.files else GP.hp_kernel_default,
                              varK=float(z["varK_in"]) if noisy else None, var_fval=nanv("var_fval"), var_fgrad=nanv("var_fgrad"))
        lkd, ok = GP.calc_lkd_all(hp, calc_lkd=True, calc_cond=True, calc_grad=False)
        assert ok
        names.append(base)
        conds.append(float(lkd.cond))
        print(f"{base:36s} cond_fro = {lkd.cond:.10e}")
        if GP.wellcond_mtd != 'precon':
            lkd_g, ok_g = GP.calc_lkd_all(hp, calc_lkd=True, calc_cond=True, calc_grad=True)
            assert ok_g and np.isclose(lkd_g.cond, lkd.cond, rtol=1e-4)        # two routes to K^-1 at cond ~ 1e10
            grads[base] = np.asarray(lkd_g.cond_grad, dtype=float)
            print(f"{base:36s} cond_fro_grad = {grads[base]}")
    np.savez_compressed(os.path.join(gg.HERE, "cond_fro_table.npz"), names=np.array(names), cond=np.array(conds),
                        **{"grad_" + k: v for k, v in grads.items()})


if __name__ == "__main__":
    main()
