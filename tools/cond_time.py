"""Diagnostic: time of calc_lkd_all(calc_cond=True) at the headline size (Lanczos through the factor in HBM)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench, gpgradpy_amd
for n, d in ((500, 4), (2000, 8)):
    X, f, g, tab = bench.make_workload(n, d)
    GP = gpgradpy_amd.GaussianProcess(d, True, 'SqExp', 'precon')
    GP.set_data(X, f, np.zeros(n), g, np.zeros((n, d)))
    hp = GP.hp_vec2dataclass(GP.hp_info_optz_lkd, tab[0])
    GP.calc_lkd_all(hp, calc_cond=True)
    t0 = time.perf_counter(); info, ok = GP.calc_lkd_all(hp, calc_cond=True); t1 = time.perf_counter()
    t2 = time.perf_counter(); GP.calc_lkd_all(hp); t3 = time.perf_counter()
    print('n=%d d=%d: value + condition number %.1f ms (value only %.1f ms), cond = %.6e (cond_max_target %.1e)'
          % (n, d, (t1 - t0) * 1e3, (t3 - t2) * 1e3, info.cond, GP.cond_max_target))
