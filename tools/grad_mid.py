"""Diagnostic: value + gradient, one at a time, at mid sizes (64-tile factorisation, 128-tile inverse), with the result for an A/B of
GPG_OVERLAP_INVERSE."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench, gpgradpy_amd
for n, d in ((600, 8), (1000, 8), (1300, 8)):
    X, f, g, tab = bench.make_workload(n, d)
    GP = gpgradpy_amd.GaussianProcess(d, True, 'SqExp', 'precon')
    GP.set_data(X, f, np.zeros(n), g, np.zeros((n, d)))
    hp = GP.hp_vec2dataclass(GP.hp_info_optz_lkd, tab[0])
    GP.calc_lkd_all(hp, calc_grad=True)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); info, ok = GP.calc_lkd_all(hp, calc_grad=True); ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); GP.calc_lkd_all(hp); tv = time.perf_counter() - t0
    print('N = %5d: value + gradient %.2f ms (value %.2f ms)  ln_lkd %.12e  |grad| %.12e  fallbacks %d' %
          (n * (d + 1), min(ts) * 1e3, tv * 1e3, info.ln_lkd, np.linalg.norm(info.ln_lkd_grad), GP.factor_fallbacks()), flush=True)
    GP.close()
