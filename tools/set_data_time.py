"""Diagnostic: cost of set_data when a Bayesian-optimisation loop adds one point per iteration (new context each time)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench, gpgradpy_amd
d = 4
X, f, g, _ = bench.make_workload(400, d)
GP = gpgradpy_amd.GaussianProcess(d, True, 'SqExp', 'precon')
for n0 in (20, 100, 300):
    ts, tl = [], []
    for n in range(n0, n0 + 6):
        t0 = time.perf_counter()
        GP.set_data(X[:n], f[:n], np.zeros(n), g[:n], np.zeros((n, d)))
        t1 = time.perf_counter()
        hp = GP.hp_vec2dataclass(GP.hp_info_optz_lkd, np.full(d, -1.0))
        GP.calc_lkd_all(hp, calc_grad=True)
        t2 = time.perf_counter()
        GP.calc_lkd_all(hp, calc_grad=True)
        t3 = time.perf_counter()
        ts.append((t1 - t0) * 1e3); tl.append(((t2 - t1) * 1e3, (t3 - t2) * 1e3))
    print(f'n = {n0}..{n0 + 5}: set_data ms', ' '.join('%.2f' % t for t in ts), '| first / second value+gradient call ms', ' '.join('%.2f/%.2f' % t for t in tl))
