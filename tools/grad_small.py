"""Diagnostic: value + gradient evaluations at cfg2 size (run under rocprofv3 --kernel-trace --stats)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench, gpgradpy_amd
n, d = 500, 4
X, f, g, tab = bench.make_workload(n, d, "cfg2")
GP = gpgradpy_amd.GaussianProcess(d, True, 'SqExp', 'precon')
GP.set_data(X, f, np.zeros(n), g, np.zeros((n, d)))
hp = GP.hp_vec2dataclass(GP.hp_info_optz_lkd, tab[0])
GP.calc_lkd_all(hp, calc_grad=True)
t0 = time.perf_counter()
for _ in range(20):
    GP.calc_lkd_all(hp, calc_grad=True)
print('value + gradient: %.3f ms' % ((time.perf_counter() - t0) / 20 * 1e3))
