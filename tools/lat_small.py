"""Diagnostic: 300 one-at-a-time likelihood evaluations (value; value + gradient) on a 100-column problem, for a kernel trace."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench, gpgradpy_amd
n, d = 20, 4
X, f, g, _ = bench.make_workload(n, d)
GP = gpgradpy_amd.GaussianProcess(d, True, 'SqExp', 'precon')
GP.set_data(X, f, np.zeros(n), g, np.zeros((n, d)))
hp = GP.hp_vec2dataclass(GP.hp_info_optz_lkd, np.full(d, -1.0))
mode = sys.argv[1] if len(sys.argv) > 1 else 'value'
for _ in range(20):
    GP.calc_lkd_all(hp, calc_grad=mode == 'grad')
t0 = time.perf_counter()
for _ in range(300):
    GP.calc_lkd_all(hp, calc_grad=mode == 'grad')
print(mode, 'ms per call', (time.perf_counter() - t0) / 300 * 1e3)
