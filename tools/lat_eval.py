"""Diagnostic: 300 posterior evaluations at one point (with gradients) on a 500-column model, for a kernel trace."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench, gpgradpy_amd
n, d = 100, 4
X, f, g, _ = bench.make_workload(n, d)
GP = gpgradpy_amd.GaussianProcess(d, True, 'SqExp', 'precon')
GP.set_data(X, f, np.zeros(n), g, np.zeros((n, d)))
hp = GP.optz_closed_form_hp(GP.hp_vec2dataclass(GP.hp_info_optz_lkd, np.full(d, -1.0)))
GP.set_hpara('set', 0, hp_vals=hp)
xq = np.random.default_rng(0).uniform(-2, 2, (1, d))
grad = len(sys.argv) > 1 and sys.argv[1] == 'grad'
for _ in range(20):
    GP.eval_model(xq, calc_grad=grad)
t0 = time.perf_counter()
for _ in range(300):
    GP.eval_model(xq, calc_grad=grad)
print('grad' if grad else 'value', 'ms per call', (time.perf_counter() - t0) / 300 * 1e3)
