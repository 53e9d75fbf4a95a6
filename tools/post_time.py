import sys, time, numpy as np
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench, gpgradpy_amd
n, d = 2000, 8
X, f, g, tab = bench.make_workload(n, d)
GP = gpgradpy_amd.GaussianProcess(d, True, 'SqExp', 'precon')
GP.set_data(X, f, np.zeros(n), g, np.zeros((n, d)))
hp = GP.hp_vec2dataclass(GP.hp_info_optz_lkd, tab[0])
hp = GP.optz_closed_form_hp(hp)
for rep in range(2):
    t0 = time.perf_counter(); GP.set_hpara('set', 0, hp_vals=hp); t1 = time.perf_counter()
    print('setup_eval_model (factor + alpha): %.1f ms' % ((t1 - t0) * 1e3))
xq = np.random.default_rng(0).uniform(-2, 2, (64, d))
for nx in (1, 64):
    GP.eval_model(xq[:nx])
    t0 = time.perf_counter(); GP.eval_model(xq[:nx]); t1 = time.perf_counter()
    print('eval_model nx=%d: %.1f ms' % (nx, (t1 - t0) * 1e3))
    t0 = time.perf_counter(); GP.eval_model(xq[:nx], calc_grad=True); t1 = time.perf_counter()
    print('eval_model nx=%d with gradients: %.1f ms' % (nx, (t1 - t0) * 1e3))
t0 = time.perf_counter(); GP.eval_model(xq[0], calc_grad=True, calc_hess=True, squeeze_nx=True); t1 = time.perf_counter()
print('eval_model with Hessians (1 point): %.1f ms' % ((t1 - t0) * 1e3))
