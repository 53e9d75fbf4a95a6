"""Diagnostic: posterior evaluation time against the number of query points at the headline size."""
import sys, time, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench, gpgradpy_amd
n, d = 2000, 8
X, f, g, tab = bench.make_workload(n, d)
GP = gpgradpy_amd.GaussianProcess(d, True, 'SqExp', 'precon')
GP.set_data(X, f, np.zeros(n), g, np.zeros((n, d)))
hp = GP.optz_closed_form_hp(GP.hp_vec2dataclass(GP.hp_info_optz_lkd, tab[0]))
GP.set_hpara('set', 0, hp_vals=hp)
xq = np.random.default_rng(0).uniform(-2, 2, (4096, d))
for nx in (1, 4, 5, 16, 64, 128, 256, 512, 896, 1024, 2048, 4096):
    GP.eval_model(xq[:nx])
    t0 = time.perf_counter(); GP.eval_model(xq[:nx]); t1 = time.perf_counter()
    GP.eval_model(xq[:nx], calc_grad=True)
    t2 = time.perf_counter(); GP.eval_model(xq[:nx], calc_grad=True); t3 = time.perf_counter()
    print('nx=%5d: %8.2f ms (%.1f us/point)   with gradients %8.2f ms' % (nx, (t1 - t0) * 1e3, (t1 - t0) * 1e6 / nx, (t3 - t2) * 1e3))
