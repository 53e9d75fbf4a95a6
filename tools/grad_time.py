import sys, time, numpy as np
sys.path.insert(0, '.')
import bench, gpgradpy_amd
n, d = 2000, 8
X, f, g, tab = bench.make_workload(n, d)
GP = gpgradpy_amd.GaussianProcess(d, True, 'SqExp', 'precon')
GP.set_data(X, f, np.zeros(n), g, np.zeros((n, d)))
hp = GP.hp_vec2dataclass(GP.hp_info_optz_lkd, tab[0])
GP.calc_lkd_all(hp, calc_grad=True)
t0 = time.perf_counter(); info, ok = GP.calc_lkd_all(hp, calc_grad=True); t1 = time.perf_counter()
print('full-size gradient eval: %.1f ms' % ((t1 - t0) * 1e3), info.ln_lkd, info.ln_lkd_grad)
th = 10.0 ** tab[0]
k = int(np.argmax(np.abs(info.ln_lkd_grad))); h = 1e-4 * th[k]
tp, tm = th.copy(), th.copy(); tp[k] += h; tm[k] -= h
fd = (GP.calc_lkd_all(GP.make_hp_class(theta=tp))[0].ln_lkd - GP.calc_lkd_all(GP.make_hp_class(theta=tm))[0].ln_lkd) / (2 * h)
print('FD check slot', k, fd, info.ln_lkd_grad[k], abs(fd - info.ln_lkd_grad[k]) / abs(fd))
