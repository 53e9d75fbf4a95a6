"""Diagnostic: time of one value + gradient evaluation at the headline size, for a few panel widths of the
forward sweeps of gpg_inverse_from_factor, plus a finite-difference check of the largest component."""
import sys, time, numpy as np
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench, gpgradpy_amd
n, d = 2000, 8
X, f, g, tab = bench.make_workload(n, d)
GP = gpgradpy_amd.GaussianProcess(d, True, 'SqExp', 'precon')
GP.set_data(X, f, np.zeros(n), g, np.zeros((n, d)))
hp = GP.hp_vec2dataclass(GP.hp_info_optz_lkd, tab[0])
for panel in (512, 256, 1024):
    GP.set_panel(panel)
    GP.calc_lkd_all(hp, calc_grad=True)
    t0 = time.perf_counter(); info, ok = GP.calc_lkd_all(hp, calc_grad=True); t1 = time.perf_counter()
    print('panel %4d: value + gradient %.1f ms' % (panel, (t1 - t0) * 1e3), info.ln_lkd)
t0 = time.perf_counter(); GP.calc_lkd_all(hp); t1 = time.perf_counter()
print('value only %.1f ms' % ((t1 - t0) * 1e3))
th = 10.0 ** tab[0]
k = int(np.argmax(np.abs(info.ln_lkd_grad))); h = 1e-4 * th[k]
tp, tm = th.copy(), th.copy(); tp[k] += h; tm[k] -= h
fd = (GP.calc_lkd_all(GP.make_hp_class(theta=tp))[0].ln_lkd - GP.calc_lkd_all(GP.make_hp_class(theta=tm))[0].ln_lkd) / (2 * h)
print('FD check slot', k, fd, info.ln_lkd_grad[k], abs(fd - info.ln_lkd_grad[k]) / abs(fd))
