import sys; sys.path.insert(0, '/root/repo')
import numpy as np, bench, gpgradpy_amd
n, d = 4000, 16
X, f, g, tab = bench.make_workload(n, d, "cfg5")
GP = gpgradpy_amd.GaussianProcess(d, True, 'Ma5f2', 'precon')
GP.set_data(X, f, np.full(n, 1e-2), g, np.full((n, d), 1e-1))
hp = GP.hp_vec2dataclass(GP.hp_info_optz_lkd, tab[0])
info_g, ok = GP.calc_lkd_all(hp, calc_grad=True)
gr = info_g.ln_lkd_grad
for k in np.argsort(-np.abs(gr[:d]))[:3]:
    for rel in (1e-3, 1e-4, 1e-5):
        th = hp.theta.copy(); h = rel * th[k]
        tp, tm = th.copy(), th.copy(); tp[k] += h; tm[k] -= h
        mk = lambda t: GP.make_hp_class(theta=t, varK=hp.varK)
        fd = (GP.calc_lkd_all(mk(tp))[0].ln_lkd - GP.calc_lkd_all(mk(tm))[0].ln_lkd) / (2 * h)
        print(k, rel, fd, gr[k], abs(fd - gr[k]) / abs(fd), flush=True)
