"""Kernel-level view of a single-point posterior evaluation: run under rocprofv3 --kernel-trace --stats."""
import sys, numpy as np
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench, gpgradpy_amd
n, d = 2000, 8
X, f, g, tab = bench.make_workload(n, d)
GP = gpgradpy_amd.GaussianProcess(d, True, 'SqExp', 'precon')
GP.set_data(X, f, np.zeros(n), g, np.zeros((n, d)))
hp = GP.optz_closed_form_hp(GP.hp_vec2dataclass(GP.hp_info_optz_lkd, tab[0]))
GP.set_hpara('set', 0, hp_vals=hp)
xq = np.random.default_rng(0).uniform(-2, 2, (64, d))
for _ in range(20):
    GP.eval_model(xq[:1])
