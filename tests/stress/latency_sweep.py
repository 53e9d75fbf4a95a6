"""Diagnostic (GPU box): latency of ONE likelihood evaluation / one posterior set-up on small data sets -- the regime of a
Bayesian-optimisation loop, where the CPU path is quick and a device path is bound by its launches -- next to the CPU oracle
timed on the same inputs (checker used as a stopwatch here, like bench.py's cpu_baseline leg).
    python tests/stress/latency_sweep.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import gpgradpy_amd
from oracle import gp_oracle as orc


def best(f, reps=5):
    f()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); t.append(time.perf_counter() - t0)
    return min(t) * 1e3


print('%5s %3s %6s | %9s %9s %9s %9s %9s | %9s' % ('n', 'd', 'N', 'lkd', 'lkd+grad', 'batch16/16', 'setup', 'eval(1)', 'cpu lkd'))
for n, d, use_grad in [(200, 2, False), (10, 2, True), (20, 4, True), (50, 4, True), (100, 4, True), (40, 10, True), (200, 4, True), (500, 4, True)]:
    X, f, g = orc.synthetic_design(n, d, seed=n)
    GP = gpgradpy_amd.GaussianProcess(d, use_grad, 'SqExp', 'precon')
    GP.set_data(X, f, np.zeros(n), g if use_grad else None, np.zeros((n, d)) if use_grad else None)
    rng = np.random.default_rng(n)
    rows = rng.uniform(-1.5, -0.7, (16, d))
    hp = GP.hp_vec2dataclass(GP.hp_info_optz_lkd, rows[0])
    t_l = best(lambda: GP.calc_lkd_all(hp))
    t_g = best(lambda: GP.calc_lkd_all(hp, calc_grad=True))
    t_b = best(lambda: GP.calc_lkd_batch(rows)) / 16
    hp2 = GP.optz_closed_form_hp(hp)
    t_s = best(lambda: GP.set_hpara('set', 0, hp_vals=hp2))
    xq = rng.uniform(-2, 2, (1, d))
    t_1 = best(lambda: GP.eval_model(xq))
    y = orc.make_data_vec(f, g) if use_grad else f
    N = y.size
    t_c = best(lambda: orc.calc_lkd(X, y, hp.theta, 'SqExp', use_grad, 'precon', GP._etaK, np.zeros(N), False))
    print('%5d %3d %6d | %9.3f %9.3f %9.3f %9.3f %9.3f | %9.3f' % (n, d, N, t_l, t_g, t_b, t_s, t_1, t_c), flush=True)
    GP.close()
