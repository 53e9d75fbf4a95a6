"""Diagnostic: one-at-a-time timings of the path's entry points over a range of sizes (looks for performance cliffs
at the schedule thresholds)."""
import sys, time, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench, gpgradpy_amd


def best(f, reps=3):
    f()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); t.append(time.perf_counter() - t0)
    return min(t) * 1e3


print('%6s %3s %7s | %9s %9s %9s %9s %9s %9s' % ('n', 'd', 'N', 'lkd', 'lkd+grad', 'batch8/8', 'setup', 'eval(1)', 'eval(64)'))
for n, d in [(50, 2), (100, 4), (250, 4), (500, 4), (700, 8), (1000, 8), (1023, 8), (1030, 8), (1500, 8), (2000, 8), (1200, 16)]:
    X, f, g, _ = bench.make_workload(n, d)
    GP = gpgradpy_amd.GaussianProcess(d, True, 'SqExp', 'precon')
    GP.set_data(X, f, np.zeros(n), g, np.zeros((n, d)))
    rng = np.random.default_rng(n)
    hp_rows = rng.uniform(-2.0, -0.7, (8, d))
    hp = GP.hp_vec2dataclass(GP.hp_info_optz_lkd, hp_rows[0])
    t_l = best(lambda: GP.calc_lkd_all(hp))
    t_g = best(lambda: GP.calc_lkd_all(hp, calc_grad=True))
    t_b = best(lambda: GP.calc_lkd_batch(hp_rows)) / 8
    hp2 = GP.optz_closed_form_hp(hp)
    t_s = best(lambda: GP.set_hpara('set', 0, hp_vals=hp2))
    xq = rng.uniform(-2, 2, (64, d))
    t_1 = best(lambda: GP.eval_model(xq[:1]))
    t_64 = best(lambda: GP.eval_model(xq))
    print('%6d %3d %7d | %9.3f %9.3f %9.3f %9.3f %9.3f %9.3f' % (n, d, n * (d + 1), t_l, t_g, t_b, t_s, t_1, t_64), flush=True)
    del GP
