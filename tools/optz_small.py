"""Diagnostic: wall time of one Bayesian-optimisation-style model update on small data sets -- set_data + set_hpara('optz') (default
multi-start, SLSQP starts in lock step) + one posterior evaluation with gradients -- with a cProfile of the host side."""
import cProfile, io, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench, gpgradpy_amd
d = 4
X, f, g, _ = bench.make_workload(400, d)
GP = gpgradpy_amd.GaussianProcess(d, True, 'SqExp', 'precon')
GP.init_optz_surr(8)
i_optz = 0
for n in (20, 21, 100, 101, 300):
    t0 = time.perf_counter()
    GP.set_data(X[:n], f[:n], np.zeros(n), g[:n], np.zeros((n, d)))
    t1 = time.perf_counter()
    GP.set_hpara('optz', i_optz); i_optz += 1
    t2 = time.perf_counter()
    GP.eval_model(X[n:n + 1] * 0.9, calc_grad=True)
    t3 = time.perf_counter()
    print(f'n = {n:4d} (N = {n * (d + 1):5d}): set_data {1e3 * (t1 - t0):7.2f} ms, set_hpara(optz) {1e3 * (t2 - t1):8.2f} ms, eval_model {1e3 * (t3 - t2):6.2f} ms', flush=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
GP.set_data(X[:n], f[:n], np.zeros(n), g[:n], np.zeros((n, d)))
pr = cProfile.Profile(); pr.enable()
GP.set_hpara('optz', i_optz)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(22); print(s.getvalue()[:3500])
