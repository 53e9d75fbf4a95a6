"""Diagnostic: two batched value + gradient calls (8 restart rows) at the headline size, for rocprofv3 kernel statistics."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench, gpgradpy_amd
n, d = 2000, 8
X, f, g, tab = bench.make_workload(n, d)
GP = gpgradpy_amd.GaussianProcess(d, True, 'SqExp', 'precon')
GP.set_data(X, f, np.zeros(n), g, np.zeros((n, d)))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
GP.calc_lkd_grad_batch(tab[:B])
t0 = time.perf_counter(); GP.calc_lkd_grad_batch(tab[:B]); t1 = time.perf_counter()
print('batched value + gradient: %.1f ms per row (%d rows)' % ((t1 - t0) * 1e3 / B, B))
