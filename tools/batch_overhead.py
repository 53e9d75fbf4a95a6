"""Diagnostic (GPU box): where a cfg2 batch call (64 restart rows, bench.py --config cfg2) spends its time -- Python wrapper, C call,
kernels (HIP events around the C call's stream work are not available from here: the C call is synchronous, so wall - kernels = host)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
import gpgradpy_amd

X, f, g, rows = bench.make_workload(500, 4, "cfg2")
GP = gpgradpy_amd.GaussianProcess(4, True, "SqExp", "precon", device=0)
GP.set_data(X, f, np.zeros(500), g, np.zeros((500, 4)))
rows = np.asarray(rows)[:64]
for _ in range(16):
    GP.calc_lkd_batch(rows)
GP._time_chofac = 0.0
t0 = time.perf_counter()
K = 64
for _ in range(K):
    GP.calc_lkd_batch(rows)
wall = (time.perf_counter() - t0) / K
print(f"per batch of {len(rows)}: wall {wall * 1e3:.3f} ms, inside gpg_lkd_batch {GP._time_chofac / K * 1e3:.3f} ms, python wrapper {(wall - GP._time_chofac / K) * 1e3:.3f} ms")
