#!/usr/bin/env python3
"""Golden vector for BASELINE cfg5's own COMBINATION at a size the reference finishes in seconds: Matern-5/2 + known noise on
f and grad f (std_f = 1e-2, std_g = 1e-1, varK a hyperparameter) + d = 16, n = 120 (N = 2040), inputs and restart row 0 exactly as
bench.make_workload(120, 16, "cfg5") builds them (default_rng(0) design, Rosenbrock a = 10, rows
[log10 theta ~ U(-3,-1)^16, log10 varK ~ U(-1,1)] seed 2).  Reference: KernelMatern5f2.py:352-450, CalcLkd.py:185-251.

Runs ONLY in the build container (reference at /root/reference, imported with the two stubs of gen_golden.py); only the .npz travels.
Usage:  python tests/golden/gen_golden_cfg5.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import gen_golden as gg  # noqa: E402


def main():
    import bench
    GaussianProcess = gg._import_reference()
    n, d = 120, 16
    X, f, g, hp_table = bench.make_workload(n, d, "cfg5")
    row = hp_table[0]
    c = gg.make_case(GaussianProcess, name='cfg5_combo_n120_d16', n=n, d=d, kernel='Ma5f2', noise='known_cfg5', seed=0,
                     theta=10.0 ** row[:d], varK=10.0 ** row[d])
    assert np.array_equal(c['x'], X) and np.array_equal(c['f'], f) and np.array_equal(c['g'], g)   # bench's generator IS the fixture's input
    c['hp_row'] = row
    np.savez_compressed(os.path.join(HERE, c['name'] + '.npz'), **c)
    print(f"{c['name']}: ok={c['b_chofac_good']} ln_lkd={c['ln_lkd']:.12e} ln_det={c['ln_det_Kmat']:.12e} beta={c['hp_beta']}")


if __name__ == '__main__':
    main()
