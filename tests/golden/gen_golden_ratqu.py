#!/usr/bin/env python3
"""Golden vectors for the rational quadratic kernel ('RatQu') from the *reference*
(KernelRatQuad.py:439-554 through calc_lkd_all / set_hpara / eval_model), with the loader and the case
builder of gen_golden.py.  Runs only in the build container (reference mounted at /root/reference).

Usage:  python tests/golden/gen_golden_ratqu.py        # writes tests/golden/RatQu_*.npz
"""
import os

import numpy as np

import gen_golden as gg


def main():
    GaussianProcess = gg._import_reference()
    cases = []
    add = lambda **k: cases.append(gg.make_case(GaussianProcess, kernel='RatQu', **k))
    add(name='RatQu_none_n2_d1', n=2, d=1, noise='none', seed=301, store_mats=True)                 # alpha = default 2
    add(name='RatQu_none_n5_d2', n=5, d=2, noise='none', seed=302, store_mats=True, hp_kernel=1.3)
    add(name='RatQu_known_n17_d4', n=17, d=4, noise='known', seed=303, store_mats=True, hp_kernel=0.7)
    add(name='RatQu_unknown_n12_d3', n=12, d=3, noise='unknown', seed=304, hp_kernel=3.0)
    add(name='RatQu_none_n64_d8', n=64, d=8, noise='none', seed=305)
    add(name='RatQu_none_n33_d3_spread', n=33, d=3, noise='none', seed=306, theta=np.array([1e-3, 3e-1, 4.0]), hp_kernel=0.2)
    add(name='RatQu_none_n20_d2_neardup', n=20, d=2, noise='none', seed=307, near_dup=True, hp_kernel=5.0)
    # no gradient-mask case: the reference's cross kernel fails for RatQu with bvec_use_grad1 only (KernelRatQuad.py:497-499
    # takes the masked B^(-alpha-1) for the unmasked side: shapes (n1, nx) vs (n1g, nx))
    add(name='RatQu_none_n40_d2_nograd', n=40, d=2, noise='none', use_grad=False, seed=309, theta=np.array([0.5, 0.8]),
        store_mats=True, hp_kernel=2.5)
    add(name='RatQu_none_n12_d2_base', n=12, d=2, noise='none', wellcond='base', seed=310, theta=np.array([0.4, 0.9]),
        store_mats=True)
    for c in cases:
        np.savez_compressed(os.path.join(gg.HERE, c['name'] + '.npz'), **c)
        print(f"{c['name']:36s} ok={c['b_chofac_good']} alpha={c['hp_kernel']} ln_lkd={c.get('ln_lkd', float('nan')):.12e}")


if __name__ == '__main__':
    main()
