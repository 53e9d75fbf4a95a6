"""Build-container only: wall time of the REFERENCE's set_hpara('optz') and of its likelihood evaluations at BASELINE cfg2
size (n = 500, d = 4), on this container's CPU cores, next to which tools/optz_time.py's device numbers are quoted
(VERDICT r01 item 7).  Imports /root/reference with the stubs of tests/golden/gen_golden.py; smt's LHS is replaced by
SciPy's Latin hypercube (seed 1), as in gpgradpy_amd/hpara_optz.py.  Writes JSON to stdout; nothing of the reference
is copied.    python tools/ref_optz_time.py [n] [d]"""
import json, os, sys, time, types
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', 'tests', 'golden'))
sys.path.insert(0, os.path.join(HERE, '..'))
import gen_golden as gg
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
d = int(sys.argv[2]) if len(sys.argv) > 2 else 4
GaussianProcess = gg._import_reference()
from scipy.stats import qmc


class LHS:                                   # stand-in for smt.sampling_methods.LHS(xlimits=..., random_state=1)
    def __init__(self, xlimits=None, random_state=1, **k):
        self.xl, self.seed = np.asarray(xlimits, dtype=float), random_state

    def __call__(self, nt):
        u = qmc.LatinHypercube(d=self.xl.shape[0], seed=self.seed).random(nt)
        return self.xl[:, 0] + u * (self.xl[:, 1] - self.xl[:, 0])


for _name, _mod in list(sys.modules.items()):             # rebind the name wherever the reference imported it
    if _name.endswith('GpHparaX0') and hasattr(_mod, 'LHS'):
        _mod.LHS = LHS
X, f, g, tab = bench.make_workload(n, d)
out = {'n': n, 'd': d, 'N': n * (d + 1), 'nproc': os.cpu_count()}
GP = GaussianProcess(d, True, 'SqExp', 'precon')
GP.set_data(X, f, np.zeros(n), g, np.zeros((n, d)))
hp = GP.make_hp_class(theta=10.0 ** tab[0])
GP.calc_lkd_all(hp)
t0 = time.perf_counter(); GP.calc_lkd_all(hp); out['ref_value_s'] = time.perf_counter() - t0
t0 = time.perf_counter(); GP.calc_lkd_all(hp, calc_grad=True); out['ref_value_grad_s'] = time.perf_counter() - t0
for mode in ('hp_best', 'lhs'):
    GPo = GaussianProcess(d, True, 'SqExp', 'precon')
    GPo.lkd_optz_start_mtd = mode
    GPo.init_optz_surr(2)
    GPo.set_data(X, f, np.zeros(n), g, np.zeros((n, d)))
    # the reference needs one stored history row to centre its start points on (GpHparaX0.py:67-130)
    GPo.hp_theta_all[0] = GPo.hp_theta_init
    GPo.hp_varK_all[0] = GPo.hp_varK_init
    t0 = time.perf_counter()
    GPo.set_hpara('optz', 1)
    out[f'ref_set_hpara_optz_{mode}_s'] = time.perf_counter() - t0
    out[f'ref_iter_mean_{mode}'] = float(GPo.hp_optz_iter_mean[1])
    out[f'ref_theta_{mode}'] = [float(v) for v in GPo.hp_vals.theta]
print(json.dumps(out))
