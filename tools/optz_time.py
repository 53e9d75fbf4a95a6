"""Timing of what the optimiser calls: value + gradient per evaluation, one at a time and batched, and the
wall time of set_hpara('optz') with the SLSQP starts in lock step against the sequential loop; JSON on stdout.
    python tools/optz_time.py [cfg2|cfg3]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench, gpgradpy_amd

cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
n, d = {'cfg2': (500, 4), 'cfg3': (2000, 8)}[cfg]
X, f, g, tab = bench.make_workload(n, d)
out = {'config': cfg, 'n': n, 'd': d, 'N': n * (d + 1)}


def new_gp():
    GP = gpgradpy_amd.GaussianProcess(d, True, 'SqExp', 'precon')
    GP.set_data(X, f, np.zeros(n), g, np.zeros((n, d)))
    return GP


GP = new_gp()
rows = tab[:8]
hp = GP.hp_vec2dataclass(GP.hp_info_optz_lkd, rows[0])
GP.calc_lkd_all(hp, calc_grad=True)
GP.calc_lkd_grad_batch(rows)
reps = 20 if cfg == 'cfg2' else 3
t0 = time.perf_counter()
for _ in range(reps):
    GP.calc_lkd_all(hp)
out['value_ms_single'] = (time.perf_counter() - t0) / reps * 1e3
t0 = time.perf_counter()
for _ in range(reps):
    GP.calc_lkd_all(hp, calc_grad=True)
out['value_grad_ms_single'] = (time.perf_counter() - t0) / reps * 1e3
for B in (2, 5, 8):
    GP.calc_lkd_grad_batch(rows[:B])
    t0 = time.perf_counter()
    for _ in range(reps):
        GP.calc_lkd_grad_batch(rows[:B])
    out[f'value_grad_ms_per_row_batch{B}'] = (time.perf_counter() - t0) / reps / B * 1e3
ln, grad, ok = GP.calc_lkd_grad_batch(rows[:2])
info = GP.calc_lkd_all(GP.hp_vec2dataclass(GP.hp_info_optz_lkd, rows[1]), calc_grad=True)[0]
out['batch_vs_single_max_rel_diff'] = float(np.max(np.abs(grad[1] - info.ln_lkd_grad) / np.abs(info.ln_lkd_grad).max()))

if cfg == 'cfg2' or '--optz' in sys.argv:
    for mode, lock in (('lockstep', True), ('sequential', False)):
        GPo = new_gp()
        GPo.lkd_optz_start_mtd = 'lhs'
        GPo.optz_n_x0 = 5
        GPo.optz_lockstep = lock
        GPo.init_optz_surr(2)
        GPo.set_data(X, f, np.zeros(n), g, np.zeros((n, d)))
        t0 = time.perf_counter()
        GPo.set_hpara('optz', 0)
        out[f'set_hpara_optz_lhs5_s_{mode}'] = time.perf_counter() - t0
        out[f'device_call_s_{mode}'] = float(GPo.time_chofac_all[0])
        out[f'optz_iter_mean_{mode}'] = float(GPo.hp_optz_iter_mean[0])
        out[f'ln_lkd_best_{mode}'] = float(-np.nanmin(GPo.optz_obj_all_last))
        if lock:
            out['lockstep_batches'], out['lockstep_rows'] = GPo.optz_lockstep_batches, GPo.optz_lockstep_rows
    GPo = new_gp()                                   # the reference's default start method: 40 value-only rows, one SLSQP run
    GPo.init_optz_surr(2)
    GPo.set_data(X, f, np.zeros(n), g, np.zeros((n, d)))
    t0 = time.perf_counter()
    GPo.set_hpara('optz', 0)
    out['set_hpara_optz_hp_best_s'] = time.perf_counter() - t0
    out['optz_iter_hp_best'] = float(GPo.hp_optz_iter_mean[0])
print(json.dumps(out))
