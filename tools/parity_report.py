"""Measured device-vs-reference errors over every golden vector (tests/golden/*.npz), next to the tolerances of
tests/tolerances.py (SURVEY.md 8d asks for the measured errors beside the budget).  Writes a table to stdout."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from conftest import golden_case_paths, load_case
import tolerances as tol
from test_gpu_parity import _gp_from_case, _hp_from_case

worst = dict(ln_det_abs_over_N=0.0, ln_lkd_rel=0.0, beta_rel=0.0, varK_rel=0.0, alpha_normwise=0.0, mu_rel=0.0, sig_abs_over_sigK=0.0,
             dmudx_rel=0.0, dsigdx_rel=0.0)
rows = []
for path in golden_case_paths():
    c = load_case(path)
    GP = _gp_from_case(c)
    hp = _hp_from_case(GP, c)
    info, ok = GP.calc_lkd_all(hp)
    if not ok:
        continue
    e = {}
    e['ln_det_abs_over_N'] = abs(info.ln_det_Kmat - c['ln_det_Kmat']) / GP.n_data
    e['ln_lkd_rel'] = abs(info.ln_lkd - c['ln_lkd']) / max(1e-300, abs(c['ln_lkd']))
    e['beta_rel'] = abs(info.hp_beta[0] - c['hp_beta'][0]) / max(1e-300, abs(c['hp_beta'][0]))
    if not c['b_has_noisy_data']:
        e['varK_rel'] = abs(info.hp_varK - c['hp_varK']) / abs(c['hp_varK'])
    hp2 = GP.optz_closed_form_hp(hp)
    GP.set_hpara('set', 0, hp_vals=hp2)
    e['alpha_normwise'] = np.linalg.norm(GP.invKernEta_fdiff - c['alpha']) / np.linalg.norm(c['alpha'])
    mu, sig, dmu, dsig = GP.eval_model(c['xq'], calc_grad=True)[:4]
    e['mu_rel'] = np.max(np.abs(mu - c['mu']) / np.maximum(1.0, np.abs(c['mu'])))
    e['sig_abs_over_sigK'] = np.max(np.abs(sig - c['sig'])) / np.sqrt(hp2.varK)
    if 'dmudx' in c:
        e['dmudx_rel'] = np.max(np.abs(dmu - c['dmudx'])) / max(1e-300, np.abs(c['dmudx']).max())
        e['dsigdx_rel'] = np.max(np.abs(dsig - c['dsigdx'])) / max(1e-300, np.abs(c['dsigdx']).max())
    for k, v in e.items():
        worst[k] = max(worst[k], float(v))
    rows.append((c['name'], e))
budget = dict(ln_det_abs_over_N='1e-9 (+2e-6/N)', ln_lkd_rel='1e-8', beta_rel='1e-8', varK_rel='1e-8', alpha_normwise='1e-5',
              mu_rel='1e-7', sig_abs_over_sigK='1e-7', dmudx_rel='see tests/tolerances.py', dsigdx_rel='see tests/tolerances.py')
print(f"# device (HIP path, default schedule) vs the reference's golden vectors: worst case over {len(rows)} cases")
print(f"{'quantity':24s} {'worst measured':>16s}   budget")
for k, v in worst.items():
    print(f"{k:24s} {v:16.3e}   {budget[k]}")
print("\n# per case: ln_lkd_rel, alpha_normwise, mu_rel, sig_abs_over_sigK")
for name, e in rows:
    print(f"{str(name):34s} {e['ln_lkd_rel']:10.2e} {e['alpha_normwise']:10.2e} {e['mu_rel']:10.2e} {e['sig_abs_over_sigK']:10.2e}")
