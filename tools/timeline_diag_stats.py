"""Diagonal / off-diagonal task statistics of a tile_probe -DGPG_STAMP timeline (100 MHz ticks -> us)."""
import csv, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
R = [{k: int(r[k]) for k in r} for r in rows]
med = np.median([v['start'] for v in R])
R = [v for v in R if abs(v['start'] - med) < 1e8 and v['end'] > v['start'] and v['end'] - v['start'] < 1e8]
t0 = min(v['start'] for v in R)
diag = [v for v in R if v['ti'] == v['tj']]; off = [v for v in R if v['ti'] != v['tj']]
us = lambda x: np.mean(x) / 100
print(f"span {(max(v['end'] for v in R) - t0) / 100:.0f} us, {len(R)} tasks")
print('diag: residency %.1f gemm %.1f spin %.1f | potrf1 %.1f L21 %.1f syrk %.1f potrf2 %.1f' % (
    us([v['end'] - v['start'] for v in diag]), us([v['gemm_cyc'] for v in diag]), us([v['spin_cyc'] for v in diag]),
    us([v['f1'] - v['f0'] for v in diag]), us([v['f2'] - v['f1'] for v in diag]), us([v['f3'] - v['f2'] for v in diag]),
    us([v['f4'] - v['f3'] for v in diag if v['f4'] > v['f3']])))
print('off : residency %.1f gemm %.1f spin %.1f | store %.1f wait+solve %.1f' % (
    us([v['end'] - v['start'] for v in off]), us([v['gemm_cyc'] for v in off]), us([v['spin_cyc'] for v in off]),
    us([v['f0'] - v['fin0'] for v in off]), us([v['f2'] - v['f0'] for v in off if v['f2'] > v['f0']])))
o5 = [v for v in off if v.get('f5', 0) > 0 and v['f7'] > 0]
if o5:
    print('off finalisation split (us): wait L11 + block-0 substitution %.1f | store X1 + wait L21 %.1f | MFMA update of block 1 (2 passes) %.1f | block-1 substitution %.1f | tail %.1f' % (
        us([v['f5'] - v['f1'] for v in o5]), us([v['f6'] - v['f5'] for v in o5]), us([v['f7'] - v['f6'] for v in o5]), us([v['f2'] - v['f7'] for v in o5]), us([v['end'] - v['f2'] for v in o5])))
for j in sorted(set(v['tj'] for v in R))[::max(1, len(set(v['tj'] for v in R)) // 8)]:
    o = [v for v in off if v['tj'] == j]; d = [v for v in diag if v['tj'] == j]
    if o and d:
        print(f"  col {j:3d}: off gemm {us([v['gemm_cyc'] for v in o]):7.1f} spin {us([v['spin_cyc'] for v in o]):6.1f} fin {us([v['end'] - v['fin0'] for v in o]):6.1f} | diag start {np.mean([v['start'] for v in d]) / 100 - t0 / 100:8.1f} "
              f"spin {us([v['spin_cyc'] for v in d]):6.1f} end {np.mean([v['end'] for v in d]) / 100 - t0 / 100:8.1f} | off start {np.mean([v['start'] for v in o]) / 100 - t0 / 100:8.1f} fin0 {np.mean([v['fin0'] for v in o]) / 100 - t0 / 100:8.1f} end {np.mean([v['end'] for v in o]) / 100 - t0 / 100:8.1f}")
