"""Per-tile-column chain of a single-matrix dataflow launch from a tile_probe -DGPG_STAMP timeline: end time of the diagonal task of
every tile column, the increment from the previous column, and what the diagonal task spent where."""
import csv, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
R = [{k: int(r[k]) for k in r} for r in rows]
med = np.median([v['start'] for v in R])
R = [v for v in R if abs(v['start'] - med) < 1e8 and v['end'] > v['start'] and v['end'] - v['start'] < 1e8]
t0 = min(v['start'] for v in R)
diag = sorted([v for v in R if v['ti'] == v['tj']], key=lambda v: v['tj'])
prev = None
inc = []
for v in diag:
    e = (v['end'] - t0) / 100
    if prev is not None:
        inc.append(e - prev)
    prev = e
inc = np.array(inc)
sub = sorted([v for v in R if v['ti'] == v['tj'] + 1], key=lambda v: v['tj'])
if sub:
    print('first sub-diagonal tile (j+1, j): mean residency %.1f us, K-loop %.1f, spin %.1f, after K-loop (wait for pieces + substitution + publish) %.1f' % (
        np.mean([v['end'] - v['start'] for v in sub]) / 100, np.mean([v['gemm_cyc'] for v in sub]) / 100, np.mean([v['spin_cyc'] for v in sub]) / 100,
        np.mean([v['end'] - v['fin0'] for v in sub]) / 100))
print(f"{len(diag)} diagonal tasks, span {(max(v['end'] for v in R) - t0) / 100:.0f} us; chain increment per tile column: mean {inc.mean():.1f} us, "
      f"median {np.median(inc):.1f}, first 10 {inc[:10].mean():.1f}, last 10 {inc[-10:].mean():.1f}")
step = max(1, len(diag) // 12)
for v in diag[::step]:
    print(f"  col {v['tj']:3d}: start {(v['start'] - t0) / 100:8.1f} spin {v['spin_cyc'] / 100:7.1f} gemm {v['gemm_cyc'] / 100:7.1f} fin0 {(v['fin0'] - t0) / 100:8.1f} end {(v['end'] - t0) / 100:8.1f}  (fin {(v['end'] - v['fin0']) / 100:6.1f})")
