"""Diagnostic: compare two per-task timelines of tools/tile_probe (-DGPG_STAMP; 100 MHz ticks): where does the dependency
chain spend its time?     python tools/timeline_compare.py a.csv b.csv"""
import csv, sys
import numpy as np

def load(path):
    rows = list(csv.DictReader(open(path)))
    d = {k: np.array([float(r[k]) for r in rows]) for k in rows[0]}
    # the CSV holds raw 100 MHz stamps when junk records (panel-solve cycle counters that share the buffer) spoil the
    # writer's time origin: keep the records that are task records and re-base them
    ok = (d['start'] > 1e9) & (d['end'] > d['start']) & (d['fin0'] >= d['start']) & (d['fin0'] <= d['end'])
    d = {k: v[ok] for k, v in d.items()}
    t0 = d['start'].min()
    for k in ('start', 'end', 'fin0'):
        d[k] = d[k] - t0
    return d

for path in sys.argv[1:]:
    d = load(path)
    us = 0.01
    start, end = d['start'] * us, d['end'] * us
    ti, tj = d['ti'].astype(int), d['tj'].astype(int)
    print(f'== {path}: {len(start)} tasks, makespan {end.max():.0f} us')
    diag = ti == tj
    order = np.argsort(tj[diag])
    dj, ds, de, dfin = tj[diag][order], start[diag][order], end[diag][order], d['fin0'][diag][order] * us
    hop = np.diff(de)
    print(f'  diagonal chain: mean hop {hop.mean():.1f} us, median {np.median(hop):.1f}; first 20 cols {hop[:20].mean():.1f}, last 20 cols {hop[-20:].mean():.1f}')
    # per hop decomposition: from diag(j) end to diag(j+1) start-of-finalisation (fin0) and its finalisation length
    fin_len = de - dfin
    print(f'  diagonal finalisation length: mean {fin_len.mean():.1f} us; wait before it (fin0 - start): mean {(dfin - ds).mean():.1f} us')
    # when did diag(j+1) start relative to diag(j) end?  negative = it was resident and waiting
    lead = ds[1:] - de[:-1]
    print(f'  diag(j+1) start minus diag(j) end: mean {lead.mean():.1f} us, frac started after {np.mean(lead > 0):.2f}, mean when late {lead[lead > 0].mean() if np.any(lead > 0) else 0:.1f}')
    # the sub-diagonal tile (j+1, j): start / end relative to diag(j) end
    sub = (ti == tj + 1)
    so = np.argsort(tj[sub])
    sj, ss, se, sfin = tj[sub][so], start[sub][so], end[sub][so], d['fin0'][sub][so] * us
    m = min(len(sj), len(de))
    print(f'  tile (j+1,j): end minus diag(j) end: mean {(se[:m] - de[:m]).mean():.1f} us; started after diag(j) end in {np.mean(ss[:m] > de[:m]):.2f} of columns (mean lateness {np.maximum(ss[:m] - de[:m], 0).mean():.1f} us)')
    print(f'  diag(j+1) end minus tile (j+1,j) end: mean {(de[1:m + 1] - se[:m][:len(de) - 1]).mean():.1f} us')
    dur = end - start
    spin = d['spin_cyc'] * us
    print(f'  task residency mean {dur.mean():.0f} us, spin share {spin.sum() / dur.sum():.3f}, finalisation share {(end - d["fin0"] * us).sum() / dur.sum():.3f}')
    if 'wg' in d and len(set(d['wg'])) < len(start):
        gaps = []
        wg = d['wg'].astype(int)
        for w in set(wg):
            idx = np.where(wg == w)[0]
            idx = idx[np.argsort(start[idx])]
            gaps += list(start[idx][1:] - end[idx][:-1])
        gaps = np.array(gaps)
        print(f'  gap between consecutive tasks of a workgroup: mean {gaps.mean():.2f} us, p90 {np.percentile(gaps, 90):.2f}')
