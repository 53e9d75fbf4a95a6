"""Occupancy over time of a dataflow launch from a tile_probe -DGPG_STAMP timeline: for each of `nbin` time bins, the number of
resident tasks and the MFMA share of their residency (gemm ticks spread uniformly over each task's [start, fin0) interval)."""
import csv, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
nbin = int(sys.argv[2]) if len(sys.argv) > 2 else 20
st = np.array([int(r["start"]) for r in rows], float); en = np.array([int(r["end"]) for r in rows], float)
fin0_all = np.array([int(r["fin0"]) for r in rows], float)
med = np.median(st)
ok = (np.abs(st - med) < 1e8) & (en > st) & (en - st < 1e8) & (fin0_all >= st) & (fin0_all <= en)     # other kernels' stamps share the buffer
t0 = st[ok].min()
st, en, fin0_all = st - t0, en - t0, fin0_all - t0
st, en = st[ok], en[ok]
gm = np.array([int(r["gemm_cyc"]) for r in rows], float)[ok]; sp = np.array([int(r["spin_cyc"]) for r in rows], float)[ok]
fin0 = fin0_all[ok]
T = en.max()
edges = np.linspace(0, T, nbin + 1)
print(f"{len(st)} tasks, span {T / 100:.1f} us; residency sum {np.sum(en - st) / 100 / 1000:.1f} ms; MFMA {np.sum(gm) / np.sum(en - st):.3f} spin {np.sum(sp) / np.sum(en - st):.3f} "
      f"rest-before-fin {np.sum(fin0 - st - gm - sp) / np.sum(en - st):.3f} finalisation+store {np.sum(en - fin0) / np.sum(en - st):.3f}")
for b in range(nbin):
    lo, hi = edges[b], edges[b + 1]
    ov = np.clip(np.minimum(en, hi) - np.maximum(st, lo), 0, None)
    res = ov.sum() / (hi - lo)
    ovg = np.clip(np.minimum(fin0, hi) - np.maximum(st, lo), 0, None)
    dens = gm / np.maximum(fin0 - st, 1)
    print(f"  t={lo / 100:8.1f} us  resident {res:6.1f}  in-MFMA-equivalent {np.sum(ovg * dens) / (hi - lo):6.1f}")
