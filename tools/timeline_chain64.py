"""Chain of the fused diagonal tasks of ONE matrix in the 64-tile kernel (tools/tile_probe_stamp <n> 2): from the previous diagonal tile's
last piece going up to this one's, split at the stamps of the fused path (p0w..p2w columns of the csv = third stamp region)."""
import csv, sys
import numpy as np
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["ti"] == r["tj"] and int(r["tj"]) > 3 and int(r["p2w"]) > 0]
rows.sort(key=lambda r: int(r["tj"]))
g = lambda r, k: int(r[k])
acc = {k: [] for k in ("flag3 up(j-1) -> seen", "seen -> piece 3 substituted", "-> update + flag of (j, j-1)", "-> potrf starts", "potrf", "chain")}
for a, b in zip(rows, rows[1:]):
    if int(b["tj"]) != int(a["tj"]) + 1:
        continue
    up_prev, up = g(a, "p2w"), g(b, "p2w")      # pz[6]: potrf returned = last piece published
    seen, sub3, upd, pstart = g(b, "p0s"), g(b, "p1w"), g(b, "p1i"), g(b, "p1s")   # pz[2], pz[3], pz[4], pz[5]
    acc["flag3 up(j-1) -> seen"].append(seen - up_prev)
    acc["seen -> piece 3 substituted"].append(sub3 - seen)
    acc["-> update + flag of (j, j-1)"].append(upd - sub3)
    acc["-> potrf starts"].append(pstart - upd)
    acc["potrf"].append(up - pstart)
    acc["chain"].append(up - up_prev)
for k, v in acc.items():
    print(f"{k:34s} {np.mean(v) / 100:6.2f} us  (median {np.median(v) / 100:.2f}, n {len(v)})")
