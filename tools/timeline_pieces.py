"""Piece-level split of the off-diagonal finalisation (column block 0) from a tools/tile_probe -DGPG_STAMP timeline:
per 16-column piece the time to the flag (wait), to the image in LDS (load + barrier) and through the 16 column steps."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
acc = {}
n = 0
for r in rows:
    ti, tj = int(r["ti"]), int(r["tj"])
    if ti == tj or int(r["p0w"]) == 0:
        continue
    f1 = int(r["f1"])                      # start of tile_solve_rows128 (after the tile store)
    prev = f1
    ok = True
    vals = []
    for s in range(4):
        w, i, su = int(r[f"p{s}w"]), int(r[f"p{s}i"]), int(r[f"p{s}s"])
        if not (prev <= w <= i <= su):
            ok = False
            break
        vals += [w - prev, i - w, su - i]
        prev = su
    if not ok:
        continue
    for k, v in enumerate(vals):
        acc[k] = acc.get(k, 0) + v
    acc["store"] = acc.get("store", 0) + (f1 - int(r["fin0"]))
    acc["res"] = acc.get("res", 0) + (int(r["end"]) - int(r["start"]))
    acc["gemm"] = acc.get("gemm", 0) + int(r["gemm_cyc"])
    acc["spin"] = acc.get("spin", 0) + int(r["spin_cyc"])
    acc["fin"] = acc.get("fin", 0) + (int(r["end"]) - int(r["fin0"]))
    f5, f6, f7 = int(r["f5"]), int(r["f6"]), int(r["f7"])
    acc["x1"] = acc.get("x1", 0) + (f6 - f5)
    acc["upd"] = acc.get("upd", 0) + (f7 - f6)
    acc["blk1"] = acc.get("blk1", 0) + (int(r["f2"]) - f7 if int(r["f2"]) > f7 else 0)
    acc["tail"] = acc.get("tail", 0) + (int(r["end"]) - int(r["f2"]) if int(r["f2"]) else 0)
    n += 1
if not n:
    sys.exit("no records")
us = lambda v: v / n / 100.0
print(f"{n} off-diagonal tasks: residency {us(acc['res']):.1f} us = MFMA loop {us(acc['gemm']):.1f} + flag waits {us(acc['spin']):.1f} + finalisation (from the end of the loop) {us(acc['fin']):.1f} + rest")
print(f"  tile store + barrier before the solve: {us(acc['store']):.1f} us")
for s in range(4):
    print(f"  piece {s}: wait {us(acc[3*s]):5.1f}  image load + barrier {us(acc[3*s+1]):5.1f}  16 column steps {us(acc[3*s+2]):5.1f} us")
print(f"  store X1 + wait L21 {us(acc['x1']):.1f} | block-1 update (2 MFMA passes) {us(acc['upd']):.1f} | block-1 pieces + store {us(acc['blk1']):.1f} | publish {us(acc['tail']):.1f} us")
