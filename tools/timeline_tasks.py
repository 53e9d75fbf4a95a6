"""Task-level accounting of a tools/tile_probe -DGPG_STAMP timeline: residency, MFMA loop, flag waits, rest (per task kind) and the share
of the launch's slot-time spent inside MFMA loops.  Stamps are s_memrealtime ticks (100 MHz), differences taken modulo 2^64."""
import csv
import sys
M = 1 << 64
rows = list(csv.DictReader(open(sys.argv[1])))
acc = {}
for r in rows:
    st, en = int(r["start"]), int(r["end"])
    res = (en - st) % M
    if res == 0 or res > 10 ** 8:
        continue
    kind = "diagonal" if int(r["ti"]) == int(r["tj"]) else "off-diagonal"
    a = acc.setdefault(kind, [0, 0, 0, 0])
    a[0] += 1; a[1] += res; a[2] += int(r["gemm_cyc"]); a[3] += int(r["spin_cyc"])
tot = sum(a[1] for a in acc.values()); loop = sum(a[2] for a in acc.values())
for kind, (n, res, gem, spin) in acc.items():
    print(f"{kind}: {n} tasks, residency {res / n / 100:.1f} us = MFMA loop {gem / n / 100:.1f} + flag waits {spin / n / 100:.1f} + rest {(res - gem - spin) / n / 100:.1f}")
print(f"slot-time {tot / 1e8:.3f} s, inside MFMA loops {100.0 * loop / tot:.1f} %")
