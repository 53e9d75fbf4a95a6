"""Phase split of fin128_offdiag (tile128_chol_kernel) from a tools/tile_probe -DGPG_STAMP timeline: f0..f6 = entry, verdict on the
diagonal tile (first barrier), image of L11 in LDS, 64 column steps, X1 stored + barrier, block-1 update, 64 column steps; f7 = 2 fast / 1 piecewise."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
acc = {"fast": [0] * 8, "slow": [0] * 8}
cnt = {"fast": 0, "slow": 0}
for r in rows:
    if int(r["ti"]) == int(r["tj"]):
        continue
    kind = {2: "fast", 1: "slow"}.get(int(r["f7"]) + 0 if False else None, None)
# (the csv writer subtracts t0 from every stamp, also from the f7 marker: recover it from the raw difference to f0)
for r in rows:
    if int(r["ti"]) == int(r["tj"]):
        continue
    f = [int(r[k]) for k in ("f0", "f1", "f2", "f3", "f4", "f5", "f6")]
    fin0, end = int(r["fin0"]), int(r["end"])
    if f[0] == 0 or f[1] < f[0]:
        continue
    kind = "fast" if f[2] >= f[1] and f[6] >= f[5] > 0 else "slow"
    a = acc[kind]
    cnt[kind] += 1
    a[0] += f[0] - fin0
    a[1] += f[1] - f[0]
    if kind == "fast":
        for k in range(2, 7):
            a[k] += f[k] - f[k - 1]
        a[7] += end - f[6]
    else:
        a[7] += end - f[1]
for kind in ("fast", "slow"):
    n = cnt[kind]
    if not n:
        continue
    us = [v / n / 100.0 for v in acc[kind]]
    if kind == "fast":
        print(f"{n} tasks on the one-piece path: hand-over {us[0]:.1f} us | verdict + barrier {us[1]:.1f} | L11 image {us[2]:.1f} | 64 steps {us[3]:.1f} | "
              f"X1 store + L22 fetch + barrier {us[4]:.1f} | block-1 update {us[5]:.1f} | L22 image + 64 steps {us[6]:.1f} | store + publish {us[7]:.1f}")
    else:
        print(f"{n} tasks on the piecewise path: hand-over {us[0]:.1f} us | verdict + barrier {us[1]:.1f} | rest {us[7]:.1f}")
