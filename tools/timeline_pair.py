"""Phase accounting of a tools/tile_probe_stamp timeline of pair128_chol_kernel (team 0's record per pair task): per task kind
(diagonal pair, off-diagonal pair on the one-piece path, off-diagonal pair on the piecewise path) the mean residency split into
tile load + flag scan, MFMA loop, hand-over, and the phases of the finalisation; plus the share of the launch's slot time each takes.
Stamps are s_memrealtime ticks (100 MHz)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
kinds = {}
tot_slot = 0
for r in rows:
    st, en = int(r["start"]), int(r["end"])
    res = en - st
    if res <= 0 or res > 10 ** 8:
        continue
    tot_slot += res
    ti, tj = int(r["ti"]), int(r["tj"])
    fin0 = int(r["fin0"])
    f = [int(r[k]) for k in ("f0", "f1", "f2", "f3", "f4", "f5", "f6")]
    path = int(r["f7"])            # 0: diagonal, 1: piecewise, 2: one piece   (absolute-minus-t0 formatting: small numbers stay small)
    wait = int(r["finwait"])
    gem, spin = int(r["gemm_cyc"]), int(r["spin_cyc"])
    if ti == tj:
        kind = "diagonal"
        ph = {"load+scan": fin0 - st - gem, "mfma": gem, "handover": f[0] - fin0, "potrf A": f[1] - f[0], "L21 solve": f[2] - f[1],
              "A22 update": f[3] - f[2], "potrf B": f[4] - f[3], "publish": en - f[4]}
    else:
        kind = "off-diag one-piece" if path == 2 else "off-diag piecewise"
        if kind.endswith("one-piece"):
            ph = {"load+scan": fin0 - st - gem, "mfma": gem, "handover": f[0] - fin0, "verdict": f[1] - f[0], "L11 image": f[2] - f[1],
                  "64 steps": f[3] - f[2], "store+L22 fetch": f[4] - f[3], "update": f[5] - f[4], "image + 64 steps": f[6] - f[5],
                  "store+publish": en - f[6]}
            pz = [int(r[k]) for k in ("p0w", "p0i", "p0s", "p1w", "p1i", "p1s", "p2w")]
            if pz[0] and pz[4]:
                ph.update({"  [tail: stores issued": pz[0] - f[6], "barrier": pz[1] - pz[0], "return": pz[2] - pz[1], "ticket + vmcnt(0)": pz[3] - pz[2],
                           "barrier ": pz[4] - pz[3], "flags]": en - pz[4], "  [handover: before the call": pz[5] - fin0, "call]": f[0] - pz[5],
                           "  [X1 store + L22 loads issued": pz[6] - f[3], "vmcnt(0) + barrier]": f[4] - pz[6]})
        else:
            ph = {"load+scan": fin0 - st - gem, "mfma": gem, "handover": f[0] - fin0, "verdict": f[1] - f[0], "block 0 pieces": f[2] - f[1],
                  "store + L21 wait": f[3] - f[2], "update": f[4] - f[3], "block 1 pieces": f[5] - f[4], "store": f[6] - f[5],
                  "publish": en - f[6], "(of which flag polls)": wait}
    a = kinds.setdefault(kind, {"n": 0, "res": 0})
    a["n"] += 1
    a["res"] += res
    for k, v in ph.items():
        a[k] = a.get(k, 0) + v
for kind, a in kinds.items():
    n = a["n"]
    print(f"{kind}: {n} pair tasks, residency {a['res'] / n / 100:.1f} us, {100.0 * a['res'] / tot_slot:.1f} % of slot time")
    print("   " + " | ".join(f"{k} {v / n / 100:.1f}" for k, v in a.items() if k not in ("n", "res")))
mf = sum(a.get("mfma", 0) for a in kinds.values())
print(f"slot time {tot_slot / 1e8 * 1e3:.2f} ms, inside MFMA loops {100.0 * mf / tot_slot:.1f} %")
