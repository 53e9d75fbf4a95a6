"""Phase shares of the dataflow factorisation from gpurun_out/tile_timeline.csv (tools/tile_probe.hip -DGPG_STAMP):
ticks of s_memrealtime (100 MHz) per task, summed over the tasks whose tile column lies in [lo, hi)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tile_timeline.csv")))
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (40, 100)
tot = spin = gemm = load = finwait = finsolve = store = 0.0
nd = 0
for r in rows:
    tj, ti = int(r["tj"]), int(r["ti"])
    if not (lo <= tj < hi) or ti == tj:
        continue
    st, en, fin0 = int(r["start"]), int(r["end"]), int(r["fin0"])
    f0, f1, f2 = int(r["f0"]), int(r["f1"]), int(r["f2"])
    tot += en - st
    sp, gm = int(r["spin_cyc"]), int(r["gemm_cyc"])          # s_memrealtime ticks as well
    spin += sp; gemm += gm
    load += (fin0 - st) - sp - gm
    finwait += f1 - f0 if f1 > f0 else 0
    finsolve += f2 - f1 if f2 > f1 else 0
    store += (f0 - fin0) + (en - f2 if f2 else 0)
    nd += 1
print(f"tile columns [{lo},{hi}): {nd} off-diagonal tasks, mean residency {tot / nd / 100:.1f} us")
for k, v in (("wait for tile columns (spin)", spin), ("MFMA loop", gemm), ("tile load + loop overhead", load),
             ("tile store / reload around the finalisation", store), ("finalisation incl. waits for L11 / L21 / L22", finsolve + finwait)):
    print(f"  {k:48s} {100 * v / tot:5.1f} %")
