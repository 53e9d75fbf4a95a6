#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes written by tools/pmc_gemm.sh (one warm-up + one timed evaluation,
single stream) into a per-kernel table.  FETCH_SIZE / WRITE_SIZE are in KiB as reported; FETCH_SIZE on gfx950
counts a wide streamed read at half its bytes (MI355X_MICROARCH.md, HBM section) - both raw and x2 shown."""
import collections
import csv
import glob
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
dur = collections.defaultdict(float)
import os
files = []
for dname in glob.glob(f"{root}/pmc_*/runc"):
    cand = sorted(glob.glob(f"{dname}/*_counter_collection.csv"), key=os.path.getmtime)
    if cand:
        files.append(cand[-1])          # newest pass only (gpurun_out accumulates older runs)
for f in files:
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add((f, r["Dispatch_Id"]))
        key = (f, r["Dispatch_Id"])
        if "SQ_WAVE_CYCLES" in f and key not in seen:
            seen.add(key)
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
print(f"{'kernel':44s} {'disp':>5s} {'ms':>8s} {'FETCH GB (x2)':>16s} {'WRITE GB':>9s} {'L2 hit':>7s} {'MFMA busy':>9s}")
for k, d in sorted(agg.items(), key=lambda kv: -dur[kv[0]]):
    n = len({x[1] for x in disp[k] if "SQ_WAVE_CYCLES" in x[0]}) or 1
    fetch = d.get("FETCH_SIZE", 0) * 1024 / 1e9
    write = d.get("WRITE_SIZE", 0) * 1024 / 1e9
    hit = d.get("TCC_HIT_sum", 0) / max(1.0, d.get("TCC_HIT_sum", 0) + d.get("TCC_MISS_sum", 0))
    # MFMA busy: SQ_VALU_MFMA_BUSY_CYCLES summed over SIMDs / (1024 SIMDs * kernel cycles); GRBM_GUI_ACTIVE is summed over 8 XCDs
    cyc = d.get("GRBM_GUI_ACTIVE", 0) / 8.0
    busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024.0 * cyc) if cyc > 0 else 0.0
    print(f"{k[:44]:44s} {n:5d} {dur[k]:8.3f} {fetch:7.2f} ({2*fetch:6.2f}) {write:9.2f} {hit:7.2f} {busy:9.2f}")
