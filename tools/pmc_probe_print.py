"""Print the per-dispatch counter totals of tile128_chol_kernel from one rocprofv3 --pmc output directory (tools/r03_ksync.sh)."""
import collections
import csv
import glob
import sys

root, v, tag = sys.argv[1:4]
f = sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True))[-1]
tot = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "tile128_chol_kernel" not in r["Kernel_Name"] and "pair128_chol_kernel" not in r["Kernel_Name"]:
        continue
    d = tot.setdefault(int(r["Dispatch_Id"]), collections.defaultdict(float))
    d[r["Counter_Name"]] += float(r["Counter_Value"])
    d["_ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
for k, d in tot.items():
    print(f"{v} {tag} dispatch {k}: " + ", ".join(f"{n}={x:.5g}" for n, x in d.items()))
