#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_gs -- python3 $R/tools/grad_small.py > $R/gpurun_out/prof_gs.log 2>&1 || echo "profile run failed"
f=$(find $R/gpurun_out/prof_gs -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<PY
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU ms per evaluation %.3f, launches per evaluation %.1f" % (tot / 21e6, sum(int(r["Calls"]) for r in rows) / 21.0))
for r in rows[:12]:
    print(r["Name"][:80], r["Calls"], r["AverageNs"], r["Percentage"])
PY
grep "value + gradient" $R/gpurun_out/prof_gs.log
