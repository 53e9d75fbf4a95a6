#!/bin/bash
# rocprofv3 kernel-trace statistics of value + gradient evaluations (tools/grad_time.py)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_grad -- python3 $R/tools/grad_time.py > $R/gpurun_out/prof_grad.log 2>&1 || echo "profile run failed"
f=$(find $R/gpurun_out/prof_grad -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<PY
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"])
PY
