#!/bin/bash
# rocprofv3 kernel-trace statistics of the batched value + gradient path (tools/grad_batch_run.py)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_gradb -- python3 $R/tools/grad_batch_run.py 8 > $R/gpurun_out/prof_gradb.log 2>&1 || echo "profile run failed"
f=$(find $R/gpurun_out/prof_gradb -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/r02_gradb_kernel_stats.csv
python3 - "$f" <<PY
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
cat $R/gpurun_out/prof_gradb.log
