#!/bin/bash
# rocprofv3 kernel-trace statistics of one-at-a-time evaluations at cfg2 size (n=500, d=4)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_small -- python3 $R/bench.py --no-cpu-baseline --config cfg2 --steps 64 --warmup 8 --batch 0 > $R/gpurun_out/prof_small.log 2>&1 || echo "profile run failed"
f=$(find $R/gpurun_out/prof_small -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<PY
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(r["Name"][:80], r["Calls"], r["AverageNs"], r["Percentage"])
PY
tail -1 $R/gpurun_out/prof_small.log | cut -c1-300
