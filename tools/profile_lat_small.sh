#!/bin/bash
# rocprofv3 kernel-trace statistics of 320 one-at-a-time evaluations of a 100-column problem (value, then value + gradient)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for m in value grad; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/lat_small_$m -- python3 $R/tools/lat_small.py $m > $R/gpurun_out/lat_small_$m.log 2>&1 || { echo "profile run failed"; exit 1; }
  grep "ms per call" $R/gpurun_out/lat_small_$m.log
  f=$(find $R/gpurun_out/lat_small_$m -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<PY
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:24]:
    print(r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"])
PY
done
