#!/bin/bash
# rocprofv3 kernel trace of 320 one-point posterior evaluations (with gradients) of a 500-column model
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/lat_eval -- python3 $R/tools/lat_eval.py grad > $R/gpurun_out/lat_eval.log 2>&1 || { echo "profile run failed"; exit 1; }
grep "ms per call" $R/gpurun_out/lat_eval.log
python3 - "$(find $R/gpurun_out/lat_eval -name '*kernel_trace.csv' | head -1)" <<PY
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
i0 = len(rows) // 2
while 'cross_kernel' not in rows[i0]['Kernel_Name']: i0 += 1
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0 - 3:i0 + 16]:
    print('%8.1f %7.1f  %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Kernel_Name'][:80]))
PY
