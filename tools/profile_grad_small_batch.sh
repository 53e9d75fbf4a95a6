#!/bin/bash
# rocprofv3 kernel-trace statistics of batched value + gradient calls at cfg2 size (5 rows per call, as the lock-step multi-start issues them)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/gb_small.py <<PY
import os, sys, time
import numpy as np
sys.path.insert(0, "$R")
import bench, gpgradpy_amd
n, d = 500, 4
X, f, g, tab = bench.make_workload(n, d)
GP = gpgradpy_amd.GaussianProcess(d, True, 'SqExp', 'precon')
GP.set_data(X, f, np.zeros(n), g, np.zeros((n, d)))
GP.calc_lkd_grad_batch(tab[:5])
t0 = time.perf_counter()
for _ in range(20): GP.calc_lkd_grad_batch(tab[:5])
print('cfg2 batched value + gradient: %.3f ms per call of 5 rows' % ((time.perf_counter() - t0) / 20 * 1e3))
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_gsb -- python3 /tmp/gb_small.py > $R/gpurun_out/prof_gsb.log 2>&1 || echo "profile run failed"
f=$(find $R/gpurun_out/prof_gsb -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<PY
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
grep "cfg2 batched" $R/gpurun_out/prof_gsb.log
