#!/bin/bash
# rocprofv3 kernel-trace statistics of the default bench command (summary copied to profiles/ by hand)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1 || echo "profile run failed"
ls $R/gpurun_out/prof_bench/*/ | head
