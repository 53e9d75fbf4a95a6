#!/bin/bash
# rocprofv3 over the DRIVER'S OWN bench command: kernel-trace statistics, then the PMC passes (FETCH_SIZE and
# WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes; no --sys-trace / --hip-trace with --pmc).
#   tools/profile_driver.sh <tag> [bench.py arguments ...]      default arguments: --gpus 1 --steps 20 --warmup 5
# Results: gpurun_out/<tag>_bench.json, <tag>_kernel_stats.csv, <tag>_kernel_trace.csv, <tag>_pmc/<counter>/...; run
# tools/pmc_driver_summarize.py on them afterwards (tools/pmc_collect_all.sh does both).
R=$GRAFT_REPO_ROOT
TAG=$1; shift
ARGS="$@"
[ -z "$ARGS" ] && ARGS="--gpus 1 --steps 20 --warmup 5"
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/${TAG}
mkdir -p $OUT
echo "python3 bench.py $ARGS" > $OUT/command.txt
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS > $OUT/bench_stats.log 2> $OUT/bench_stats.err || { echo "stats run failed"; exit 1; }
tail -1 $OUT/bench_stats.log > $R/gpurun_out/${TAG}_bench.json
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv
cp $(find $OUT/stats -name "*kernel_trace.csv" | head -1) $R/gpurun_out/${TAG}_kernel_trace.csv
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS"; do
  tag=$(echo $set | awk '{print $1}')
  timeout -k 10 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$tag -- python3 $R/bench.py $ARGS > $OUT/pmc_$tag.log 2>&1 || { echo "pass $tag failed"; exit 1; }
  echo "pass $tag done"
done
