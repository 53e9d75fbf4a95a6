#!/bin/bash
# PMC passes over a short single-stream run (one evaluation); summaries land in gpurun_out/pmc_*/
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --no-cpu-baseline --steps 1 --warmup 1"
rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | awk '{print $1}')
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $ARGS > $R/gpurun_out/pmc_$tag.log 2>&1 || echo "pass $tag failed"
done
ls $R/gpurun_out/pmc_*/*/ | head -30
