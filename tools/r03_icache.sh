#!/bin/bash
# instruction-cache counters of the 128-tile factorisation (is the 123-KB straight-line finalisation I-cache bound?)
cd /tmp && export TMPDIR=/tmp
for cfg in "2560 5 64" "18048 5 10"; do
  tag=$(echo $cfg | awk '{print $1}')
  for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
    t=$(echo $set | awk '{print $1}')
    O=$GRAFT_REPO_ROOT/gpurun_out/r03i_${tag}_$t
    GPG_PAIR=0 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O -- $GRAFT_REPO_ROOT/tools/tile_probe $cfg /dev/null > $O.log 2>&1 || { echo "pmc $tag $t failed"; tail -3 $O.log; continue; }
    python3 $GRAFT_REPO_ROOT/tools/pmc_probe_print.py "$O" "$tag" "$t" | tail -1
  done
done
