#!/bin/bash
# A/B of the in-loop workgroup barrier (GPG_KSYNC k-steps apart) in the 128-tile factorisation: time, then FETCH / L2-hit counters.
# usage: tools/r03_ksync.sh <probe suffix> ...      (binaries tools/tile_probe_<suffix>)
cd $GRAFT_REPO_ROOT/tools
for v in "$@"; do
  for cfg in "18048 5 10" "18048 1" "2560 5 64"; do
    echo -n "$v: "; timeout -k 10 120 ./tile_probe_$v $cfg /dev/null || exit 1
  done
done
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $set | awk '{print $1}')
    O=$GRAFT_REPO_ROOT/gpurun_out/r03k_${v}_$tag
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O -- $GRAFT_REPO_ROOT/tools/tile_probe_$v 18048 5 10 /dev/null > $O.log 2>&1 || { echo "pmc $v $tag failed"; tail -5 $O.log; exit 1; }
    python3 $GRAFT_REPO_ROOT/tools/pmc_probe_print.py "$O" "$v" "$tag"
  done
done
