#!/bin/bash
# traffic / L2 / clock counters of the current build: tile128_chol_kernel (GPG_PAIR=0) against pair128_chol_kernel (GPG_PAIR=1), ten cfg3-size matrices
cd /tmp && export TMPDIR=/tmp
for p in 0 1; do
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $set | awk '{print $1}')
    O=$GRAFT_REPO_ROOT/gpurun_out/r03p_pair${p}_$tag
    GPG_PAIR=$p timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O -- $GRAFT_REPO_ROOT/tools/tile_probe_cur 18048 5 10 /dev/null > $O.log 2>&1 || { echo "pmc $p $tag failed"; tail -5 $O.log; exit 1; }
    python3 $GRAFT_REPO_ROOT/tools/pmc_probe_print.py "$O" "pair=$p" "$tag" | tail -1
  done
done
