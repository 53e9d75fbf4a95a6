#!/bin/bash
# pair128_chol_kernel: barrier spacing in the MFMA loop against memory traffic (FETCH_SIZE) and time
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $set | awk '{print $1}')
    O=$GRAFT_REPO_ROOT/gpurun_out/r03q_${v}_$tag
    GPG_PAIR=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O -- $GRAFT_REPO_ROOT/tools/$v 18048 5 10 /dev/null > $O.log 2>&1 || { echo "pmc $v $tag failed"; tail -5 $O.log; exit 1; }
    python3 $GRAFT_REPO_ROOT/tools/pmc_probe_print.py "$O" "$v" "$tag" | tail -1
  done
done
