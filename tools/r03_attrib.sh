#!/bin/bash
# Round 3, item (a): where did 118.5 -> 135.7 GB per matrix come from?  The driver's command (without the CPU leg) under
# rocprofv3 for the product build (persistent workgroups + tickets) and for -DGPG_NO_PERSIST (round-1 schedule: one task per
# workgroup in dispatch order) on ONE box in ONE call.
R=$GRAFT_REPO_ROOT
cd $R
for v in default nopersist; do
  if [ $v = nopersist ]; then export GPG_LIB=libgpgrad_hip_nopersist.so; else unset GPG_LIB; fi
  bash tools/profile_driver.sh r03a_$v --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline || exit 1
  cd $R
  python3 tools/pmc_driver_summarize.py gpurun_out/r03a_$v --config cfg3 --mats 5,10,10 > gpurun_out/r03a_${v}_pmc_summary.txt || exit 1
  cat gpurun_out/r03a_${v}_pmc_summary.txt
done
