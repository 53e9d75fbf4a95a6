#!/bin/bash
# Round 3, second measurement set (after the inlined finalisation): the driver's command and cfg2 (64 per launch) under rocprofv3
# (kernel-trace statistics + PMC passes) -> gpurun_out/r03d_*; then the randomised stress runs.
R=$GRAFT_REPO_ROOT
TAG=${1:-r03d}
cd $R
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json 2>/dev/null
run() {   # name, kernel, config, mats, bench arguments...
  local name=$1 kern=$2 cfg=$3 mats=$4; shift 4
  bash $R/tools/profile_driver.sh ${TAG}_$name "$@" || exit 1
  cd $R
  python3 tools/pmc_driver_summarize.py gpurun_out/${TAG}_$name --config $cfg --kernel $kern --mats $mats --update gpurun_out/pmc_traffic.json \
    --source "profiles/${TAG}_${name}_pmc_summary.txt: rocprofv3 --pmc over python3 bench.py $* (FETCH_SIZE x 2 + WRITE_SIZE, last timed launch)" \
    > gpurun_out/${TAG}_${name}_pmc_summary.txt || exit 1
  cat gpurun_out/${TAG}_${name}_pmc_summary.txt
  cut -c1-600 gpurun_out/${TAG}_${name}_bench.json
}
run drv pair128_chol_kernel cfg3 5,10,10 --gpus 1 --steps 20 --warmup 5
run cfg2 pair128_chol_kernel cfg2 64,64 --config cfg2 --steps 64 --warmup 64 --no-cpu-baseline
