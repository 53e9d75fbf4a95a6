#!/bin/bash
# Round 3, second measurement set, part 2: cfg2 one evaluation at a time and cfg5 under rocprofv3 (kernel-trace statistics + PMC passes)
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
  cat gpurun_out/${TAG}_${name}_pmc_summary.txt | tail -4
  cut -c1-300 gpurun_out/${TAG}_${name}_bench.json
}
run cfg2one tile_chol_kernel cfg2 1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1 --config cfg2 --steps 16 --warmup 4 --batch 0 --no-cpu-baseline
run cfg5 tile128_chol_kernel cfg5 1,1,1,1,1 --config cfg5 --steps 4 --warmup 1 --no-cpu-baseline
