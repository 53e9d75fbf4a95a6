#!/bin/bash
# the DEFAULT bench invocation (python3 bench.py: 8 steps, 8 warm-up = launches of 8 matrices) under rocprofv3 -> the 8-matrix entry of pmc_traffic.json
R=$GRAFT_REPO_ROOT
cd $R
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
bash $R/tools/profile_driver.sh r03g_default --gpus 1 --steps 8 --warmup 8 --no-cpu-baseline || exit 1
cd $R
python3 tools/pmc_driver_summarize.py gpurun_out/r03g_default --config cfg3 --kernel pair128_chol_kernel --mats 8,8 --update gpurun_out/pmc_traffic.json \
  --source "profiles/r03g_default_pmc_summary.txt: rocprofv3 --pmc over python3 bench.py --gpus 1 --steps 8 --warmup 8 (FETCH_SIZE x 2 + WRITE_SIZE, last timed launch)" \
  > gpurun_out/r03g_default_pmc_summary.txt || exit 1
cat gpurun_out/r03g_default_pmc_summary.txt | tail -3
