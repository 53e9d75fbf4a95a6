#!/bin/bash
# Round 3, last measurement set (after the batch rule change: the driver's 20 timed evaluations are ONE launch): the driver's command
# under rocprofv3 (kernel-trace statistics + PMC passes) -> gpurun_out/r03f_drv_*
R=$GRAFT_REPO_ROOT
TAG=${1:-r03f}
cd $R
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json 2>/dev/null
bash $R/tools/profile_driver.sh ${TAG}_drv --gpus 1 --steps 20 --warmup 5 || exit 1
cd $R
python3 tools/pmc_driver_summarize.py gpurun_out/${TAG}_drv --config cfg3 --kernel pair128_chol_kernel --mats 5,20 --update gpurun_out/pmc_traffic.json \
  --source "profiles/${TAG}_drv_pmc_summary.txt: rocprofv3 --pmc over python3 bench.py --gpus 1 --steps 20 --warmup 5 (FETCH_SIZE x 2 + WRITE_SIZE, last timed launch)" \
  > gpurun_out/${TAG}_drv_pmc_summary.txt || exit 1
cat gpurun_out/${TAG}_drv_pmc_summary.txt
cut -c1-400 gpurun_out/${TAG}_drv_bench.json
