#!/bin/bash
# Round-2 measurement set (run on the GPU box through gpurun): the driver's own command under rocprofv3 (kernel-trace
# statistics + PMC passes), summarised into gpurun_out/<tag>_pmc_summary.txt and gpurun_out/pmc_traffic.json, then the
# cfg2 / cfg5 bench lines with their kernel statistics.  Copy what is to be judged into profiles/.
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
bash $R/tools/profile_driver.sh ${TAG}_drv --gpus 1 --steps 20 --warmup 5 || exit 1
cd $R
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json 2>/dev/null
python3 tools/pmc_driver_summarize.py gpurun_out/${TAG}_drv --config cfg3 --mats 5,10,10 --update gpurun_out/pmc_traffic.json \
  --source "profiles/${TAG}_drv_pmc_summary.txt: rocprofv3 --pmc over python3 bench.py --gpus 1 --steps 20 --warmup 5 (FETCH_SIZE x 2 + WRITE_SIZE, last timed launch)" \
  > gpurun_out/${TAG}_drv_pmc_summary.txt || exit 1
cat gpurun_out/${TAG}_drv_pmc_summary.txt
cd /tmp && export TMPDIR=/tmp
for cfg in cfg2 cfg5; do
  if [ $cfg = cfg2 ]; then A="--config cfg2 --steps 64 --warmup 64"; else A="--config cfg5 --steps 4 --warmup 1"; fi
  mkdir -p $R/gpurun_out/${TAG}_$cfg
  timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_$cfg/stats -- python3 $R/bench.py $A > $R/gpurun_out/${TAG}_$cfg/bench.log 2> $R/gpurun_out/${TAG}_$cfg/bench.err || { echo "$cfg failed"; exit 1; }
  tail -1 $R/gpurun_out/${TAG}_$cfg/bench.log > $R/gpurun_out/${TAG}_bench_$cfg.json
  cp $(find $R/gpurun_out/${TAG}_$cfg/stats -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_${cfg}_kernel_stats.csv
  cut -c1-400 $R/gpurun_out/${TAG}_bench_$cfg.json
done
