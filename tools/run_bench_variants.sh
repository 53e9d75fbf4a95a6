# A/B of the factorisation schedules and the batch size on the headline workload (one process per variant)
for v in "auto -1" "auto 0" "auto 2" "auto 8" "tile128 0" "tile64 0" "blocked 0"; do
set -- $v; mode=$1; batch=$2
timeout -k 10 300 python bench.py --no-cpu-baseline --factor-mode $mode --batch $batch > gpurun_out/b_${mode}_b${batch}.json && python -c "
import json; r=json.load(open('gpurun_out/b_${mode}_b${batch}.json')); print('mode=$mode batch=$batch', round(r['value'],2), 'evals/s', round(r['ms_per_step'],2), 'ms', round(r['roofline']['achieved'],1), 'TF')"
done
