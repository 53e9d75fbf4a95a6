# A/B of the Cholesky options on the headline workload (one process per variant)
for cfg in "1 256" "0 256" "3 256" "1 512"; do
set -- $cfg; la=$1; pn=$2
timeout -k 10 200 python bench.py --prof-all --no-cpu-baseline --lookahead $la --panel $pn > gpurun_out/b_la${la}_p${pn}_prof.json && python -c "
import json; r=json.load(open('gpurun_out/b_la${la}_p${pn}_prof.json')); print('opts=$la panel=$pn prof', round(r['value'],2), round(r['ms_per_step'],2), round(r['roofline']['achieved'],1), {k: round(v,2) for k,v in r['kernel_ms_per_eval'].items()})"
timeout -k 10 200 python bench.py --no-cpu-baseline --lookahead $la --panel $pn > gpurun_out/b_la${la}_p${pn}.json && python -c "
import json; r=json.load(open('gpurun_out/b_la${la}_p${pn}.json')); print('opts=$la panel=$pn', round(r['value'],2), round(r['ms_per_step'],2), round(r['roofline']['achieved'],1))"
done
