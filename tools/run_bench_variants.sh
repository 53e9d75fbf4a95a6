for la in 0 1 2 3; do
timeout -k 10 200 python bench.py --prof-all --no-cpu-baseline --lookahead $la > gpurun_out/b3_la${la}_prof.json && python -c "
import json; r=json.load(open('gpurun_out/b3_la${la}_prof.json')); print('LA=$la prof', round(r['value'],2), round(r['ms_per_step'],2), round(r['roofline']['achieved'],1), {k: round(v,2) for k,v in r['kernel_ms_per_eval'].items()})"
timeout -k 10 200 python bench.py --no-cpu-baseline --lookahead $la > gpurun_out/b3_la${la}.json && python -c "
import json; r=json.load(open('gpurun_out/b3_la${la}.json')); print('LA=$la', round(r['value'],2), round(r['ms_per_step'],2), round(r['roofline']['achieved'],1))"
done
