for cfg in "256 256" "384 384" "512 512"; do
set -- $cfg; pn=$1; sp=$2
timeout -k 10 200 python bench.py --prof-all --no-cpu-baseline --panel $pn --super $sp > gpurun_out/b_p${pn}_s${sp}_prof.json && python -c "
import json; r=json.load(open('gpurun_out/b_p${pn}_s${sp}_prof.json')); print('panel=$pn super=$sp prof', round(r['value'],2), round(r['ms_per_step'],2), round(r['roofline']['achieved'],1), {k: round(v,2) for k,v in r['kernel_ms_per_eval'].items()})"
timeout -k 10 200 python bench.py --no-cpu-baseline --panel $pn --super $sp > gpurun_out/b_p${pn}_s${sp}.json && python -c "
import json; r=json.load(open('gpurun_out/b_p${pn}_s${sp}.json')); print('panel=$pn super=$sp', round(r['value'],2), round(r['ms_per_step'],2), round(r['roofline']['achieved'],1))"
done
