#!/bin/bash
# pair128_chol_kernel against tile128_chol_kernel (GPG_PAIR=0) on one box: time, then the bench with its parity check
cd $GRAFT_REPO_ROOT/tools
for cfg in "18048 5 10" "18048 5 5" "9216 5 8" "2560 5 64" "4608 5 16"; do
  for p in 0 1; do
    echo -n "pair=$p: "; GPG_PAIR=$p timeout -k 10 120 ./tile_probe $cfg /dev/null || exit 1
  done
done
for v in pk4 pk16; do echo -n "$v: "; timeout -k 10 120 ./tile_probe_$v 18048 5 10 /dev/null || exit 1; done
cd $GRAFT_REPO_ROOT
for p in 0 1; do GPG_PAIR=$p timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline | cut -c1-1400 || exit 1; done
timeout -k 10 300 python3 bench.py --steps 10 --warmup 5 | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print({k:d[k] for k in ('value','parity_ln_lkd_rel_err_row0','factor_fallbacks')}, d['roofline']['achieved'], d['cpu_baseline']['threads_tried'])"
