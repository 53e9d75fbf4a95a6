#!/bin/bash
# Round 3: the re-pipelined MFMA loop (exact s_waitcnt, no VALU in the loop) against the previous build (tile_probe_ks16)
cd $GRAFT_REPO_ROOT/tools
for cfg in "18048 5 10" "18048 1" "2560 5 64" "4608 5 16" "9216 1" "9216 5 8"; do
  echo -n "old: "; GPG_PAIR=0 timeout -k 10 120 ./tile_probe_ks16 $cfg /dev/null || exit 1
  echo -n "new: "; GPG_PAIR=0 timeout -k 10 120 ./tile_probe $cfg /dev/null || exit 1
  echo -n "new pair: "; GPG_PAIR=1 timeout -k 10 60 ./tile_probe $cfg /dev/null || exit 1
done
cd $GRAFT_REPO_ROOT
GPG_PAIR=0 timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print({k:d.get(k) for k in ('value','parity_ln_lkd_rel_err_row0','factor_fallbacks')}, d['roofline']['achieved'], d['cpu_baseline']['threads_tried'])"
