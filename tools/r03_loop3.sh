#!/bin/bash
cd $GRAFT_REPO_ROOT/tools
./fin_probe | head -1
for rep in 1 2; do
for cfg in "18048 5 10" "2560 5 64" "4608 5 16" "9216 1" "2560 2" "18048 1"; do
  for v in tile_probe_ks16 tile_probe tile_probe_pf3; do
    echo -n "$v: "; GPG_PAIR=0 timeout -k 10 120 ./$v $cfg /dev/null || exit 1
  done
  echo -n "pair: "; GPG_PAIR=1 timeout -k 10 120 ./tile_probe $cfg /dev/null || exit 1
done
done
