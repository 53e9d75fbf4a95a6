#!/bin/bash
# Round 3: the unscaled-chain substitution against the old one (fin_probe_old / tile_probe_ks16 were built before the change)
cd $GRAFT_REPO_ROOT/tools
echo "== fin_probe old"; timeout -k 10 120 ./fin_probe_old || exit 1
echo "== fin_probe new"; timeout -k 10 120 ./fin_probe || exit 1
for cfg in "18048 5 10" "18048 1" "2560 5 64" "2560 2" "1280 6 64" "4608 5 16" "9216 1"; do
  echo -n "old: "; GPG_PAIR=0 timeout -k 10 120 ./tile_probe_ks16 $cfg /dev/null || exit 1
  echo -n "new: "; GPG_PAIR=0 timeout -k 10 120 ./tile_probe $cfg /dev/null || exit 1
done
for cfg in "18048 5 10" "9216 5 8" "2560 5 64"; do
  echo -n "new pair: "; GPG_PAIR=1 timeout -k 10 60 ./tile_probe $cfg /dev/null || exit 1
done
