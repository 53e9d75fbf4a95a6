#!/bin/bash
# generic A/B of two tile_probe builds on one box: tools/r03_ab.sh <probe A> <probe B> [reps]
cd $GRAFT_REPO_ROOT/tools
A=$1; B=$2; R=${3:-2}
for rep in $(seq 1 $R); do
  for cfg in "2560 5 64" "4608 5 16" "9216 5 8" "18048 5 10" "2560 1" "9216 1" "18048 1"; do
    echo -n "$A $cfg: "; timeout -k 10 120 ./$A $cfg /dev/null || exit 1
    echo -n "$B $cfg: "; timeout -k 10 120 ./$B $cfg /dev/null || exit 1
  done
done
