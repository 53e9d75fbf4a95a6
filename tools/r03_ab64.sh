#!/bin/bash
# A/B of two tile_probe builds on the 64-tile kernel: tools/r03_ab64.sh <probe A> <probe B> [reps]
cd $GRAFT_REPO_ROOT/tools
A=$1; B=$2; R=${3:-2}
for rep in $(seq 1 $R); do
  for cfg in "640 2" "1280 2" "2560 2" "5120 2" "9216 2" "1280 6 64" "640 6 64"; do
    echo -n "$A $cfg: "; timeout -k 10 120 ./$A $cfg /dev/null || exit 1
    echo -n "$B $cfg: "; timeout -k 10 120 ./$B $cfg /dev/null || exit 1
  done
done
