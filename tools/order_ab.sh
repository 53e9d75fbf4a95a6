#!/bin/bash
# A/B of the two ticket orders of the dataflow factorisation (GPG_PROBE_ORDER = 0 column-major, 1 critical path first)
cd "$(dirname "$0")"
for cfg in "2560 2" "2560 6 64" "2560 5 64" "4608 2" "4608 6 16" "4608 5 16" "9216 2" "9216 1" "9216 5 8" "18048 1" "18048 5 10"; do
  for o in 0 1; do
    echo -n "order $o: "; GPG_PROBE_ORDER=$o timeout -k 10 120 ./tile_probe $cfg /dev/null || exit 1
  done
done
