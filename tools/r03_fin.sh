#!/bin/bash
# Round 3: fin128_offdiag (LDS hand-over of column block 0, one-piece images when the diagonal tile is complete) against the build before it
cd $GRAFT_REPO_ROOT/tools
for rep in 1 2; do
for cfg in "18048 5 10" "2560 5 64" "4608 5 16" "9216 1" "18048 1" "9216 5 8"; do
  for p in 0 1; do
    echo -n "prev pair=$p: "; GPG_PAIR=$p timeout -k 10 120 ./tile_probe_prev $cfg /dev/null || exit 1
    echo -n "new  pair=$p: "; GPG_PAIR=$p timeout -k 10 120 ./tile_probe $cfg /dev/null || exit 1
  done
done
done
