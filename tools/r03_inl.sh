#!/bin/bash
# Round 3: finalisation inlined into the 128-tile kernels (tile_probe) against the out-of-line call (tile_probe_prev, -DGPG_FIN_INLINE=0)
cd $GRAFT_REPO_ROOT/tools
for rep in 1 2; do
  for cfg in "2560 5 64" "4608 5 16" "9216 5 8" "18048 5 10" "2560 1" "9216 1" "18048 1"; do
    echo -n "prev $cfg: "; timeout -k 10 120 ./tile_probe_prev $cfg /dev/null || exit 1
    echo -n "inl  $cfg: "; timeout -k 10 120 ./tile_probe $cfg /dev/null || exit 1
  done
done
