#!/bin/bash
# A/B of the dataflow-tail switch point (diagnostic): prints evals/s per variant
cd "$(dirname "$0")/.."
run() { echo -n "$1 : "; env $2 timeout -k 10 200 python bench.py --no-cpu-baseline $3 2>/dev/null | grep -o '"value": [0-9.]*, "unit"\|"achieved": [0-9.]*\|"kernel_ms_per_eval.*' | head -3 | tr '\n' ' '; echo; }
for t in ${TAILS:-0 4096 8192}; do run "cfg3 tail=$t" "GPG_TAIL_COLS=$t" "$EXTRA"; done
run "cfg2 tail=0" "GPG_TAIL_COLS=0" "--config cfg2 $EXTRA"
run "cfg2 tail=4096" "GPG_TAIL_COLS=4096" "--config cfg2 $EXTRA"
