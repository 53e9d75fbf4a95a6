#!/bin/bash
# A/B of panel-width schedules (diagnostic): prints evals/s per variant
cd "$(dirname "$0")/.."
run() { echo -n "$1 : "; env $2 python bench.py --no-cpu-baseline $3 2>/dev/null | grep -o '"value": [0-9.]*, "unit"\|"achieved": [0-9.]*' | head -2 | tr '\n' ' '; echo; }
run "base512" "GPG_X=0" ""
run "big1024>=6000" "GPG_NB_BIG=1024 GPG_BIG_ROWS=6000" ""
run "big1024>=10000" "GPG_NB_BIG=1024 GPG_BIG_ROWS=10000" ""
run "big1024>=14000" "GPG_NB_BIG=1024 GPG_BIG_ROWS=14000" ""
run "big768>=6000" "GPG_NB_BIG=768 GPG_BIG_ROWS=6000" ""
run "panel768" "GPG_X=0" "--panel 768"
run "panel384" "GPG_X=0" "--panel 384"
run "big512>=6000,base256" "GPG_NB_BIG=512 GPG_BIG_ROWS=6000" "--panel 256"
run "big512>=3000,base256" "GPG_NB_BIG=512 GPG_BIG_ROWS=3000" "--panel 256"
