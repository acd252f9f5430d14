#!/bin/bash
for r in 104 106 107 108; do
  RCPPML_RW_NW=${NW:-12} RCPPML_RW_NR=${NR:-17} tools/rprof.sh r$r python tools/rhs_tiled_bench.py 10 $r 2 | grep -E "rhs_win_kernel|finish" | sed -E 's/\(.*\)//' | sed "s/^/rate $r: /"
done
