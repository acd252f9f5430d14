#!/bin/bash
# rhs time of the window plans against the number of row partitions (rate 1.75 and auto)
for cfg in "0 0 0" "8 107 2" "8 107 1" "8 107 3" "8 107 4" "4 107 2" "3 107 2" "2 107 2" "6 107 2" "0 107 0"; do
  echo "== Pw S Ph = $cfg"; python tools/rhs_tiled_bench.py $cfg 2>&1 | grep -E "^side|tiled kernel"
done
