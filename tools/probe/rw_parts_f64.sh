#!/bin/bash
# fp64 C2: rhs time of the window plans against partitions and rates (Pw S Ph; S = 100 + 4 x rate, 0 = planner's choice)
for cfg in "0 0 0" "0 0 1" "0 0 2" "0 0 3" "0 0 4" "4 0 2" "8 0 2" "16 0 2" "0 104 0" "0 105 0" "0 106 0" "0 107 0" "0 108 0"; do
  echo "== Pw S Ph = $cfg"; DT=f64 python tools/rhs_tiled_bench.py $cfg 2>&1 | grep -E "^side|tiled kernel" | cut -c1-330
done
