#!/bin/bash
# C2, rate given by $RATE (default 107): the window kernel with forced workgroup shapes / partitions (dev build)
for cfg in "12 17 2 10" "8 26 2 10" "8 39 3 12" "12 14 2 8" "16 10 2 8" "8 16 1 5"; do
  set -- $cfg
  echo "== NW $1 NR $2 Ph $3 Pw $4"
  RCPPML_RW_NW=$1 RCPPML_RW_NR=$2 python tools/rhs_tiled_bench.py $4 ${RATE:-107} $3 2>&1 | grep -E "tiled kernel|^side" | sed -E "s/'slot_count.*'fill'/'fill'/; s/'stream_bytes.*//"
done
