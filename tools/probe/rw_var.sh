#!/bin/bash
for lib in base early; do for cfg in "12 17 2 10" "16 13 2 10" "8 26 2 10"; do
  set -- $cfg
  echo "== lib $lib NW $1 NR $2 Ph $3 Pw $4"
  RCPPML_GPU_LIB_PATH=$PWD/rcppml_amd/lib/RcppML_gpu_$lib.so RCPPML_RW_NW=$1 RCPPML_RW_NR=$2 python tools/rhs_tiled_bench.py $4 ${RATE:-107} $3 2>&1 | grep -E "tiled kernel"
done; done
