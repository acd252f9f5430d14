#!/bin/bash
for cfg in "nc 6" "ncsr 6" "ncbr 6" "ncsrbr 6" "ncsrbr 0"; do
  set -- $cfg
  echo "== lib $1 dbg $2"
  RCPPML_GPU_LIB_PATH=$PWD/rcppml_amd/lib/RcppML_gpu_$1.so RCPPML_RW_DBG=$2 python tools/rhs_tiled_bench.py 0 ${RATE:-107} 0 2>&1 | grep "tiled kernel"
done
