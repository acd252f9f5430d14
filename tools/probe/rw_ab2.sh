#!/bin/bash
for i in 1 2; do
for cfg in "dev 0" "nobar 0" "abl 0" "abl 2" "abl 4" "abl 6"; do
  set -- $cfg
  echo "== lib $1 dbg $2"; RCPPML_GPU_LIB_PATH=$PWD/rcppml_amd/lib/RcppML_gpu_$1.so RCPPML_RW_DBG=$2 python tools/rhs_tiled_bench.py 8 107 2 2>&1 | grep -E "tiled kernel: rhs_H"
done; done
