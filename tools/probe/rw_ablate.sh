#!/bin/bash
# ablation of the window rhs kernel on C2 (H side numbers): needs RcppML_gpu_abl.so (-DRW_ABLATE) and RcppML_gpu_nc.so (+ -DRW_NO_COMPUTE)
for cfg in "abl 0" "abl 2" "abl 4" "abl 6" "nc 0" "nc 2" "nc 4" "nc 6"; do
  set -- $cfg
  echo "== lib $1 dbg $2"
  RCPPML_GPU_LIB_PATH=$PWD/rcppml_amd/lib/RcppML_gpu_$1.so RCPPML_RW_DBG=$2 python tools/rhs_tiled_bench.py 0 ${RATE:-107} 0 2>&1 | grep "tiled kernel"
done
