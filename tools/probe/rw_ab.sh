#!/bin/bash
# A/B of two dev builds of the window kernel on C2 (RcppML_gpu_dev.so vs RcppML_gpu_asm.so), three runs each, interleaved
for i in 1 2 3; do for lib in dev asm; do
  echo "== $lib"; RCPPML_GPU_LIB_PATH=$PWD/rcppml_amd/lib/RcppML_gpu_$lib.so python tools/rhs_tiled_bench.py 8 107 2 2>&1 | grep -E "tiled kernel"
done; done
