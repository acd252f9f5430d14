#!/bin/bash
# profiles/ablate_rhs_tiled.sh -- what bounds rhs_tiled_kernel: the same C2 launches with parts of the kernel switched off
# (RCPPML_RT_DBG bit 1 = no compute loop, bit 2 = no slab LDS-DMA, bit 4 = no slot loads; wrong results by construction).
# Needs the probe build: make -C rcppml_amd/csrc EXPERIMENTS=1  (-> rcppml_amd/lib/RcppML_gpu_exp.so; the shipping
# library has none of these switches).  Output: gpurun_out/r02/ablate_rhs_tiled.txt
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export RCPPML_GPU_LIB_PATH=$PWD/rcppml_amd/lib/RcppML_gpu_exp.so
OUT=gpurun_out/r02; mkdir -p $OUT
for d in 0 1 2 4 6 7; do
  D=$OUT/abl_$d; rm -rf $D
  RCPPML_RT_DBG=$d rocprofv3 --output-format csv --kernel-trace -d $D -o t -- python tools/rhs_tiled_bench.py > $D.log 2>&1
  echo "RCPPML_RT_DBG=$d"; python profiles/sumtrace.py $(find $D -name "*kernel_trace.csv" | head -1) | grep "rhs_tiled_kernel"
done | tee $OUT/ablate_rhs_tiled.txt
