#!/bin/bash
# kernel time of the window rhs against the slot rate (1.0 / 1.5 / 1.75 / 2.0 slots per column and phase) at one shape: T = a + b x steps
for r in 104 106 107 108; do
  RCPPML_GPU_LIB_PATH=${LIB:-$PWD/rcppml_amd/lib/RcppML_gpu.so} tools/rprof.sh r$r python tools/rhs_tiled_bench.py 8 $r 2 | grep -E "rhs_win_kernel|finish" | sed -E 's/\(.*\)//' | sed "s/^/rate $r: /"
done
