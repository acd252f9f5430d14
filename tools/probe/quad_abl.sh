#!/bin/bash
for lib in RcppML_gpu RcppML_gpu_NOWEIGHT RcppML_gpu_NOMFMA RcppML_gpu_NOGATHER; do echo "== $lib"; RCPPML_GPU_LIB_PATH=$PWD/rcppml_amd/lib/$lib.so timeout 600 python tools/probe/irls_quad_probe.py 2>&1 | grep -E "side H cd_maxit   1"; done
