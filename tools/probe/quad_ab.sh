#!/bin/bash
for lib in ${LIBS:-RcppML_gpu}; do echo "== $lib"; RCPPML_GPU_LIB_PATH=$PWD/rcppml_amd/lib/$lib.so timeout 600 python tools/probe/irls_quad_probe.py 2>&1 | grep side; done
