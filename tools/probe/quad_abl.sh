#!/bin/bash
# One-term-at-a-time ablations of the quad IRLS kernel's weighted-Gram phase (results: profiles/r04_irls_pmc.txt).  The variant
# libraries were built with tools/quick_tu.sh from TEMPORARY edits of irls_nb_mfma32q_kernel that are not kept in the tree:
#   NOWEIGHT  w = 1.5f + 0.f * (th + recon) instead of the NB weight;   NOMFMA  acc[u] += ws[u].x * fv[u] instead of the MFMA;
#   NOGATHER  fv4 = (0.01 row, 0.02, a, 0.04) instead of the load of the row of F.  (With irls_max_iter = 5 the fake data lets the
#   IRLS loop converge early -- only the one-pass rows are clean ablations.)
for lib in RcppML_gpu RcppML_gpu_NOWEIGHT RcppML_gpu_NOMFMA RcppML_gpu_NOGATHER; do echo "== $lib"; RCPPML_GPU_LIB_PATH=$PWD/rcppml_amd/lib/$lib.so timeout 600 python tools/probe/irls_quad_probe.py 2>&1 | grep -E "side H cd_maxit   1"; done
