#!/usr/bin/env python3
"""Timing of the cross-validation half-updates (speckled mask, per-column Gram correction) at bench scale through the
device-level C ABI: H side and W side, zeros held out (the reference's default) or not."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcppml_amd import als, data
m, n, k = int(sys.argv[1]) if len(sys.argv) > 1 else 20000, int(sys.argv[2]) if len(sys.argv) > 2 else 100000, int(sys.argv[3]) if len(sys.argv) > 3 else 32
A, _, _ = data.simulate_nmf_sparse(m, n, k, 0.0115, seed=123, device=torch.device("cuda", 0))
At = A.transpose()
W0, H0 = data.init_factors(42, k, m, n, np.float32)
ops = als.HipOps(0, "f32")
W, H = ops.to_device(W0), ops.to_device(H0)
Ad, Atd = ops.upload_csc(A), ops.upload_csc(At)
print("A %d x %d nnz %d k %d" % (m, n, A.nnz, k))
for mz in (1, 0):
    for solver in (0, 1):
        for name, csc, F, X, nrows, tr in (("H", Ad, W, H, m, 0), ("W", Atd, H, W, n, 1)):
            G = ops.gram(F, 2e-15, 0.1)
            Xc = X.clone()
            def run():
                ops.ctx.solve_cv(ops.dt, csc["p"], csc["i"], csc["x"], csc["cols"], nrows, F, G, Xc, k, 0.1, 7, mask_zeros=mz, transposed=tr,
                                 cd_maxit=100, solver_mode=solver)
            run(); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); run(); e.record(); torch.cuda.synchronize()
            print("cv %s mask_zeros=%d solver=%s: %.2f ms" % (name, mz, "cd" if solver == 0 else "chol", s.elapsed_time(e)))
