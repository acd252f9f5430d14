#!/usr/bin/env python3
"""fp32 C5 iteration by iteration (bench_c5's step): NB likelihood after every outer iteration, and the relative change of H after
the first half-update, for two builds of the library (RCPPML_GPU_LIB_PATH) -- how fast two arithmetic orders of the same
algorithm drift apart."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rcppml_amd import als, data
m, n, k = 10000, 200000, 32
A, _, _ = data.simulate_nb_counts(m, n, k, density=0.02, size=5.0, seed=123)
At = A.transpose()
W0, H0 = data.init_factors(42, k, m, n, np.float32)
ops = als.HipOps(0, "f32", record_events=False)
W, H = ops.to_device(W0), ops.to_device(H0)
Ad, Atd = ops.upload_csc(A), ops.upload_csc(At)
theta = torch.full((m,), 10.0, dtype=ops.tdtype, device="cuda")
d = torch.ones((k,), dtype=ops.tdtype, device="cuda")
sums, G = ops.empty((k,)), ops.empty((k, k))
out4 = torch.zeros((4,), dtype=torch.float64, device="cuda")
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    for side in ("H", "W"):
        F, X, csc = (W, H, Ad) if side == "H" else (H, W, Atd)
        ops.gram(F, 1e-15, 0.0, out=G)
        ops.ctx.solve_irls(ops.dt, 5, csc["p"], csc["i"], csc["x"], csc["cols"], F, G, X, k, 0.0, 0.0, 1, 100, 5, 1e-4,
                           theta if side == "H" else None, None if side == "H" else theta)
        if it == 0 and side == "H":
            print("H after the first half-update: sum %.9e  max %.9e" % (float(H.double().sum()), float(H.max())))
        ops.row_norms(X, 0, out=sums)
        ops.apply_scaling(X, sums, 0, d)
    ops.ctx.nb_size_update_loss(ops.dt, Atd["p"], Atd["i"], Atd["x"], m, At.nnz, W, d, H, n, k, 0.01, 1e6, theta, out4)
    print("iteration %2d  NLL %.9e" % (it + 1, float(out4[0].item())))
