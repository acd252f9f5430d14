#!/usr/bin/env python3
"""One configuration of the fp32 k = 32 NB-IRLS half-update on the C5 shape, a few launches (for rocprofv3 passes):
usage irls_one.py <side H|W> <columns per wave 1|4> <cd_maxit> <irls_max_iter>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rcppml_amd import als, data, _abi
side, cpw, cdm, irm = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
m, n, k = 10000, 200000, 32
A, _, _ = data.simulate_nb_counts(m, n, k, density=0.02, size=5.0, seed=123)
W0, H0 = data.init_factors(42, k, m, n, np.float32)
ops = als.HipOps(0, "f32")
W, H = ops.to_device(W0), ops.to_device(H0)
csc = ops.upload_csc(A if side == "H" else A.transpose())
theta = torch.full((m,), 10.0, dtype=W.dtype, device="cuda")
F, ncols = (W, n) if side == "H" else (H, m)
G = ops.gram(F, 1e-15, 0.0)
X = torch.zeros((ncols, k), dtype=W.dtype, device="cuda")
ops.ctx.set_option(_abi.OPT_IRLS_COLUMNS_PER_WAVE, cpw)
for _ in range(3):
    ops.ctx.solve_irls_nb(ops.dt, csc["p"], csc["i"], csc["x"], ncols, F, G, X, k, 0.0, 0.0, 1, cdm, irm, 1e-4,
                          theta if side == "H" else None, None if side == "H" else theta)
torch.cuda.synchronize()
