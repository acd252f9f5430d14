#!/usr/bin/env python3
"""Diagnostic: distribution of CD sweeps per column (what cd_nnls_col_fixed returns) on the bench workload,
per ALS iteration and side, for the lane kernel.  Run on the GPU box."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcppml_amd import als, data

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
m, n, k = 20000, 100000, 64
A, _, _ = data.simulate_nmf_sparse(m, n, k, 0.0115, seed=123, device=torch.device("cuda", 0))
At = A.transpose()
nd = np.float32 if dtype == "f32" else np.float64
W0, H0 = data.init_factors(42, k, m, n, nd)
ops = als.HipOps(0, dtype)
cfg = als.AlsConfig(k=k)
W, H = ops.to_device(W0), ops.to_device(H0)
Ad, Atd = ops.upload_csc(A), ops.upload_csc(At)
sums, d = ops.empty((k,)), ops.empty((k,))
for it in range(8):
    for side in ("H", "W"):
        F, X, csc = (W, H, Ad) if side == "H" else (H, W, Atd)
        G = ops.gram(F, 1e-15, 0.0)
        B = ops.rhs(csc, F)
        sw = torch.zeros(X.shape[0], dtype=torch.int32, device="cuda")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ops.ctx.solve_cd(ops.dt, G, B, X, k, X.shape[0], warm=int(it > 0), maxit=100, tol=1e-8, sweeps_out=sw)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        s = sw.cpu().numpy()
        wavemax = s[: (len(s) // 64) * 64].reshape(-1, 64).max(axis=1)
        nz = float((X != 0).float().mean().item())
        print("it %d %s: %.2f ms  sweeps mean %.1f median %d p90 %d p99 %d max %d | per-wave max: mean %.1f | frac(100) %.3f | nonzero frac of X %.3f"
              % (it, side, (t1 - t0) * 1e3, s.mean(), np.median(s), np.percentile(s, 90), np.percentile(s, 99), s.max(),
                 wavemax.mean(), (s >= 100).mean(), nz))
        ops.row_norms(X, 0, out=sums)
        ops.apply_scaling(X, sums, 0, d)
