#!/usr/bin/env python3
"""End-to-end time of the 73-pointer plugin call on BASELINE configs[1] (host buffers in, host buffers out):
intercept (upload, host/device transpose, download) and slope (ms per ALS iteration)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcppml_amd import _abi, data
m, n, k = 20000, 100000, 64
A, _, _ = data.simulate_nmf_sparse(m, n, k, 0.0115, seed=123, device=torch.device("cuda", 0))
W0, H0 = data.init_factors(42, k, m, n, np.float64)
p, i, x = A.p.astype(np.int32), A.i.astype(np.int32), A.x.astype(np.float64)
res = {}
for iters in (1, 1, 11, 21):
    W, H = W0.copy(), H0.copy()
    t0 = time.perf_counter()
    r = _abi.nmf_unified(p, i, x, m, n, k, W, H, entry="float", max_iter=iters, tol=0.0, solver_mode=0)
    dt = time.perf_counter() - t0
    assert r["status"] == 0 and r["iter"] == iters
    res[iters] = dt
    print("max_iter=%2d: %.1f ms total (loss %.6g)" % (iters, dt * 1e3, r["loss"]))
slope = (res[21] - res[11]) / 10
print("slope %.3f ms/iteration, intercept %.1f ms (nnz %d: %.0f MB of CSC in, %.0f MB of factors in+out)" % (
    slope * 1e3, (res[11] - 11 * slope) * 1e3, A.nnz, (A.nnz * 12 + n * 4) / 1e6, 2 * (m + n) * k * 8 / 1e6))
# one-time phases of the call (RCPPML verbose = 2: wall time per setup phase, to stderr)
W, H = W0.copy(), H0.copy()
_abi.nmf_unified(p, i, x, m, n, k, W, H, entry="float", max_iter=1, tol=0.0, solver_mode=0, verbose=2)
