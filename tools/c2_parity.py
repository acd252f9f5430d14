#!/usr/bin/env python3
"""North-star parity figure at FULL size: BASELINE configs[1] (20 000 x 100 000, 1 %, k = 64, CD), GPU plugin vs the CPU
oracle on identical inputs, same iteration count (tol = 0): relative deviation of the loss (target <= 1e-6 in fp64) and
of d, max-abs deviation of the L1-normalised W and H."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import oracle as O
from oracle.oracle import Csc
from rcppml_amd import _abi, data
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
m, n, k = 20000, 100000, 64
A, _, _ = data.simulate_nmf_sparse(m, n, k, 0.0115, seed=123, device=torch.device("cuda", 0))
W0, H0 = data.init_factors(42, k, m, n, np.float64)
p, i, x = A.p.astype(np.int32), A.i.astype(np.int32), A.x.astype(np.float64)
Ao = Csc((m, n), A.p, A.i, A.x)
try:
    O.build(native=True); native = True
except Exception:
    native = False
for entry, dtype in (("double", np.float64), ("float", np.float32)):
    W, H = W0.copy(), H0.copy()
    t0 = time.perf_counter()
    res = _abi.nmf_unified(p, i, x, m, n, k, W, H, entry=entry, max_iter=iters, tol=0.0, solver_mode=0)
    tg = time.perf_counter() - t0
    t0 = time.perf_counter()
    ref = O.nmf_fit(Ao, W0, H0, dtype, max_iter=iters, tol=0.0, solver_mode=0, threads=0, native=native)
    tc = time.perf_counter() - t0
    print("%s: %d iterations  GPU %.3fs  CPU oracle (%d threads) %.1fs  loss gpu %.12g ref %.12g rel %.2e  d rel %.2e  max|dW| %.2e  max|dH| %.2e" % (
        entry, iters, tg, O.num_threads(), tc, res["loss"], ref.loss, abs(res["loss"] - ref.loss) / abs(ref.loss),
        np.abs(res["d"] - ref.d).max() / np.abs(ref.d).max(), np.abs(W - ref.W_T).max(), np.abs(H - ref.H).max()))
