#!/usr/bin/env python3
"""Cross-validation fit, GPU plugin vs the CPU oracle (all host threads) on the same inputs: wall time per iteration and
agreement of the test-loss history.  Default: 10 000 x 40 000, 1 %-dense, k = 16, 10 % of ALL entries held out."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import oracle as O
from oracle.oracle import Csc
from rcppml_amd import _abi, data
m, n, k, iters = 10000, 40000, 16, 4
A, _, _ = data.simulate_nmf_sparse(m, n, k, 0.0115, seed=123, device=torch.device("cuda", 0))
W0, H0 = data.init_factors(42, k, m, n, np.float64)
p, i, x = A.p.astype(np.int32), A.i.astype(np.int32), A.x.astype(np.float64)
Ao = Csc((m, n), A.p, A.i, A.x)
try:
    O.build(native=True); native = True
except Exception:
    native = False
for mz in (0, 1):
    for prec, dtype in ((_abi.F32, np.float32), (_abi.F64, np.float64)):
        W, H = W0.copy(), H0.copy()
        t0 = time.perf_counter()
        res = _abi.nmf_cv(p, i, x, m, n, k, W, H, entry="ex", max_iter=iters, tol=0.0, solver_mode=1, holdout_fraction=0.1,
                          cv_seed=5, mask_zeros=mz, precision=prec, cv_patience=0, sort_model=0)
        tg = time.perf_counter() - t0
        assert res["status"] == 0, res.get("error")
        t0 = time.perf_counter()
        ref = O.nmf_fit_cv(Ao, W0, H0, dtype, max_iter=iters, tol=0.0, solver_mode=1, holdout_fraction=0.1, cv_seed=5,
                           mask_zeros=bool(mz), cv_patience=0, threads=0, native=native)
        tc = time.perf_counter() - t0
        print("mask_zeros=%d %s: GPU %.3fs (%d it)  CPU oracle %d threads %.2fs  speedup %.0fx  test loss gpu %.6g ref %.6g rel %.1e" % (
            mz, "fp32" if prec == _abi.F32 else "fp64", tg, res["iter"], O.num_threads(), tc, tc / tg, res["test_loss"], ref.test_loss,
            abs(res["test_loss"] - ref.test_loss) / abs(ref.test_loss)))
