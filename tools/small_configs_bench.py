#!/usr/bin/env python3
"""End-to-end plugin time on the small BASELINE configs (C1 hawaiibirds k=10 Cholesky+clip, C3 movielens k=32
L1=(0,0.1) CD) through the 73-pointer entry, next to the CPU oracle on the same inputs: wall time per ALS iteration
including upload, host transpose and download (these inputs are launch-latency bound on a GPU)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from rcppml_amd import _abi
from tests.util import load_fixture

for name, k, kw in (("hawaiibirds", 10, dict(solver_mode=1)), ("movielens", 32, dict(solver_mode=0, L1_H=0.1))):
    A = load_fixture(name)
    W0, H0 = O.init_factors(42, k, A.rows, A.cols, np.float32)
    iters = 100
    for entry in ("float", "double"):
        best = 1e9
        for rep in range(3):
            W, H = W0.astype(np.float64).copy(), H0.astype(np.float64).copy()
            t0 = time.perf_counter()
            res = _abi.nmf_unified(A.p, A.i, A.x, A.rows, A.cols, k, W, H, entry=entry, max_iter=iters, tol=0.0, **kw)
            best = min(best, time.perf_counter() - t0)
        assert res["status"] == 0 and res["iter"] == iters
        print("%-12s %-6s GPU plugin: %7.2f ms total, %6.1f us/iteration (loss %.6g)" % (name, entry, best * 1e3, best * 1e6 / iters, res["loss"]))
    for threads in (1, 0):
        t0 = time.perf_counter()
        ref = O.nmf_fit(A, W0, H0, np.float32, max_iter=iters, tol=0.0, solver_mode=kw["solver_mode"],
                        L1=(0.0, kw.get("L1_H", 0.0)), threads=threads)
        dt = time.perf_counter() - t0
        print("%-12s fp32   CPU oracle (threads=%s): %7.2f ms total, %6.1f us/iteration (loss %.6g)" % (
            name, threads or "all", dt * 1e3, dt * 1e6 / iters, ref.loss))
