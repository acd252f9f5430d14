#!/usr/bin/env python3
"""Where the one-kernel fit stops paying: steady-state microseconds per ALS iteration of rcppml_hip_als_small_fit (40 iterations continuing a
10-iteration fit, fp32) against hipGraph replays of the multi-launch iteration on the same input, over a grid of sizes and ranks, CD and
Cholesky.  The rule of rcppml_hip_als_small_eligible (k <= 16, m + n <= 3072, nnz <= 2^17) is read off this table (profiles/r06_small_threshold.txt)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcppml_amd import _abi, als, data
from tests.util import lowrank_csc

print("%6s %6s %3s %8s %10s %-5s %9s %9s %6s  %s" % ("m", "n", "k", "nnz", "(m+n)k^2", "solv", "one us", "multi us", "ratio", "plugin takes"))
for (m, n) in ((100, 600), (183, 1183), (400, 2400), (800, 3200), (1500, 6000)):
    Ao = lowrank_csc(m, n, 6, 0.08, seed=m)
    A = data.CSC((m, n), Ao.p, Ao.i, Ao.x)
    At = A.transpose()
    for k in (4, 10, 16, 24, 32):
        W0, H0 = data.init_factors(3, k, m, n, np.float32)
        for solver in (0, 1):
            ops = als.HipOps(0, "f32")
            a, at = ops.upload_csc(A), ops.upload_csc(At)
            tr = ops.sumsq(a["x"])
            W, H, d = ops.to_device(W0), ops.to_device(H0), ops.zeros((k,)) + 1
            res = torch.zeros(8, dtype=torch.float64, device="cuda")
            ops.ctx.als_small_fit(ops.dt, a, at, m, n, k, W, H, d, tr, solver_mode=solver, max_iter=10, tol=0.0, iter0=0, result8=res)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                ops.ctx.als_small_fit(ops.dt, a, at, m, n, k, W, H, d, tr, solver_mode=solver, max_iter=40, tol=0.0, iter0=10, result8=res)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / 40 * 1e6)
            assert float(res[4].item()) == 1.0
            side = torch.cuda.Stream(device=0)
            with torch.cuda.stream(side):
                ops2 = als.HipOps(0, "f32")
                st = als.ShardedALS(ops2, als.Comm(None), A, At, W0, H0, als.AlsConfig(k=k, max_iter=100, tol=0.0, solver_mode=solver))
                for _ in range(4):
                    st.step()
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    st.step()
                torch.cuda.synchronize()
                bm = 1e9
                for _ in range(3):
                    t0 = time.perf_counter()
                    for _ in range(40):
                        g.replay()
                    torch.cuda.synchronize()
                    bm = min(bm, (time.perf_counter() - t0) / 40 * 1e6)
            print("%6d %6d %3d %8d %10.3g %-5s %9.1f %9.1f %6.2f  %s" % (m, n, k, A.nnz, (m + n) * k * k, "cd" if solver == 0 else "chol", best, bm, bm / best,
                                                                       "one kernel" if _abi.small_eligible(m, n, A.nnz, k) else "multi-launch"))
