#!/usr/bin/env python3
"""fp32 k <= 32 IRLS half-update: one column per wavefront against four (RCPPML_OPT_IRLS_COLUMNS_PER_WAVE) on the C5 shape --
results compared bit for bit, time per call for a few (cd_maxit, irls_max_iter) pairs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rcppml_amd import als, data, _abi
m, n, k = 10000, int(sys.argv[1]) if len(sys.argv) > 1 else 200000, int(os.environ.get("K", "32"))
A, _, _ = data.simulate_nb_counts(m, n, k, density=0.02, size=5.0, seed=123)
At = A.transpose()
W0, H0 = data.init_factors(42, k, m, n, np.float32)
ops = als.HipOps(0, "f32")
W, H = ops.to_device(W0), ops.to_device(H0)
Ad, Atd = ops.upload_csc(A), ops.upload_csc(At)
theta = torch.full((m,), 10.0, dtype=W.dtype, device="cuda")


def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for side, csc, F, ncols, th in (("H", Ad, W, n, (theta, None)), ("W", Atd, H, m, (None, theta))):
    G = ops.gram(F, 1e-15, 0.0)
    for cdm, irm in ((1, 1), (100, 1), (1, 5), (100, 5)):
        res = {}
        for cpw in (1, 4):
            ops.ctx.set_option(_abi.OPT_IRLS_COLUMNS_PER_WAVE, cpw)
            X = torch.zeros((ncols, k), dtype=W.dtype, device="cuda")
            t = timeit(lambda: ops.ctx.solve_irls_nb(ops.dt, csc["p"], csc["i"], csc["x"], ncols, F, G, X, k, 0.0, 0.0, 1, cdm, irm, 1e-4, th[0], th[1]))
            res[cpw] = (t, X)
        same = bool(torch.equal(res[1][1], res[4][1]))
        dev = float((res[1][1] - res[4][1]).abs().max())
        print("side %s cd_maxit %3d irls %d: 1/wave %.3f ms  4/wave %.3f ms  identical %s (max abs diff %.3g)" % (side, cdm, irm, res[1][0], res[4][0], same, dev))
ops.ctx.set_option(_abi.OPT_IRLS_COLUMNS_PER_WAVE, 0)
