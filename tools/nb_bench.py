#!/usr/bin/env python3
"""Timing of BASELINE configs[4] (C5): loss='nb' IRLS NMF on 10 000 x 200 000 counts, k = 32 -- kernel-level
(one IRLS half-update per side, the NB size update and the NB loss) through the device-level C ABI."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcppml_amd import als, data, _abi
m, n, k = 10000, int(sys.argv[1]) if len(sys.argv) > 1 else 200000, int(os.environ.get("K", "32"))
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
t0 = time.time()
A, _, _ = data.simulate_nb_counts(m, n, k, density=0.02, size=5.0, seed=123)
At = A.transpose()
print("data: %d x %d nnz %d (%.1fs)" % (m, n, A.nnz, time.time() - t0))
nd = np.float32 if dtype == "f32" else np.float64
W0, H0 = data.init_factors(42, k, m, n, nd)
ops = als.HipOps(0, dtype)
W, H = ops.to_device(W0), ops.to_device(H0)
Ad, Atd = ops.upload_csc(A), ops.upload_csc(At)
theta_row = torch.full((m,), 10.0, dtype=W.dtype, device="cuda")


def timeit(name, fn, reps=3):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    print("%-12s %.3f ms" % (name, s.elapsed_time(e) / reps))


G_h = ops.gram(W, 1e-15, 0.0)
for cdm, irm in ((1, 1), (100, 1), (1, 5)):
    timeit("irls_H cd_maxit=%d irls=%d" % (cdm, irm), lambda: ops.ctx.solve_irls_nb(ops.dt, Ad["p"], Ad["i"], Ad["x"], n, W, G_h, H, k, 0.0, 0.0, 1, cdm, irm, 1e-4, theta_row, None))
timeit("irls_H", lambda: ops.ctx.solve_irls_nb(ops.dt, Ad["p"], Ad["i"], Ad["x"], n, W, G_h, H, k, 0.0, 0.0, 1, 100, 5, 1e-4,
                                               theta_row, None))
G_w = ops.gram(H, 1e-15, 0.0)
timeit("irls_W", lambda: ops.ctx.solve_irls_nb(ops.dt, Atd["p"], Atd["i"], Atd["x"], m, H, G_w, W, k, 0.0, 0.0, 1, 100, 5, 1e-4,
                                               None, theta_row))
d = torch.ones((k,), dtype=W.dtype, device="cuda")
nb_size = torch.full((m,), 10.0, dtype=W.dtype, device="cuda")
timeit("nb_size", lambda: ops.ctx.nb_size_update(ops.dt, Atd["p"], Atd["i"], Atd["x"], m, W, d, H, n, k, 0.01, 1e6, nb_size))
out = torch.zeros((1,), dtype=torch.float64, device="cuda")
timeit("nb_loss", lambda: ops.ctx.nb_loss(ops.dt, Ad["p"], Ad["i"], Ad["x"], n, W, d, H, theta_row, k, out))
