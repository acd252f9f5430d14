#!/usr/bin/env python3
"""Microbenchmark of the SpMM-like rhs kernel on the bench workload (both half-updates)."""
import os, sys, time
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
W, H = ops.to_device(W0), ops.to_device(H0)
Ad, Atd = ops.upload_csc(A), ops.upload_csc(At)
for name, csc, F in (("H", Ad, W), ("W", Atd, H)):
    B = ops.rhs(csc, F)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        ops.rhs(csc, F, out=B)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    sv = 4 if dtype == "f32" else 8
    alg = csc["nnz"] * (4 + sv) + (csc["cols"] + 1) * 4 + k * csc["rows"] * sv + k * csc["cols"] * sv
    print("rhs_%s %s: %.3f ms  algorithmic %.1f GB/s  gather %.2f TB/s" % (name, os.environ.get("RCPPML_GPU_RHS_VARIANT", "base"), ms, alg / ms / 1e6, csc["nnz"] * k * sv / ms / 1e9))
