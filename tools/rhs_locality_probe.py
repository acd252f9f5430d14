#!/usr/bin/env python3
"""Probe: is the rhs gather bound by L2 misses?  Same nonzero count and column structure, but row indices folded
into a smaller range so the gathered factor fits one XCD's L2 (timing only; results are meaningless)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcppml_amd import als, data
m, n, k = 20000, 100000, 64
A, _, _ = data.simulate_nmf_sparse(m, n, k, 0.0115, seed=123, device=torch.device("cuda", 0))
ops = als.HipOps(0, "f32")
for rows in (20000, 8000, 4000, 1000, 250):
    F = ops.to_device(np.random.default_rng(0).uniform(size=(rows, k)).astype(np.float32))
    csc = ops.upload_csc(A)
    csc["i"] = (csc["i"] % rows).contiguous()
    B = ops.rhs(csc, F)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        ops.rhs(csc, F, out=B)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print("F rows %6d (%.2f MB): %.3f ms  gather %.2f TB/s" % (rows, rows * k * 4 / 1e6, ms, A.nnz * k * 4 / ms / 1e9))
