#!/usr/bin/env python3
"""Probe: does processing the nonzeros in row-slab phases (gathered slab of F resident in one XCD's L2) pay?
Per phase a sub-CSC holding only the nonzeros of that row slab is built with torch; the phases are timed back to back
(separate output buffers: the accumulate cost is not included)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rcppml_amd import als, data
m, n, k = 20000, 100000, 64
A, _, _ = data.simulate_nmf_sparse(m, n, k, 0.0115, seed=123, device=torch.device("cuda", 0))
At = A.transpose()
W0, H0 = data.init_factors(42, k, m, n, np.float32)
ops = als.HipOps(0, "f32")
W, H = ops.to_device(W0), ops.to_device(H0)
for name, M, F in (("H", A, W), ("W", At, H)):
    csc = ops.upload_csc(M)
    rows, cols = M.rows, M.cols
    colid = torch.repeat_interleave(torch.arange(cols, device="cuda"), (csc["p"][1:] - csc["p"][:-1]).long())
    for P in (1, 2, 3, 4, 8, 13, 16):
        slab = (rows + P - 1) // P
        subs = []
        for p in range(P):
            sel = (csc["i"] >= p * slab) & (csc["i"] < (p + 1) * slab)
            cnt = torch.bincount(colid[sel], minlength=cols)
            pp = torch.zeros(cols + 1, dtype=torch.int32, device="cuda")
            pp[1:] = torch.cumsum(cnt, 0).int()
            subs.append(dict(p=pp, i=csc["i"][sel].contiguous(), x=csc["x"][sel].contiguous(), cols=cols, rows=rows, nnz=int(sel.sum())))
        Bs = [ops.rhs(s, F) for s in subs]
        torch.cuda.synchronize()
        s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        for _ in range(10):
            for s, B in zip(subs, Bs):
                ops.rhs(s, F, out=B)
        e0.record(); torch.cuda.synchronize()
        print("rhs_%s P=%2d: %.3f ms" % (name, P, s0.elapsed_time(e0) / 10))
