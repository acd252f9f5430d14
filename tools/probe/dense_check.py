#!/usr/bin/env python3
"""Large-size check of the hand-written dense right-hand sides against torch.matmul (fp64 accumulate on a sample)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from rcppml_amd import _abi
ctx = _abi.Context(0)
for (m, n, k) in ((8192, 32768, 64), (5000, 9001, 40), (4096, 4096, 128)):
    g = torch.Generator(device="cuda").manual_seed(1)
    At = torch.rand((n, m), device="cuda", generator=g)            # memory = column-major m x n
    F = torch.rand((m, k), device="cuda", generator=g); B = torch.zeros((n, k), device="cuda")
    F2 = torch.rand((n, k), device="cuda", generator=g); B2 = torch.zeros((m, k), device="cuda")
    ctx.rhs_dense(_abi.F32, At, m, n, 0, F, k, B); ctx.rhs_dense(_abi.F32, At, m, n, 1, F2, k, B2); ctx.sync()
    ref = At.double() @ F.double()                                   # (n, k): B[:, j] = sum_i A[i, j] F[:, i]
    ref2 = At.double().T @ F2.double()                               # (m, k)
    print(m, n, k, "fwd rel err %.2e" % ((B.double() - ref).abs().max() / ref.abs().max()).item(),
          "bwd rel err %.2e" % ((B2.double() - ref2).abs().max() / ref2.abs().max()).item())
    Ad, Fd, F2d = At.double(), F.double(), F2.double()
    Bd = torch.zeros((n, k), device="cuda", dtype=torch.float64); B2d = torch.zeros((m, k), device="cuda", dtype=torch.float64)
    ctx.rhs_dense(_abi.F64, Ad, m, n, 0, Fd, k, Bd); ctx.rhs_dense(_abi.F64, Ad, m, n, 1, F2d, k, B2d); ctx.sync()
    print("   f64: fwd rel err %.2e" % ((Bd - ref).abs().max() / ref.abs().max()).item(),
          "bwd rel err %.2e" % ((B2d - ref2).abs().max() / ref2.abs().max()).item())
    import time
    for dt, a_, f_, b_, f2_, b2_ in ((_abi.F32, At, F, B, F2, B2), (_abi.F64, Ad, Fd, Bd, F2d, B2d)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            ctx.rhs_dense(dt, a_, m, n, 0, f_, k, b_); ctx.rhs_dense(dt, a_, m, n, 1, f2_, k, b2_)
        ctx.sync(); torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 5
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            r1 = a_ @ f_; r2 = a_.T @ f2_                               # library GEMMs (rocBLAS / hipBLASLt) on the same products
        torch.cuda.synchronize()
        tl = (time.perf_counter() - t0) / 5
        print("   %s pair %.3f ms = %.1f TFLOP/s   (torch.matmul pair %.3f ms)" % ("f32" if dt == _abi.F32 else "f64", t * 1e3, 4.0 * m * n * k / t / 1e12, tl * 1e3))
