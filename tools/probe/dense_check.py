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
