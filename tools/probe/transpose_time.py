"""Device-side CSC transpose (rcppml_hip_transpose_csc): ms per call for a few shapes, the tall ones (> 32 768 rows) through the radix passes."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from rcppml_amd import _abi
ctx = _abi.Context(0)
for m, n, nnz in ((20000, 100000, 20_000_000), (100000, 20000, 20_000_000), (1_000_000, 2000, 20_000_000), (2_000_000, 500, 5_000_000)):
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    cols = torch.randint(0, n, (nnz,), device="cuda", generator=g, dtype=torch.int64).sort().values
    rows = torch.randint(0, m, (nnz,), device="cuda", generator=g, dtype=torch.int64)
    key = (cols * m + rows).unique()                      # sorted (col, row) pairs without duplicates
    cols, rows = key // m, (key % m).to(torch.int32)
    nz = int(key.numel())
    p = torch.zeros(n + 1, dtype=torch.int32, device="cuda")
    p[1:] = torch.cumsum(torch.bincount(cols, minlength=n), 0).to(torch.int32)
    x = torch.rand(nz, device="cuda", dtype=torch.float32)
    tp = torch.empty(m + 1, dtype=torch.int32, device="cuda"); ti = torch.empty(nz, dtype=torch.int32, device="cuda"); tx = torch.empty(nz, device="cuda")
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.transpose_csc(_abi.F32, m, n, p, rows, x, tp, ti, tx)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    # check against a torch sort by (row, col)
    order = torch.argsort(rows.to(torch.int64) * n + cols, stable=True)
    ok = bool(torch.equal(ti, cols[order].to(torch.int32))) and bool(torch.equal(tx, x[order]))
    print("%8d x %7d  nnz %9d  transpose %.2f ms  (%s path)  correct %s" % (m, n, nz, dt, "radix" if m > 32768 else "LDS-counter", ok), flush=True)
