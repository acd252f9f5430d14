#!/usr/bin/env python3
"""Does the 22-wavefront rANS launch run at a reduced engine clock?  Decode pbmc3k.spz alone, then again while a side
stream keeps the chip busy with large GEMMs."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from rcppml_amd import _abi
buf = np.fromfile(os.path.join(ROOT, "tests", "golden", "pbmc3k.spz"), np.uint8)
st, m, n, nnz, vt = _abi.spz_info(buf)
ctx = _abi.Context(0)
dp = torch.zeros(n + 1, dtype=torch.int32, device="cuda"); di = torch.zeros(nnz, dtype=torch.int32, device="cuda"); dx = torch.zeros(nnz, dtype=torch.float64, device="cuda")
def run(k=5):
    ts = []
    for _ in range(k):
        t0 = time.perf_counter(); ctx.spz_decode(buf, dp, di, dx); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e3
run(2)
print("alone: %.1f ms" % run())
side = torch.cuda.Stream()
a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
with torch.cuda.stream(side):
    for _ in range(400):
        a @ a
print("with GEMMs on a side stream: %.1f ms" % run())
torch.cuda.synchronize()
print("alone again: %.1f ms" % run())
