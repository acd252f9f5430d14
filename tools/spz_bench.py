#!/usr/bin/env python3
"""tools/spz_bench.py -- decode time of the bundled pbmc3k.spz (2.28 M nonzeros, 22 rANS streams) on the GPU
(rcppml_hip_spz_decode: host parse + upload of the 2.1 MB file + three kernels, synchronised) next to the CPU oracle's
serial restatement of the reference decoder.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from oracle import oracle as O  # noqa: E402
from rcppml_amd import _abi  # noqa: E402

buf = np.fromfile(os.path.join(ROOT, "tests", "golden", "pbmc3k.spz"), np.uint8)
st, m, n, nnz, vt = _abi.spz_info(buf)
ctx = _abi.Context(0)
dp = torch.zeros(n + 1, dtype=torch.int32, device="cuda")
di = torch.zeros(nnz, dtype=torch.int32, device="cuda")
dx = torch.zeros(nnz, dtype=torch.float64, device="cuda")
for _ in range(3):
    ctx.spz_decode(buf, dp, di, dx)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    t0 = time.perf_counter()
    ctx.spz_decode(buf, dp, di, dx)
    ts.append(time.perf_counter() - t0)
O.spz_decode(buf)
tc = []
for _ in range(3):
    t0 = time.perf_counter()
    O.spz_decode(buf)
    tc.append(time.perf_counter() - t0)
gpu, cpu = float(np.median(ts)), float(np.median(tc))
print(json.dumps(dict(file="pbmc3k.spz", bytes=int(buf.size), m=m, n=n, nnz=nnz, streams=22, gpu_decode_ms=gpu * 1e3,
                      gpu_symbols_per_s=2 * nnz / gpu, cpu_oracle_1thread_ms=cpu * 1e3, decoded_bytes=int(nnz * 12 + 4 * (n + 1)),
                      gpu_decoded_GBps=(nnz * 12) / gpu / 1e9)))
