"""fp64 Gram: the single-pass k <= 64 kernel (all sixteen tiles per block) against the (nblk, 4)-grid kernel of the library build before it
(tools/probe/old_lib, temporary): bitwise comparison of G on several shapes, and of the fused tail ops; run once per library through
RCPPML_GPU_LIB_PATH, results compared through files."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from rcppml_amd import _abi
tag = sys.argv[1]
ctx = _abi.Context(0)
out = {}
for k, c in ((64, 100000), (64, 20000), (64, 777), (64, 3), (50, 30011), (60, 5000), (49, 123)):
    rs = np.random.default_rng(k + c)
    X = torch.from_numpy((rs.uniform(size=(c, k)) * (rs.uniform(size=(c, k)) < 0.7))).cuda()
    G = torch.empty((k, k), dtype=torch.float64, device="cuda")
    ctx.gram(_abi.F64, X, k, c, 1e-15, 0.25, G)
    ctx.sync()
    out["G_%d_%d" % (k, c)] = G.cpu().numpy()
    import time
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        ctx.gram(_abi.F64, X, k, c, 1e-15, 0.0, G)
    torch.cuda.synchronize()
    print(tag, "gram f64 k=%d c=%d: %.1f us per call" % (k, c, (time.perf_counter() - t0) / 20 * 1e6), flush=True)
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "gpurun_out", "f64_gram_%s.npz" % tag), **out)
