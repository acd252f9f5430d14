"""Times the LDS row-tiled right-hand side against the gather kernel on the C2 shape (20 000 x 100 000, 1 %, k = 64):
H side (F = W_T, 5.1 MB) and W side (F = H, 25.6 MB).  Usage: python tools/rhs_tiled_bench.py [P_w] [S]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rcppml_amd import _abi  # noqa: E402
from rcppml_amd.data import simulate_nmf_sparse  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def main():
    m, n, k = int(os.environ.get("M", "20000")), int(os.environ.get("N", "100000")), 64
    dens = float(os.environ.get("DENS", "0.01"))
    dtype = np.float32 if os.environ.get("DT", "f32") == "f32" else np.float64
    k = int(os.environ.get("K", "64"))
    A, _, _ = simulate_nmf_sparse(m, n, k, dens, seed=123, device=torch.device("cuda", 0))
    ctx = _abi.Context(0)
    dt = _abi.F32 if dtype == np.float32 else _abi.F64
    tt = torch.float32 if dtype == np.float32 else torch.float64
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    Ap, Ai, Ax = dev(A.p.astype(np.int32)), dev(A.i.astype(np.int32)), dev(A.x.astype(dtype))
    Tp = torch.empty(m + 1, dtype=torch.int32, device="cuda")
    Ti = torch.empty(A.nnz, dtype=torch.int32, device="cuda")
    Tx = torch.empty(A.nnz, dtype=tt, device="cuda")
    ctx.transpose_csc(dt, m, n, Ap, Ai, Ax, Tp, Ti, Tx)
    rng = np.random.default_rng(0)
    W = dev(rng.uniform(size=(m, k)).astype(dtype))
    H = dev(rng.uniform(size=(n, k)).astype(dtype))
    Bh = torch.empty((n, k), dtype=tt, device="cuda")
    Bw = torch.empty((m, k), dtype=tt, device="cuda")
    Bh2, Bw2 = torch.empty_like(Bh), torch.empty_like(Bw)
    t_h = timeit(lambda: ctx.rhs(dt, Ap, Ai, Ax, n, W, k, Bh))
    t_w = timeit(lambda: ctx.rhs(dt, Tp, Ti, Tx, m, H, k, Bw))
    print("gather kernel: rhs_H %.3f ms  rhs_W %.3f ms" % (t_h, t_w))
    args = [int(a) for a in sys.argv[1:]]
    Pw = args[0] if len(args) > 0 else 0
    S = args[1] if len(args) > 1 else 0
    Ph = args[2] if len(args) > 2 else 0
    t0 = time.time()
    ph = ctx.rhs_plan(dt, Ap, Ai, Ax, n, m, k, Ph, S)
    pw = ctx.rhs_plan(dt, Tp, Ti, Tx, m, n, k, Pw, S)
    torch.cuda.synchronize()
    print("plans built in %.1f ms" % ((time.time() - t0) * 1e3))
    for name, plan, F, B, Bref in (("H", ph, W, Bh2, Bh), ("W", pw, H, Bw2, Bw)):
        if plan is None:
            print("side", name, "not eligible")
            continue
        print("side", name, plan.info())
        t = timeit(lambda: ctx.rhs_planned(plan, F, B))
        err = float((B - Bref).abs().max() / Bref.abs().max())
        print("tiled kernel: rhs_%s %.3f ms   max rel diff vs gather %.2e" % (name, t, err))


if __name__ == "__main__":
    main()
