#!/usr/bin/env python3
"""Staged, unbuffered walk through BASELINE configs[3] at full extent (30 000 x 1 300 000, 3 %, k = 128): where does the time go?
usage: python -u tools/c4full_probe.py [shards_to_generate] [stages: g,p1,p8,dev]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcppml_amd import _abi, data

M, N, K = 30000, 1300000, 128
nsh_gen = int(sys.argv[1]) if len(sys.argv) > 1 else 8
stages = (sys.argv[2] if len(sys.argv) > 2 else "g,p1").split(",")
T0 = time.perf_counter()


def say(msg):
    print("[%7.1f s] %s" % (time.perf_counter() - T0, msg), flush=True)


dev = torch.device("cuda", 0)
nsh = N // 8
parts = []
for r in range(nsh_gen):
    t = time.perf_counter()
    a = data.simulate_nmf_sparse(M, nsh, K, 0.031, seed=11, device=dev, col_offset=r * nsh, ncol_total=N)[0]
    parts.append(a)
    say("shard %d: nnz %d in %.1f s" % (r, a.nnz, time.perf_counter() - t))
n = nsh * nsh_gen
off = np.cumsum([0] + [a.nnz for a in parts])
A = data.CSC((M, n), np.concatenate([parts[0].p[:1].astype(np.int64)] + [a.p[1:].astype(np.int64) + off[r] for r, a in enumerate(parts)]),
             np.concatenate([a.i for a in parts]), np.concatenate([a.x.astype(np.float32).astype(np.float64) for a in parts]))
del parts
torch.cuda.empty_cache()
say("matrix %d x %d nnz %d assembled" % (M, n, A.nnz))
W0, H0 = data.init_factors(42, K, M, n, np.float32)
W0, H0 = W0.astype(np.float64), H0.astype(np.float64)
say("factors drawn")


def plugin(ndev, w_solve=None, iters=2, precision=None):
    W, H = W0.copy(), H0.copy()
    env = dict(RCPPML_GPU_DEVICES=str(ndev) if ndev > 1 else None, RCPPML_GPU_DEVICES_SHARE="1" if ndev > 1 else None, RCPPML_GPU_W_SOLVE=w_solve)
    for k_, v in env.items():
        if v is None:
            os.environ.pop(k_, None)
        else:
            os.environ[k_] = v
    t = time.perf_counter()
    res = _abi.nmf_unified(A.p, A.i, A.x, A.rows, A.cols, K, W, H, entry="ex", precision=_abi.F32 if precision is None else precision, max_iter=iters, tol=0.0, solver_mode=0,
                           sort_model=0, want_history=True, verbose=2)
    say("plugin ndev=%d w_solve=%s: status %d %s, %.1f s, loss history %s" % (ndev, w_solve, res["status"], res.get("error", ""), time.perf_counter() - t,
                                                                             res.get("loss_history")))
    return res, W, H


if "p1" in stages:
    plugin(1)
if "p1f64" in stages:          # parity mode at full extent: 1024-byte rows in the window kernel, eight row tiles in the fp64 MFMA solve
    r64, W64, H64 = plugin(1, precision=_abi.F64)
    r32, W32, H32 = plugin(1)
    say("fp64 vs fp32 at full extent: loss %s vs %s, max |dW| / max W %.2e, max |dH| / max H %.2e" % (
        r64.get("loss_history"), r32.get("loss_history"), np.abs(W64 - W32).max() / W64.max(), np.abs(H64 - H32).max() / H64.max()))
if "p8" in stages:
    plugin(8)
    plugin(8, "block")
if "plan" in stages:
    from rcppml_amd import als
    ops = als.HipOps(0, "f32")
    a = ops.upload_csc(A)
    ops.sync(); say("CSC uploaded through torch")
    for deferred in (True, False):
        t = time.perf_counter()
        try:
            if deferred:
                pl = ops.ctx.rhs_plan_indices(ops.dt, a["p"], a["i"], a["cols"], a["rows"], K)
            else:
                pl = ops.ctx.rhs_plan(ops.dt, a["p"], a["i"], a["x"], a["cols"], a["rows"], K)
            ops.sync()
            say("plan of A (deferred=%s): %s in %.2f s" % (deferred, pl.info() if pl is not None else None, time.perf_counter() - t))
        except Exception as e:
            say("plan of A (deferred=%s) failed: %r" % (deferred, e))
        pl = None
    t = time.perf_counter()
    at = ops.transpose_csc(a)
    ops.sync(); say("device transpose %.2f s" % (time.perf_counter() - t))
    for deferred in (True, False):
        t = time.perf_counter()
        try:
            if deferred:
                pl = ops.ctx.rhs_plan_indices(ops.dt, at["p"], at["i"], at["cols"], at["rows"], K)
            else:
                pl = ops.ctx.rhs_plan(ops.dt, at["p"], at["i"], at["x"], at["cols"], at["rows"], K)
            ops.sync()
            say("plan of A^T (deferred=%s): %s in %.2f s" % (deferred, pl.info() if pl is not None else None, time.perf_counter() - t))
        except Exception as e:
            say("plan of A^T (deferred=%s) failed: %r" % (deferred, e))
        pl = None
    del a, at, ops
    torch.cuda.empty_cache()
if "dev" in stages:
    from rcppml_amd import als
    ops = als.HipOps(0, "f32")
    t = time.perf_counter()
    cfg = als.AlsConfig(k=K, max_iter=3, tol=0.0)
    st = als.ShardedALS(ops, als.Comm(None), A, None, W0, H0, cfg)
    ops.sync()
    say("device loop set-up %.1f s" % (time.perf_counter() - t))
    for it in range(3):
        t = time.perf_counter()
        loss = float(st.step()[0].item())
        say("device loop iteration %d: %.3f s loss %.9g" % (it, time.perf_counter() - t, loss))
say("done")
