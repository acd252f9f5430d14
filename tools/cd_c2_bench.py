#!/usr/bin/env python3
"""CD solve on the bench workload's steady state (C2, iteration `warm` of the ALS): the same (G, B, X, order) inputs through
every kernel variant and LMF geometry, ms per solve + work counters.  Run on the GPU box:  python tools/cd_c2_bench.py [iters]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcppml_amd import als, data, _abi

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m, n, k = 20000, 100000, 64
A, _, _ = data.simulate_nmf_sparse(m, n, k, 0.0115, seed=123, device=torch.device("cuda", 0))
At = A.transpose()
W0, H0 = data.init_factors(42, k, m, n, np.float32)
ops = als.HipOps(0, "f32")
ctx = ops.ctx
W, H = ops.to_device(W0), ops.to_device(H0)
Ad, Atd = ops.upload_csc(A), ops.upload_csc(At)
sums, d = ops.empty((k,)), ops.empty((k,))
sw = {"H": torch.zeros(n, dtype=torch.int32, device="cuda"), "W": torch.zeros(m, dtype=torch.int32, device="cuda")}
order = {"H": torch.zeros(n, dtype=torch.int32, device="cuda"), "W": torch.zeros(m, dtype=torch.int32, device="cuda")}
snap = {}
for it in range(iters + 1):
    for side in ("H", "W"):
        F, X, csc = (W, H, Ad) if side == "H" else (H, W, Atd)
        nc = X.shape[0]
        G = ops.gram(F, 1e-15, 0.0)
        B = ops.rhs(csc, F)
        if it > 0:
            ctx.order_columns(sw[side], nc, order[side])
        if it == iters:
            snap[side] = (G.clone(), B.clone(), X.clone(), order[side].clone())
        ctx.solve_cd(ops.dt, G, B, X, k, nc, warm=int(it > 0), maxit=100, tol=1e-8, sweeps_out=sw[side],
                     col_order=order[side] if it > 0 else None)
        ops.row_norms(X, 0, out=sums)
        ops.apply_scaling(X, sums, 0, d)
    if it == iters:
        break

def bench(side, label, variant, lg=0, wps=0, count=0, use_order=True):
    G, B, X0, od = snap[side]
    nc = X0.shape[0]
    ctx.set_option(_abi.OPT_CD_LMF_LANE_GROUPS, lg)
    ctx.set_option(_abi.OPT_CD_LMF_WAVES_PER_SIMD, wps)
    ctx.set_option(_abi.OPT_CD_COUNT_NOOP, count)
    X = X0.clone()
    s_out = torch.zeros(nc, dtype=torch.int32, device="cuda")
    def run():
        X.copy_(X0)
        ctx.solve_cd(ops.dt, G, B, X, k, nc, warm=1, maxit=100, tol=1e-8, variant=variant, sweeps_out=s_out,
                     col_order=od if use_order else None)
    run(); torch.cuda.synchronize()
    ctx.stats(reset=True)
    ts = []
    for _ in range(5):
        X.copy_(X0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ctx.solve_cd(ops.dt, G, B, X, k, nc, warm=1, maxit=100, tol=1e-8, variant=variant, sweeps_out=s_out,
                     col_order=od if use_order else None)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    st = ctx.stats(reset=True)
    cs = st["cd_column_sweeps"] / 5
    ms = float(np.median(ts))
    tf = 2 * k * k * cs / (ms * 1e-3) / 1e12
    extra = ""
    if st["cd_slot_sweeps"]:
        extra = " idle %.3f" % (1 - st["cd_column_sweeps"] / st["cd_slot_sweeps"])
    if count and st["cd_slot_sweeps"]:
        cw = 64 // (lg if lg else 1)
        steps = st["cd_slot_sweeps"] / cw * k
        extra += " noop-steps %.4f" % (st["cd_noop_steps"] / steps)
    print("%s %-26s %.3f ms  (min %.3f) col-sweeps %.3e  %.1f TFLOP/s = %.3f of f32 MFMA peak%s  checksum %.6e"
          % (side, label, ms, min(ts), cs, tf, tf / 157.3, extra, float(X.double().sum().item())), flush=True)
    return X.clone(), s_out.clone()

for side in ("H", "W"):
    ref, sref = bench(side, "mfma32 (r02 default H)", _abi.CD_MFMA)
    x16, _ = bench(side, "mfma16 (r02 default W)", _abi.CD_MFMA16)
    for lg in (1, 2, 4):
        for wps in (1, 2, 3, 4, 5, 6, 8):
            if lg == 1 and wps > 2:
                continue
            x, s = bench(side, "lmf lg=%d wps=%d" % (lg, wps), _abi.CD_LMF, lg, wps)
            dx = (x - ref).abs().max().item()
            ds = (s - sref).abs()
            print("      vs mfma32: max|dX| %.2e, columns with different sweep count %d (max diff %d)" % (dx, int((ds > 0).sum().item()), int(ds.max().item())))
    bench(side, "lmf auto", _abi.CD_AUTO)
    bench(side, "lmf auto, natural order", _abi.CD_AUTO, use_order=False)
    bench(side, "lmf lg=4 wps=1 counting", _abi.CD_LMF, 4, 1, count=1)
    bench(side, "lmf lg=1 wps=1 counting", _abi.CD_LMF, 1, 1, count=1)
