#!/usr/bin/env python3
"""Microbenchmark of the coordinate-descent kernel at a FIXED sweep count (tol = 0): ns per column and sweep."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rcppml_amd import als, _abi
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 64
sweeps = 20
ops = als.HipOps(0, dtype)
td = torch.float32 if dtype == "f32" else torch.float64
g = torch.Generator(device="cuda").manual_seed(1)
F = torch.rand((20000, k), device="cuda", dtype=td, generator=g)
G = ops.gram(F, 1e-15, 0.0)
for n in (20000, 100000, 400000):
    X = torch.rand((n, k), device="cuda", dtype=td, generator=g)
    B = X @ G + 0.1 * torch.randn((n, k), device="cuda", dtype=td, generator=g)
    for variant, name, lg, wps in ((_abi.CD_AUTO, "auto", 0, 0), (_abi.CD_MFMA, "mfma32", 0, 0), (_abi.CD_MFMA16, "mfma16", 0, 0), (_abi.CD_GROUP, "group", 0, 0),
                                   (_abi.CD_LMF, "lmf lg1 wps1", 1, 1), (_abi.CD_LMF, "lmf lg1 wps2", 1, 2), (_abi.CD_LMF, "lmf lg1 wps3", 1, 3),
                                   (_abi.CD_LMF, "lmf lg2 wps2", 2, 2), (_abi.CD_LMF, "lmf lg2 wps3", 2, 3), (_abi.CD_LMF, "lmf lg4 wps3", 4, 3)):
        if dtype != "f32" and variant == _abi.CD_LMF:
            continue
        ops.ctx.set_option(_abi.OPT_CD_LMF_LANE_GROUPS, lg); ops.ctx.set_option(_abi.OPT_CD_LMF_WAVES_PER_SIMD, wps)
        Xw = torch.zeros_like(X)
        def run():
            ops.ctx.solve_cd(ops.dt, G, B, Xw, k, n, 0.0, 0, 1, 0.0, 0.0, 1, sweeps, 0.0, 0.0, 0.0, variant)
        run(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            run()
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 5
        print("cd %s %s k=%d n=%d: %.3f ms for %d sweeps -> %.2f ns/col/sweep, %.1f us per sweep" % (
            dtype, name, k, n, ms, sweeps, ms * 1e6 / n / sweeps, ms * 1e3 / sweeps))
