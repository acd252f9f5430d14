"""debug helper: window rhs plan on small matrices vs numpy, prints where the result differs"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rcppml_amd import _abi
from tests.util import random_csc
ctx = _abi.Context(0)
def run(rows, cols, dens, k, P, code, dtype=np.float32, seed=1):
    A = random_csc(rows, cols, dens, seed=seed)
    dt = _abi.F32 if dtype == np.float32 else _abi.F64
    F = np.random.default_rng(3).standard_normal((rows, k)).astype(dtype)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    dp, di, dx = dev(A.p), dev(A.i), dev(A.values(dtype))
    plan = ctx.rhs_plan(dt, dp, di, dx, cols, rows, k, P, code)
    if plan is None:
        print("no plan", rows, cols, dens, k, P, code); return
    dB = torch.full((cols, k), 7.0, dtype=torch.float32 if dtype == np.float32 else torch.float64, device="cuda")
    ctx.rhs_planned(plan, dev(F), dB)
    B = dB.cpu().numpy()
    import scipy.sparse as sp
    M = sp.csc_matrix((A.values(np.float64), A.i, A.p), shape=(rows, cols))
    ref = (M.T @ F.astype(np.float64))
    err = np.abs(B - ref).max(axis=1) / (np.abs(ref).max() + 1e-30)
    bad = np.nonzero(~(err < 1e-4))[0]
    print(f"rows {rows} cols {cols} dens {dens} k {k} P {P} code {code}: {plan.info()} max err {np.nanmax(err):.2e} nan {np.isnan(B).sum()} bad cols {len(bad)} first {bad[:10]}")
for args in [(300, 200, 0.02, 64, 1, 108), (300, 200, 0.02, 64, 1, 104), (300, 200, 0.02, 64, 1, 107), (3000, 900, 0.01, 64, 1, 108), (3000, 900, 0.01, 64, 2, 108),
             (3000, 900, 0.01, 64, 2, 107), (3000, 5000, 0.01, 64, 3, 0), (3000, 900, 0.01, 128, 1, 0), (20000, 9000, 0.01, 64, 2, 107)]:
    run(*args)
run(3000, 900, 0.01, 64, 2, 0, np.float64)
run(3000, 900, 0.01, 32, 2, 0, np.float64)
