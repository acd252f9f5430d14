import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import oracle as O
from rcppml_amd import _abi
from tests.util import random_csc
ctx = _abi.Context(0)
A = random_csc(700, 1500, 0.012, seed=64)
k = 64
F = np.random.default_rng(k + A.rows).standard_normal((A.rows, k)).astype(np.float32)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
dp, di, dx = dev(A.p), dev(A.i), dev(A.values(np.float32))
plan = ctx.rhs_plan(_abi.F32, dp, di, dx, A.cols, A.rows, k, 0, 0)
print(plan.info())
dB = torch.full((A.cols, k), 7.0, dtype=torch.float32, device="cuda")
ctx.rhs_planned(plan, dev(F), dB)
B = dB.cpu().numpy(); R = O.rhs(A, F, np.float32)
err = np.abs(B - R).max(axis=1)
bad = np.nonzero(err > 1e-4)[0]
print("bad columns", len(bad), "of", A.cols, bad[:40])
if len(bad):
    j = bad[0]
    print("col", j, "nnz", A.p[j+1]-A.p[j], "rows", A.i[A.p[j]:A.p[j+1]])
    print("B", B[j][:8], "R", R[j][:8])
    # which single-tile contributions are missing? tiles of 256 rows
    S = plan.info()["slots"]
    tot_slot = 0; tot_spill = 0
    for t in range(3):
        sel = (A.i[A.p[j]:A.p[j+1]] // 256) == t
        rows = A.i[A.p[j]:A.p[j+1]][sel]; vals = A.x[A.p[j]:A.p[j+1]][sel]
        tot_slot = tot_slot + (vals[:S, None] * F[rows[:S]]).sum(axis=0)
        tot_spill = tot_spill + (vals[S:, None] * F[rows[S:]]).sum(axis=0)
    print(" slots only", tot_slot[:4], " spill only", tot_spill[:4])
    for t in range(3):
        sel = (A.i[A.p[j]:A.p[j+1]] // 256) == t
        rows = A.i[A.p[j]:A.p[j+1]][sel]; vals = A.x[A.p[j]:A.p[j+1]][sel]
        part = (vals[:, None] * F[rows]).sum(axis=0)
        print(" tile", t, "n", sel.sum(), "partial", part[:4])
