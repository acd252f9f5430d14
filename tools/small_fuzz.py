#!/usr/bin/env python3
"""Random-option fuzz of the one-kernel fit (rcppml_amd/csrc/kernels_small.hip.h) through the plugin boundary: for every draw the fit with
the path on, the same fit on the multi-launch loop (RCPPML_GPU_NO_SMALL=1) and the CPU oracle's, fp64; reports every case whose loss
or factors leave the bars of tests/test_gpu_small.py.  usage: python tools/small_fuzz.py [cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from rcppml_amd import _abi
from tests.util import random_csc, lowrank_csc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad, ran, skipped = [], 0, 0
for c in range(cases):
    k = int(rng.integers(1, 17))
    m, n = int(rng.integers(k + 1, 700)), int(rng.integers(k + 1, 900))
    if rng.random() < 0.5:
        A = lowrank_csc(m, n, int(rng.integers(1, 8)), float(rng.uniform(0.03, 0.25)), seed=int(rng.integers(1 << 30)))
    else:
        A = random_csc(m, n, float(rng.uniform(0.01, 0.2)), seed=int(rng.integers(1 << 30)))
    if A.nnz == 0 or not _abi.small_eligible(m, n, A.nnz, k):
        skipped += 1
        continue
    solver = int(rng.integers(0, 2))
    L1 = (float(rng.choice([0, 0, 0.01, 0.1])), float(rng.choice([0, 0, 0.02, 0.2])))
    L2 = (float(rng.choice([0, 0, 0.01])), float(rng.choice([0, 0, 0.05])))
    norm = int(rng.choice([0, 0, 0, 1]))
    cd_maxit = int(rng.choice([100, 100, 10, 3]))
    tol = float(rng.choice([0.0, 1e-4, 1e-3]))
    iters = int(rng.integers(2, 12))
    W0, H0 = O.init_factors(int(rng.integers(1, 1 << 20)), k, m, n, np.float64)
    ref = O.nmf_fit(A, W0, H0, np.float64, max_iter=iters, tol=tol, solver_mode=solver, L1=L1, L2=L2, norm_type=norm, cd_maxit=cd_maxit)
    out = {}
    for name, env in (("one", None), ("multi", "1")):
        if env is None:
            os.environ.pop("RCPPML_GPU_NO_SMALL", None)
        else:
            os.environ["RCPPML_GPU_NO_SMALL"] = env
        W, H = W0.copy(), H0.copy()
        r = _abi.nmf_unified(A.p, A.i, A.x, m, n, k, W, H, entry="ex", precision=_abi.F64, max_iter=iters, tol=tol, solver_mode=solver,
                             L1_W=L1[0], L1_H=L1[1], L2_W=L2[0], L2_H=L2[1], norm_type=norm, cd_maxit=cd_maxit, want_history=True)
        out[name] = (r, W, H)
    ran += 1
    r1, W1, H1 = out["one"]
    r2, W2, H2 = out["multi"]
    ok = r1["status"] == 0 and r2["status"] == 0
    dead = (not np.isfinite(ref.loss)) or ref.d.min() < 1e-9
    if ok and not dead:
        e_or = abs(r1["loss"] - ref.loss) / max(abs(ref.loss), 1e-300)
        e_ml = abs(r1["loss"] - r2["loss"]) / max(abs(r2["loss"]), 1e-300)
        f_or = max(np.abs(W1 - ref.W_T).max(), np.abs(H1 - ref.H).max())
        f_ml = max(np.abs(W1 - W2).max(), np.abs(H1 - H2).max())
        it_ok = r1["iter"] == ref.iter == r2["iter"] and r1["converged"] == ref.converged
        # (a convergence decision within rounding of the tolerance may flip: reported, not counted, when the losses agree)
        if e_or > 1e-6 or f_or > 1e-5 or e_ml > 1e-8 or f_ml > 1e-6 or not it_ok:
            bad.append(dict(case=c, m=m, n=n, k=k, nnz=A.nnz, solver=solver, L1=L1, L2=L2, norm=norm, cd_maxit=cd_maxit, tol=tol, iters=iters,
                            loss_vs_oracle=e_or, loss_vs_multi=e_ml, fac_vs_oracle=f_or, fac_vs_multi=f_ml, iter=(r1["iter"], r2["iter"], ref.iter)))
    elif not ok:
        bad.append(dict(case=c, status=(r1["status"], r2["status"]), error=r1.get("error")))
print("small_fuzz: %d cases run (%d draws not eligible), %d outside the bars" % (ran, skipped, len(bad)))
for b in bad:
    print(b)
