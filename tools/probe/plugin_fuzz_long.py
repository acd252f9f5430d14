"""One-off: the random-options fit comparison of tests/test_gpu_fuzz.py::test_fuzz_plugin_fits_fp64 with many more trials
(and fp32 against the fp64 oracle at a loose tolerance).  Usage: python tools/probe/plugin_fuzz_long.py [trials] [seed]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import oracle as O
from rcppml_amd import _abi
from tests.util import lowrank_csc
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rs = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2025)
bad = 0
for trial in range(trials):
    k = int(rs.choice([2, 3, 7, 9, 16, 17, 33, 48, 64, 65, 100, 128, 130]))
    m, n = int(rs.integers(k + 5, k + 160)), int(rs.integers(k + 5, k + 220))
    A = lowrank_csc(m, n, max(2, k // 2), float(rs.choice([0.15, 0.5])), seed=trial)
    W0, H0 = O.init_factors(int(rs.integers(1, 1000)), k, m, n, np.float64)
    solver = int(rs.integers(0, 2)) if k <= 64 else 0
    L1 = (float(rs.choice([0.0, 0.01])), float(rs.choice([0.0, 0.05])))
    L2 = (float(rs.choice([0.0, 0.1])), float(rs.choice([0.0, 0.01])))
    ub = (0.0, float(rs.choice([0.0, 0.0, 0.2])))
    norm_type = int(rs.choice([0, 0, 1]))
    tol = float(rs.choice([0.0, 1e-4]))
    ref = O.nmf_fit(A, W0, H0, np.float64, max_iter=4, tol=tol, solver_mode=solver, L1=L1, L2=L2, ub=ub, norm_type=norm_type)
    W, H = W0.copy(), H0.copy()
    res = _abi.nmf_unified(A.p, A.i, A.x, m, n, k, W, H, entry="double", max_iter=4, tol=tol, solver_mode=solver,
                           L1_W=L1[0], L1_H=L1[1], L2_W=L2[0], L2_H=L2[1], ub_W=ub[0], ub_H=ub[1], norm_type=norm_type)
    cfg = (trial, k, m, n, solver, L1, L2, ub, norm_type, tol)
    ok = (res["status"] == 0 and res["iter"] == ref.iter and abs(res["loss"] - ref.loss) <= 1e-6 * abs(ref.loss) + 1e-12
          and np.abs(W - ref.W_T).max() < 1e-6 and np.abs(H - ref.H).max() < 1e-6)
    if not ok:
        bad += 1
        print("MISMATCH", cfg, res.get("status"), res.get("error"), res.get("iter"), ref.iter, res.get("loss"), ref.loss)
print("trials", trials, "mismatches", bad)
