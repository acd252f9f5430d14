import numpy as np, sys
sys.path.insert(0, "/root/repo")
from oracle import oracle as O
from rcppml_amd import _abi
from tests.util import lowrank_csc
rs = np.random.default_rng(31337)
for trial in range(10):
    k = int(rs.choice([2, 3, 7, 9, 16, 17, 33]))
    m, n = int(rs.integers(k + 5, 120)), int(rs.integers(k + 5, 150))
    A = lowrank_csc(m, n, max(2, k // 2), float(rs.choice([0.15, 0.5])), seed=trial)
    W0, H0 = O.init_factors(int(rs.integers(1, 1000)), k, m, n, np.float64)
    solver = int(rs.integers(0, 2))
    L1 = (float(rs.choice([0.0, 0.01])), float(rs.choice([0.0, 0.05])))
    L2 = (float(rs.choice([0.0, 0.1])), float(rs.choice([0.0, 0.01])))
    ub = (0.0, float(rs.choice([0.0, 0.0, 0.2])))
    norm_type = int(rs.choice([0, 0, 1]))
    tol = float(rs.choice([0.0, 1e-4]))
    if trial != 5: continue
    print("cfg", k, m, n, solver, L1, L2, ub, norm_type, tol, "nnz", A.nnz)
    for it in (1, 2, 3, 5, 8, 15):
        for sv in (solver, 0):
            ref = O.nmf_fit(A, W0, H0, np.float64, max_iter=it, tol=0.0, solver_mode=sv, L1=L1, L2=L2, ub=ub, norm_type=norm_type)
            W, H = W0.copy(), H0.copy()
            res = _abi.nmf_unified(A.p, A.i, A.x, m, n, k, W, H, entry="double", max_iter=it, tol=0.0, solver_mode=sv,
                                   L1_W=L1[0], L1_H=L1[1], L2_W=L2[0], L2_H=L2[1], ub_W=ub[0], ub_H=ub[1], norm_type=norm_type)
            print("iters", it, "solver", sv, "loss gpu %.10g ref %.10g rel %.2e  dW %.2e dH %.2e" % (res["loss"], ref.loss, abs(res["loss"]-ref.loss)/abs(ref.loss), np.abs(W-ref.W_T).max(), np.abs(H-ref.H).max()))
