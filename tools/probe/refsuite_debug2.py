"""norm = "none" on pbmc3k[1:300, 1:150]: the CD half-updates of iteration 3, GPU kernel vs oracle, from the oracle's state after 2."""
import sys, os
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib.util, torch
spec = importlib.util.spec_from_file_location("rs", os.path.join(ROOT, "tests", "test_gpu_reference_suite.py"))
rs = importlib.util.module_from_spec(spec); spec.loader.exec_module(rs)
from rcppml_amd import _abi
import scipy.sparse as sp
O = rs.O
buf = np.fromfile(os.path.join(ROOT, "tests", "golden", "pbmc3k.spz"), dtype=np.uint8)
st, M, NN, nnz, vt = O.spz_info(buf); p, i, x = O.spz_decode(buf)
pb = sp.csc_matrix((np.asarray(x, np.float64), np.asarray(i, np.int32), np.asarray(p, np.int32)), shape=(M, NN))
S = rs.sub(pb, 300, 150); m, n = S.shape
A = rs.csc_o(S); At = A.transpose()
k = 5
W0, H0 = rs.inits(42, m, n, k)
ref2 = O.nmf_fit(A, W0, H0, np.float64, max_iter=2, tol=0.0, solver_mode=0, norm_type=2, sort_model=False)
W, H = ref2.W_T, ref2.H
ctx = _abi.Context(0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for side, (Ax, F, X) in (("H", (A, W, H)), ("W", (At, H, W))):
    G = O.gram(F) ; G[np.diag_indices(k)] += 0.0
    B = O.rhs(Ax, F)
    X_or = O.fused_cd(Ax, F, G, X, warm=True)
    X_nb = O.nnls_batch(G, B - X @ G.T, X, warm=True) if False else None
    dG, dB, dX = dev(G), dev(B), dev(X.copy())
    sw = torch.zeros(X.shape[0], dtype=torch.int32, device="cuda")
    ctx.solve_cd(_abi.F64, dG, dB, dX, k, X.shape[0], warm=1, zero_init=0, maxit=100, tol=1e-8, sweeps_out=sw)
    ctx.sync()
    Xg = dX.cpu().numpy()
    dif = np.abs(Xg - X_or).max(axis=1)
    j = int(np.argmax(dif))
    print(side, "max|dX| %.3e at column %d (sweeps there %d; max sweeps %d, columns at the cap %d)" % (dif.max(), j, int(sw[j]), int(sw.max()), int((sw >= 100).sum())), "scale", np.abs(X_or).max())
    # the worst column, one column at a time with the oracle's column solver at several sweep caps
    b = B[j] - G @ X[j]
    for cap in (99, 100, 101, 200, 1000):
        xo, _, it = O.cd_col(G, b.copy(), X[j].copy(), maxit=cap, tol=1e-8)
        print("   oracle cd_col cap %4d -> sweeps %4d  |x - gpu| %.3e |x - fused| %.3e" % (cap, it, np.abs(xo - Xg[j]).max(), np.abs(xo - X_or[j]).max()))
