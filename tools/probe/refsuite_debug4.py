import sys, os
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib.util
spec = importlib.util.spec_from_file_location("rs", os.path.join(ROOT, "tests", "test_gpu_reference_suite.py"))
rs = importlib.util.module_from_spec(spec); spec.loader.exec_module(rs)
from rcppml_amd import nmf as N, _abi
import scipy.sparse as sp
O = rs.O
buf = np.fromfile(os.path.join(ROOT, "tests", "golden", "pbmc3k.spz"), dtype=np.uint8)
st, M, NN, nnz, vt = O.spz_info(buf); p, i, x = O.spz_decode(buf)
pb = sp.csc_matrix((np.asarray(x, np.float64), np.asarray(i, np.int32), np.asarray(p, np.int32)), shape=(M, NN))
S = rs.sub(pb, 300, 150); m, n = S.shape; A = rs.csc_o(S); At = A.transpose(); k = 5
D = S.toarray()
loss = lambda W, H: float(((D - W @ H.T) ** 2).sum())
W0, H0 = rs.inits(42, m, n, k)
for sort in (True, False):
    g2 = N.nmf(S, k, maxit=2, tol=0.0, seed=42, precision="fp64", solver="cd", norm="none", sort_model=sort)
    g3 = N.nmf(S, k, maxit=3, tol=0.0, seed=42, precision="fp64", solver="cd", norm="none", sort_model=sort)
    print("sort", sort, "gpu it2 loss", g2.misc["loss"], loss(g2.w, g2.h.T), "it3", g3.misc["loss"], loss(g3.w, g3.h.T), "d", g3.d)
    W2, H2 = g2.w, g2.h.T.copy()
    eye = 1e-15 * np.eye(k)
    H3 = O.fused_cd(A, W2, O.gram(W2) + eye, H2, warm=True)
    W3 = O.fused_cd(At, H3, O.gram(H3) + eye, W2, warm=True)
    print("   oracle pieces from the GPU's state after 2:", loss(W3, H3), " |H3 - gpu| %.2e |W3 - gpu| %.2e" % (np.abs(H3 - g3.h.T).max(), np.abs(W3 - g3.w).max()))
# the 73-pointer entry
for it in (2, 3):
    W, H = W0.copy(), H0.copy()
    r = _abi.nmf_unified(A.p, A.i, A.x, m, n, k, W, H, entry="double", max_iter=it, tol=0.0, solver_mode=0, norm_type=2)
    print("entry double it", it, r["loss"], loss(W, H), r["d"])
