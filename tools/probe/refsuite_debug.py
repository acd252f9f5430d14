"""Where do the upper-bound / norm = none / graph fits on pbmc3k[1:300, 1:150] leave the oracle?  Per-iteration comparison."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import importlib.util
spec = importlib.util.spec_from_file_location("rs", os.path.join(os.path.dirname(__file__), "..", "..", "tests", "test_gpu_reference_suite.py"))
rs = importlib.util.module_from_spec(spec); spec.loader.exec_module(rs)
from rcppml_amd import nmf as N
import scipy.sparse as sp
O = rs.O
buf = np.fromfile(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "pbmc3k.spz"), dtype=np.uint8)
st, M, NN, nnz, vt = O.spz_info(buf); p, i, x = O.spz_decode(buf)
pb = sp.csc_matrix((np.asarray(x, np.float64), np.asarray(i, np.int32), np.asarray(p, np.int32)), shape=(M, NN))
S = rs.sub(pb, 300, 150); m, n = S.shape
W0, H0 = rs.inits(42, m, n, 5)
def chain(dim):
    Adj = sp.diags([np.ones(dim - 1), np.ones(dim - 1)], [-1, 1], format="csc")
    return sp.csc_matrix(sp.diags(np.asarray(Adj.sum(axis=0)).ravel()) - Adj)
LW, LH = chain(m), chain(n)
oc = lambda L: O.Csc(L.shape, L.indptr.astype(np.int32), L.indices.astype(np.int32), L.data.astype(np.float64))
cases = [("ub", dict(upper_bound=(0.5, 0.5)), dict(ub=(0.5, 0.5))),
         ("ubH", dict(upper_bound=(0.0, 0.5)), dict(ub=(0.0, 0.5))),
         ("ubW", dict(upper_bound=(0.5, 0.0)), dict(ub=(0.5, 0.0))),
         ("none", dict(norm="none"), dict(norm_type=2)),
         ("gH", dict(graph_H=LH, graph_lambda=(0.0, 0.1)), dict(graph_H=(oc(LH), 0.1))),
         ("gW", dict(graph_W=LW, graph_lambda=(0.1, 0.0)), dict(graph_W=(oc(LW), 0.1)))]
for name, kw, okw in cases:
    for solver, sm in (("cd", 0), ("cholesky", 1)):
        for it in (1, 2, 3, 20):
            ref = O.nmf_fit(rs.csc_o(S), W0, H0, np.float64, max_iter=it, tol=0.0, solver_mode=sm, **okw)
            mod = N.nmf(S, 5, maxit=it, tol=0.0, seed=42, precision="fp64", solver=solver, **kw)
            print(name, solver, it, "loss gpu %.10g oracle %.10g rel %.2e  dW %.2e dH %.2e dd %.2e" % (
                mod.misc["loss"], ref.loss, abs(mod.misc["loss"] - ref.loss) / abs(ref.loss), np.abs(mod.w - ref.W_T).max(),
                np.abs(mod.h.T - ref.H).max(), np.abs(mod.d - ref.d).max() / np.abs(ref.d).max()), flush=True)
