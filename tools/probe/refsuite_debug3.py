import sys, os
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib.util
spec = importlib.util.spec_from_file_location("rs", os.path.join(ROOT, "tests", "test_gpu_reference_suite.py"))
rs = importlib.util.module_from_spec(spec); spec.loader.exec_module(rs)
from rcppml_amd import nmf as N
import scipy.sparse as sp
O = rs.O
buf = np.fromfile(os.path.join(ROOT, "tests", "golden", "pbmc3k.spz"), dtype=np.uint8)
st, M, NN, nnz, vt = O.spz_info(buf); p, i, x = O.spz_decode(buf)
pb = sp.csc_matrix((np.asarray(x, np.float64), np.asarray(i, np.int32), np.asarray(p, np.int32)), shape=(M, NN))
S = rs.sub(pb, 300, 150); m, n = S.shape
W0, H0 = rs.inits(42, m, n, 5)
for it in (2, 3, 4, 5):
    ref = O.nmf_fit(rs.csc_o(S), W0, H0, np.float64, max_iter=it, tol=0.0, solver_mode=0, norm_type=2)
    mod = N.nmf(S, 5, maxit=it, tol=0.0, seed=42, precision="fp64", solver="cd", norm="none")
    print(os.environ.get("RCPPML_GPU_NO_GRAPH"), "none cd", it, "rel %.2e" % (abs(mod.misc["loss"] - ref.loss) / abs(ref.loss)), "hist gpu", mod.misc["loss_history"][-3:], "oracle", ref.loss_history[-3:])
