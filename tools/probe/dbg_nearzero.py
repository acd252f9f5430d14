import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O
from rcppml_amd import _abi
from tests.test_gpu_plugin import _csc_from_dense
rs = np.random.default_rng(5)
rs.uniform(size=(12, 9)); rs.uniform(0.1, 1, size=(1, 15)); rs.uniform(0.1, 1, size=(14, 1)); rs.uniform(size=(6, 7)); rs.uniform(0.5, 1, size=10)
D = rs.uniform(size=(9, 11)) * 1e-15
A = _csc_from_dense(D); m, n, k = 9, 11, 2
W0, H0 = O.init_factors(7, k, m, n, np.float64)
for it in (1, 2, 3, 4, 5, 6, 7, 8):
    W, H = W0.copy(), H0.copy()
    res = _abi.nmf_unified(A.p, A.i, A.x, m, n, k, W, H, entry=os.environ.get("ENTRY", "double"), max_iter=it, tol=0.0, solver_mode=0, sort_model=0)
    ref = O.nmf_fit(A, W0, H0, np.float32 if os.environ.get("ENTRY") == "float" else np.float64, max_iter=it, tol=0.0, solver_mode=0, sort_model=False)
    print("it", it, "gpu loss %.4g ref %.4g | d gpu %s ref %s | maxW %.3g/%.3g maxH %.3g/%.3g" % (res["loss"], ref.loss, res["d"], ref.d, W.max(), ref.W_T.max(), H.max(), ref.H.max()))
print("trAtA", np.sum(A.x**2))
