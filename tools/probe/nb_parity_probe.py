"""Why does a whole NB fit deviate from the CPU oracle by 1e-5 in fp64 when every half-update agrees to 1e-6 on identical inputs?
Hypothesis: the IRLS exit `max rel change < irls_tol` (1e-4) is a discrete decision -- a column near the threshold takes one pass
more or less under rounding-level differences and then differs by ~1e-4.  With irls_tol = 0 (always irls_max_iter passes) the
deviation should collapse."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O
from rcppml_amd import _abi, data
m, n, k = 4000, 60000, 32
A, _, _ = data.simulate_nb_counts(m, n, k, density=0.02, size=5.0, seed=123)
W0, H0 = data.init_factors(42, k, m, n, np.float64)
O.build(native=True)
C = O.Csc((m, n), A.p, A.i, A.x)
p, i, x = A.p.astype(np.int32), A.i.astype(np.int32), A.x.astype(np.float64)
for irls_tol in (1e-4, 0.0):
    for iters in (1, 2, 3):
        W, H = W0.copy(), H0.copy()
        res = _abi.nmf_unified(p, i, x, m, n, k, W, H, entry="double", max_iter=iters, tol=0.0, loss_type=5, irls_tol=irls_tol)
        ref = O.nmf_fit(C, W0, H0, np.float64, max_iter=iters, tol=0.0, loss_type=5, irls_tol=irls_tol, threads=0, native=True)
        rs = np.random.default_rng(1)
        ref2 = O.nmf_fit(C, W0 * (1 + 1e-14 * rs.standard_normal(W0.shape)), H0 * (1 + 1e-14 * rs.standard_normal(H0.shape)), np.float64,
                         max_iter=iters, tol=0.0, loss_type=5, irls_tol=irls_tol, threads=0, native=True)
        print("irls_tol %g iters %d: GPU vs CPU loss dev %.3e  H max dev %.3e  W max dev %.3e | CPU self dev %.3e" % (
            irls_tol, iters, abs(res["loss"] - ref.loss) / abs(ref.loss), np.abs(H - ref.H).max(), np.abs(W - ref.W_T).max(),
            abs(ref2.loss - ref.loss) / abs(ref.loss)))
