import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import oracle as O
from rcppml_amd import _abi as abi
from tests.util import lowrank_csc
for k in (128, 100):
  for iters in (3, 8):
    A = lowrank_csc(500, 900, 12, 0.08, seed=k)
    W0, H0 = O.init_factors(77, k, A.rows, A.cols, np.float64)
    ref = O.nmf_fit(A, W0.astype(np.float32), H0.astype(np.float32), np.float32, max_iter=iters, tol=0.0, solver_mode=0)
    ref64 = O.nmf_fit(A, W0, H0, np.float64, max_iter=iters, tol=0.0, solver_mode=0)
    W, H = W0.copy(), H0.copy()
    res = abi.nmf_unified(A.p, A.i, A.x, A.rows, A.cols, k, W, H, entry="ex", max_iter=iters, tol=0.0, solver_mode=0, precision=0)
    print(k, iters, "loss rel", abs(res["loss"]-ref.loss)/abs(ref.loss), "d", np.abs(res["d"]-ref.d).max()/np.abs(ref.d).max(), "W", np.abs(W-ref.W_T).max(), "H", np.abs(H-ref.H).max(),
          "| oracle f32 vs f64: W", np.abs(ref.W_T-ref64.W_T).max(), "H", np.abs(ref.H-ref64.H).max(), "| gpu f32 vs oracle f64: W", np.abs(W-ref64.W_T).max())
