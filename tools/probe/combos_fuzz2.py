"""One-off: random-option fits of the later round-5 combinations against the oracle, fp64 --
  (a) dense input under a distribution loss through the 50-pointer dense entry (oracle dense_input),
  (b) cross-validation with a user mask through rcppml_gpu_nmf_cv_masked_ex (MSE and IRLS losses).
Usage: python tools/probe/combos_fuzz2.py [trials] [seed]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import oracle as O
from rcppml_amd import _abi
from tests.util import lowrank_csc, random_csc
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 80
rs = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
bad = 0
for trial in range(trials):
    if trial % 2 == 0:                                   # ---- (a) dense + loss
        k = int(rs.choice([2, 3, 5, 8, 16, 17, 32]))
        m, n = int(rs.integers(2 * k + 8, 2 * k + 70)), int(rs.integers(2 * k + 8, 2 * k + 90))
        loss_type = int(rs.choice([0, 4, 5, 6, 7, 8]))
        mu = rs.gamma(2.0, 1.0, (m, 3)) @ rs.gamma(2.0, 0.5, (3, n))
        M = rs.negative_binomial(4.0, 4.0 / (4.0 + mu)).astype(np.float64) if loss_type in (0, 4, 5, 8) else mu * rs.gamma(8.0, 0.125, (m, n)) + 0.05
        disp = int(rs.choice([0, 1, 2, 3]))
        robust = float(rs.choice([0.0, 0.0, 1.345]))
        if loss_type == 0 and robust == 0:
            robust = 1.345
        power = float(rs.choice([1.3, 1.5, 2.6]))
        L1 = (float(rs.choice([0.0, 0.01])), float(rs.choice([0.0, 0.02])))
        W0, H0 = O.init_factors(int(rs.integers(1, 1000)), k, m, n, np.float64)
        ref = O.nmf_fit(O.dense_as_csc(M), W0, H0, np.float64, max_iter=3, tol=0.0, cd_maxit=20, dense_input=True, loss_type=loss_type,
                        dispersion_mode=disp, tweedie_power=power, robust_delta=robust, L1=L1)
        W, H = W0.copy(), H0.copy()
        res = _abi.nmf_dense(M, k, W, H, entry="double", max_iter=3, tol=0.0, cd_maxit=20, loss_type=loss_type, dispersion_mode=disp,
                             tweedie_power=power, robust_delta=robust, L1_W=L1[0], L1_H=L1[1])
        cfg = ("dense", trial, k, m, n, loss_type, disp, robust, power, L1)
        if res["status"] != 0:
            bad += 1; print("FAILED", cfg, res["error"]); continue
        finite = np.isfinite(ref.loss) and abs(ref.loss) < 1e10
        ok = res["iter"] == ref.iter and (not finite or (abs(res["loss"] - ref.loss) <= 1e-5 * abs(ref.loss) + 1e-12
                                                         and np.abs(W - ref.W_T).max() < 1e-4 and np.abs(H - ref.H).max() < 1e-4))
        ok = ok and len(res["theta"]) == (n if disp == 3 else m)
        if not ok:
            bad += 1; print("MISMATCH", cfg, res["iter"], ref.iter, res["loss"], ref.loss, np.abs(W - ref.W_T).max(), np.abs(H - ref.H).max())
    else:                                                # ---- (b) CV + user mask
        k = int(rs.choice([2, 3, 4, 5, 8]))
        m, n = int(rs.integers(6 * k + 20, 6 * k + 90)), int(rs.integers(6 * k + 20, 6 * k + 120))
        loss_type = int(rs.choice([0, 0, 4, 6, 8]))
        A = lowrank_csc(m, n, 3, float(rs.choice([0.2, 0.4])), seed=3000 + trial)
        if loss_type in (4, 8):
            A.x[:] = np.round(A.x * 3) + 1.0
        Mk = random_csc(m, n, float(rs.choice([0.03, 0.1, 0.2])), seed=4000 + trial)
        mz = int(rs.integers(0, 2))
        solver = int(rs.integers(0, 2)) if loss_type != 4 else 0
        frac = float(rs.choice([0.1, 0.2]))
        cvs = int(rs.integers(1, 1000))
        W0, H0 = O.init_factors(int(rs.integers(1, 1000)), k, m, n, np.float64)
        kw = dict(max_iter=4, tol=0.0, solver_mode=solver, holdout_fraction=frac, cv_seed=cvs, cv_patience=0)
        ref = O.nmf_fit_cv(A, W0, H0, np.float64, L1=(0.0, 0.01), L2=(0.02, 0.0), mask_zeros=bool(mz), mask=Mk, loss_type=loss_type,
                           irls_max_iter=3, **kw)
        W, H = W0.copy(), H0.copy()
        res = _abi.nmf_cv(A.p, A.i, A.x, m, n, k, W, H, entry="irls_ex", L1_H=0.01, L2_W=0.02, mask_zeros=mz, sort_model=0, precision=_abi.F64,
                          loss_type=loss_type, irls_max_iter=3, mask=(Mk.p, Mk.i), **kw)
        cfg = ("cvmask", trial, k, m, n, loss_type, mz, solver, frac)
        if res["status"] != 0:
            bad += 1; print("FAILED", cfg, res.get("error")); continue
        ok = res["iter"] == ref.iter and np.allclose(res["test_history"], ref.test_history, rtol=1e-5, atol=0) \
            and np.allclose(res["train_history"], ref.train_history, rtol=1e-5, atol=0) and np.abs(W - ref.W_T).max() < 1e-4
        if not ok:
            bad += 1; print("MISMATCH", cfg, res["test_history"], ref.test_history, np.abs(W - ref.W_T).max())
print("trials", trials, "mismatches", bad)
