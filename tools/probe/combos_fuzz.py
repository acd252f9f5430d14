"""One-off: random-option fits of the round-5 combinations (explicit mask + distribution loss; dispersion = per_col) through
rcppml_gpu_nmf_ex (fp64) against the oracle.  Usage: python tools/probe/combos_fuzz.py [trials] [seed]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import oracle as O
from rcppml_amd import _abi, data
from tests.util import random_csc
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rs = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 77)
bad = 0
for trial in range(trials):
    k = int(rs.choice([2, 3, 5, 8, 16, 17, 32, 33, 64, 70, 96, 128]))
    m, n = int(rs.integers(3 * k + 8, 3 * k + 140)), int(rs.integers(3 * k + 8, 3 * k + 200))    # well-posed per-column systems: >= 2k unmasked rows
    loss_type = int(rs.choice([0, 4, 5, 6, 7, 8]))
    A0, _, _ = data.simulate_nb_counts(m, n, 3, density=float(rs.choice([0.1, 0.3])), size=5.0, seed=1000 + trial)
    A = O.Csc(A0.shape, A0.p, A0.i, A0.x)
    if loss_type >= 6:
        A.x[:] = A.x * rs.uniform(0.5, 1.5, size=A.x.shape) + 0.1
    use_mask = bool(rs.integers(0, 2))
    disp = int(rs.choice([0, 1, 2, 3])) if not use_mask else int(rs.choice([0, 1, 2]))
    if loss_type == 0:
        disp = 2
    robust = float(rs.choice([0.0, 0.0, 1.345])) if (use_mask or loss_type == 0) else 0.0
    if loss_type == 0 and not use_mask:
        use_mask = True                                   # plain MSE without a mask is not what this probe is about
    M = random_csc(m, n, float(rs.choice([0.02, 0.1, 0.3])), seed=2000 + trial) if use_mask else None
    power = float(rs.choice([1.3, 1.5, 2.6]))
    W0, H0 = O.init_factors(int(rs.integers(1, 1000)), k, m, n, np.float64)
    L1 = (float(rs.choice([0.0, 0.01])), float(rs.choice([0.0, 0.02])))
    iters = 3
    kw = dict(max_iter=iters, tol=0.0, loss_type=loss_type, dispersion_mode=disp, tweedie_power=power, robust_delta=robust, L1=L1)
    ref = O.nmf_fit(A, W0, H0, np.float64, mask=M, **kw)
    W, H = W0.copy(), H0.copy()
    res = _abi.nmf_unified(A.p, A.i, A.x, m, n, k, W, H, entry="ex", max_iter=iters, tol=0.0, loss_type=loss_type, gp_dispersion_mode=disp,
                           tweedie_power=power, robust_delta=robust, L1_W=L1[0], L1_H=L1[1], precision=1, mask=(M.p, M.i) if M is not None else None)
    cfg = (trial, k, m, n, loss_type, disp, use_mask, (M.nnz / float(m * n)) if M is not None else 0.0, robust, power, L1)
    if res["status"] != 0:
        bad += 1; print("FAILED", cfg, res.get("error")); continue
    # masked fits are MSE solves (tight); unmasked IRLS fits amplify rounding (looser, as in tests/test_gpu_nb.py)
    ltol, ftol = (1e-6, 1e-6) if use_mask else (1e-4, 1e-3)
    finite = np.isfinite(ref.loss) and abs(ref.loss) < 1e8
    ok = res["iter"] == ref.iter and (not finite or (abs(res["loss"] - ref.loss) <= ltol * abs(ref.loss) + 1e-12
                                                     and np.abs(W - ref.W_T).max() < ftol * max(1.0, np.abs(ref.W_T).max())
                                                     and np.abs(H - ref.H).max() < ftol * max(1.0, np.abs(ref.H).max())))
    want_len = (n if disp == 3 else m) if (loss_type != 0 or robust > 0) else 0
    ok = ok and len(res["theta"]) == want_len
    if not ok:
        bad += 1
        print("MISMATCH", cfg, res["iter"], ref.iter, res["loss"], ref.loss, np.abs(W - ref.W_T).max(), np.abs(H - ref.H).max(), len(res["theta"]), want_len)
print("trials", trials, "mismatches", bad)
