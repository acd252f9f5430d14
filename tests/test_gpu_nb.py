"""GPU parity tests of the NB (negative-binomial) IRLS path -- BASELINE config 5 shape at test size
(reference primitives/cpu/nnls_batch_irls.hpp, nmf/fit_cpu.hpp:1094-1265, nmf/explicit_loss.hpp) -- kernel level
and through the 73-pointer plugin entry with loss_type = 5.

Tolerances: the IRLS weights hit the 1e6 cap on the first pass (x = 0) and the weighted Gram is a sum of up to
thousands of rank-1 terms with weights spanning 6 orders of magnitude, so fp64 agreement is ~1e-8, fp32 ~1e-2 on
individual solutions; the NB loss (a sum over all nonzeros) agrees far tighter."""
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _nb_problem(m, n, k, seed):
    from rcppml_amd import data
    A, w, h = data.simulate_nb_counts(m, n, k, density=0.15, size=5.0, seed=seed)
    return O.Csc(A.shape, A.p, A.i, A.x)


@pytest.fixture(scope="module")
def env():
    import torch
    from rcppml_amd import _abi
    return torch, _abi, _abi.Context(0)


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-7), (np.float32, 3e-2)])
@pytest.mark.parametrize("k", [4, 16, 32, 64, 80, 128])
def test_irls_nb_half_updates(env, dtype, tol, k):
    torch, _abi, ctx = env
    A = _nb_problem(150, 220, 4, seed=k)
    At = A.transpose()
    rng = np.random.default_rng(k)
    dt = _abi.F32 if dtype == np.float32 else _abi.F64
    tt = torch.float32 if dtype == np.float32 else torch.float64
    for (M, F_rows, by_row) in ((A, A.rows, True), (At, At.rows, False)):
        F = rng.uniform(0.05, 1.0, size=(F_rows, k)).astype(dtype)
        F /= F.sum(axis=0, keepdims=True)
        F *= 30.0
        G = O.gram(F)
        theta = rng.uniform(2.0, 20.0, size=(M.rows if by_row else M.cols)).astype(dtype)
        ref = O.irls_nb(M, F, G, k, L1=0.0, L2=1e-3, theta_row=theta if by_row else None,
                        theta_col=None if by_row else theta, dtype=dtype)
        dX = torch.full((M.cols, k), 3.0, dtype=tt, device="cuda")
        ctx.solve_irls_nb(dt, _dev(torch, M.p), _dev(torch, M.i), _dev(torch, M.values(dtype)), M.cols, _dev(torch, F),
                          _dev(torch, G), dX, k, l1=0.0, l2=1e-3, theta_row=_dev(torch, theta) if by_row else None,
                          theta_col=None if by_row else _dev(torch, theta))
        X = dX.cpu().numpy()
        assert X.min() >= 0 and np.all(np.isfinite(X))
        assert np.abs(X - ref).max() / np.abs(ref).max() < tol


@pytest.mark.parametrize("k", [16, 32, 48, 64])
@pytest.mark.parametrize("opts", [dict(L1=0.05), dict(nonneg=False), dict(L1=0.02, nonneg=False)])
def test_irls_static_sweep_options(env, k, opts):
    """The fp32 per-column-Gram kernels run their CD as static coordinate sweeps (cd_static_sweeps_f32, kernels.hip.h): the L1
    term inside the step and nonneg = FALSE (no clamp: the step is the plain quotient) against the oracle's
    cd_nnls_col_fixed restatement through the GP half-update (KL weights; nnls_batch_irls.hpp:202-329)."""
    torch, _abi, ctx = env
    A = _nb_problem(150, 220, 4, seed=k + 7)
    rng = np.random.default_rng(k + 1)
    F = rng.uniform(0.05, 1.0, size=(A.rows, k)).astype(np.float32)
    F /= F.sum(axis=0, keepdims=True)
    F *= 30.0
    G = O.gram(F)
    L1, nonneg = opts.get("L1", 0.0), opts.get("nonneg", True)
    ref = O.irls(4, A, F, G, k, L1=L1, L2=1e-3, nonneg=nonneg, dtype=np.float32)
    dX = torch.full((A.cols, k), 3.0, dtype=torch.float32, device="cuda")
    ctx.solve_irls(_abi.F32, 4, _dev(torch, A.p), _dev(torch, A.i), _dev(torch, A.values(np.float32)), A.cols, _dev(torch, F),
                   _dev(torch, G), dX, k, l1=L1, l2=1e-3, nonneg=int(nonneg))
    X = dX.cpu().numpy()
    assert np.all(np.isfinite(X)) and (not nonneg or X.min() >= 0)
    err = np.abs(X - ref) / np.abs(ref).max()
    if nonneg:
        assert err.max() < 3e-2
    else:
        # unclamped least squares by 100 fp32 CD sweeps on the weighted Grams: a handful of ill-conditioned columns amplify the
        # rounding differences between two fp32 evaluations of the same sweeps; everything else agrees as in the clamped case
        assert X.min() < 0 or ref.min() >= 0          # the unclamped solve really leaves the orthant where the oracle's does
        bad_cols = (err.max(axis=1) >= 3e-2).sum()
        assert bad_cols <= 2 and np.median(err) < 1e-4, (bad_cols, float(np.median(err)))


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-9), (np.float32, 2e-3)])
@pytest.mark.parametrize("k", [8, 96])
def test_nb_size_and_loss(env, dtype, tol, k):
    torch, _abi, ctx = env
    A = _nb_problem(120, 180, 3, seed=5)
    At = A.transpose()
    rng = np.random.default_rng(1)
    W_T = rng.uniform(size=(A.rows, k)).astype(dtype); W_T /= W_T.sum(axis=0, keepdims=True)
    H = rng.uniform(size=(A.cols, k)).astype(dtype); H /= H.sum(axis=0, keepdims=True)
    d = rng.uniform(50, 500, size=k).astype(dtype)
    th0 = np.full(A.rows, 10.0, dtype)
    ref_r = O.nb_size_update(A, W_T, H, d, th0, dtype=dtype)
    dt = _abi.F32 if dtype == np.float32 else _abi.F64
    dth = _dev(torch, th0)
    ctx.nb_size_update(dt, _dev(torch, At.p), _dev(torch, At.i), _dev(torch, At.values(dtype)), A.rows, _dev(torch, W_T),
                       _dev(torch, d), _dev(torch, H), A.cols, k, 0.01, 1e6, dth)
    r = dth.cpu().numpy()
    assert np.abs(r - ref_r).max() / np.abs(ref_r).max() < tol * 10
    out = torch.zeros(2, dtype=torch.float64, device="cuda")
    ctx.nb_loss(dt, _dev(torch, A.p), _dev(torch, A.i), _dev(torch, A.values(dtype)), A.cols, _dev(torch, W_T), _dev(torch, d),
                _dev(torch, H), _dev(torch, ref_r), k, out)
    ref_l = O.nb_loss(A, W_T, d, H, ref_r, dtype=dtype)
    assert abs(out[0].item() - ref_l) / abs(ref_l) < tol


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-12), (np.float32, 1e-9)])
@pytest.mark.parametrize("k", [8, 32, 96])
def test_nb_size_update_loss_fused(env, dtype, tol, k):
    """rcppml_hip_nb_size_update_loss (what the fit loop runs for per-row sizes) against the two separate calls: the sizes
    bit for bit, the likelihood to the order of its fp64 sum (by row instead of by column; the terms themselves are identical),
    and against the oracle at the tolerance of test_nb_size_and_loss.  Rows without nonzeros and a row with a huge count included."""
    torch, _abi, ctx = env
    A = _nb_problem(120, 180, 3, seed=5)
    x = A.x.copy()
    At0 = A.transpose()
    import scipy.sparse as sp
    S = sp.csc_matrix((x, A.i, A.p), shape=(A.rows, A.cols)).tolil()
    S[7, :] = 0
    S[50, :] = 0
    S[3, 11] = 5000
    S = S.tocsc(); S.eliminate_zeros(); S.sort_indices()
    A = O.Csc(S.shape, S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data.astype(np.float64))
    At = A.transpose()
    rng = np.random.default_rng(1)
    W_T = rng.uniform(size=(A.rows, k)).astype(dtype); W_T /= W_T.sum(axis=0, keepdims=True)
    H = rng.uniform(size=(A.cols, k)).astype(dtype); H /= H.sum(axis=0, keepdims=True)
    d = rng.uniform(50, 500, size=k).astype(dtype)
    th0 = np.full(A.rows, 10.0, dtype)
    dt = _abi.F32 if dtype == np.float32 else _abi.F64
    dev = lambda a: _dev(torch, a)
    tp, ti, tx, W_d, d_d, H_d = dev(At.p), dev(At.i), dev(At.values(dtype)), dev(W_T), dev(d), dev(H)
    th_sep = dev(th0)
    ctx.nb_size_update(dt, tp, ti, tx, A.rows, W_d, d_d, H_d, A.cols, k, 0.01, 1e6, th_sep)
    out_sep = torch.zeros(2, dtype=torch.float64, device="cuda")
    ctx.nb_loss(dt, dev(A.p), dev(A.i), dev(A.values(dtype)), A.cols, W_d, d_d, H_d, th_sep, k, out_sep)
    th_f = dev(th0)
    out_f = torch.zeros(2, dtype=torch.float64, device="cuda")
    ctx.nb_size_update_loss(dt, tp, ti, tx, A.rows, At.nnz, W_d, d_d, H_d, A.cols, k, 0.01, 1e6, th_f, out_f)
    assert torch.equal(th_sep, th_f)
    l_sep, l_f = out_sep[0].item(), out_f[0].item()
    assert np.isfinite(l_f) and abs(l_f - l_sep) / abs(l_sep) < tol, (l_f, l_sep)
    ref_r = O.nb_size_update(A, W_T, H, d, th0, dtype=dtype)
    ref_l = O.nb_loss(A, W_T, d, H, ref_r, dtype=dtype)
    assert abs(l_f - ref_l) / abs(ref_l) < (1e-9 if dtype == np.float64 else 2e-3)


@pytest.mark.parametrize("dispersion", [2, 1, 0])
def test_nb_fit_through_plugin(dispersion):
    """nmf(loss='nb') through rcppml_gpu_nmf_unified_double (loss_type = 5): theta returned through out_theta."""
    from rcppml_amd import _abi
    A = _nb_problem(100, 160, 3, seed=9)
    k = 5
    W0, H0 = O.init_factors(11, k, A.rows, A.cols, np.float64)
    ref = O.nmf_fit(A, W0, H0, np.float64, max_iter=6, tol=0.0, loss_type=5, dispersion_mode=dispersion, threads=1)
    W, H = W0.copy(), H0.copy()
    res = _abi.nmf_unified(A.p, A.i, A.x, A.rows, A.cols, k, W, H, entry="double", max_iter=6, tol=0.0, loss_type=5,
                           gp_dispersion_mode=dispersion)
    assert res["status"] == 0, res.get("error")
    assert res["iter"] == ref.iter
    # NB-IRLS amplifies rounding: first-pass weights sit at the 1e6 cap and the size estimate divides by an
    # "excess variance" that is a difference of large sums, so individual r_i can move by percents for 1e-6 changes
    # of the factors.  Loss and factors stay tight; theta is compared in distribution.
    assert abs(res["loss"] - ref.loss) / abs(ref.loss) < 1e-5
    assert np.abs(W - ref.W_T).max() < 1e-4 and np.abs(H - ref.H).max() < 1e-4
    assert np.abs(res["d"] - ref.d).max() / ref.d.max() < 1e-4
    assert len(res["theta"]) == A.rows
    rel = np.abs(res["theta"] - ref.theta) / np.abs(ref.theta)
    print("theta rel err: median %.2e p99 %.2e max %.2e" % (np.median(rel), np.percentile(rel, 99), rel.max()))
    assert np.median(rel) < 1e-4 and rel.max() < 0.2
    assert np.all(res["theta"] >= 0.01) and np.all(res["theta"] <= 1e6)


@pytest.mark.parametrize("loss_type,loss_kw", [(5, {}), (4, dict(gp_dispersion_mode=2)), (6, {})])
def test_irls_fit_through_plugin_at_ranks_above_64(loss_type, loss_kw):
    """The 73-pointer entry with an IRLS loss at k = 72 / 128 (BASELINE configs[3]'s rank): half-updates, dispersion updates and
    likelihoods run on the two-features-per-lane kernels; fp64 against the oracle's fit."""
    from rcppml_amd import _abi
    A = _nb_problem(90, 140, 3, seed=4) if loss_type != 6 else _positive_problem(90, 140, seed=4)
    k = 128 if loss_type == 5 else 72
    W0, H0 = O.init_factors(3, k, A.rows, A.cols, np.float64)
    ref = O.nmf_fit(A, W0, H0, np.float64, max_iter=3, tol=0.0, loss_type=loss_type, threads=0,
                    **({"dispersion_mode": 2} if loss_type != 5 else {}))
    W, H = W0.copy(), H0.copy()
    res = _abi.nmf_unified(A.p, A.i, A.x, A.rows, A.cols, k, W, H, entry="double", max_iter=3, tol=0.0, loss_type=loss_type, **loss_kw)
    assert res["status"] == 0, res.get("error")
    assert res["iter"] == ref.iter
    assert abs(res["loss"] - ref.loss) / abs(ref.loss) < 1e-6
    # (72 factors on a 90 x 140 matrix: the Gamma weights 1 / mu^2 of near-zero predictions leave the factors loosely determined --
    # the deviance agrees to 1e-6, single factor entries to 1e-3)
    ftol = 1e-3 if loss_type == 6 else 1e-5
    assert np.abs(W - ref.W_T).max() < ftol * max(1.0, np.abs(ref.W_T).max()) and np.abs(H - ref.H).max() < ftol * max(1.0, np.abs(ref.H).max())


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-7), (np.float32, 3e-2)])
@pytest.mark.parametrize("k", [4, 12, 32, 33, 100])
def test_irls_gp_half_update_and_loss(env, dtype, tol, k):
    """loss = "gp" (LossType 4): half-updates with the KL weight 1/max(mu, 1e-4) (fit_cpu.hpp:568-574) and the GP
    likelihood over the nonzeros (math/loss.hpp:382-398), vs the oracle (whose per-element pieces are pinned to the
    reference's math/loss.hpp bit for bit, tests/test_oracle_ref.py)."""
    torch, _abi, ctx = env
    A = _nb_problem(150, 220, 4, seed=k + 100)
    rng = np.random.default_rng(k)
    dt = _abi.F32 if dtype == np.float32 else _abi.F64
    tt = torch.float32 if dtype == np.float32 else torch.float64
    F = rng.uniform(0.05, 1.0, size=(A.rows, k)).astype(dtype)
    F /= F.sum(axis=0, keepdims=True)
    F *= 30.0
    G = O.gram(F)
    ref = O.irls(4, A, F, G, k, L1=0.0, L2=1e-3, dtype=dtype)
    dX = torch.full((A.cols, k), 3.0, dtype=tt, device="cuda")
    ctx.solve_irls(dt, 4, _dev(torch, A.p), _dev(torch, A.i), _dev(torch, A.values(dtype)), A.cols, _dev(torch, F), _dev(torch, G),
                   dX, k, l1=0.0, l2=1e-3)
    X = dX.cpu().numpy()
    assert X.min() >= 0 and np.all(np.isfinite(X))
    assert np.abs(X - ref).max() / np.abs(ref).max() < tol
    # GP likelihood with theta = 0 and with a positive theta
    d = rng.uniform(0.5, 2.0, size=k).astype(dtype)
    for th_val in (0.0, 0.3):
        theta = np.full(A.rows, th_val, dtype)
        lref = O.irls_loss(4, A, F, d, X.astype(dtype), theta, dtype=dtype)
        out = torch.zeros((1,), dtype=torch.float64, device="cuda")
        ctx.irls_loss(dt, 4, _dev(torch, A.p), _dev(torch, A.i), _dev(torch, A.values(dtype)), A.cols, _dev(torch, F), _dev(torch, d),
                      _dev(torch, X.astype(dtype)), _dev(torch, theta), k, out)
        assert abs(float(out.item()) - lref) <= (1e-9 if dtype == np.float64 else 2e-3) * abs(lref)


def test_gp_fit_through_plugin_and_python_surface():
    """nmf(loss = "gp", dispersion = "none") = Poisson / KL-divergence NMF: 73-pointer entry (loss_type 4) vs the oracle
    fit; dispersion other than none is refused (out_status = -1); the Python mirror accepts loss = "gp"."""
    from rcppml_amd import _abi, nmf as N
    A = _nb_problem(100, 160, 3, seed=19)
    k = 6
    W0, H0 = O.init_factors(11, k, A.rows, A.cols, np.float64)
    ref = O.nmf_fit(A, W0, H0, np.float64, max_iter=6, tol=0.0, loss_type=4, dispersion_mode=0, threads=1)
    W, H = W0.copy(), H0.copy()
    res = _abi.nmf_unified(A.p, A.i, A.x, A.rows, A.cols, k, W, H, entry="double", max_iter=6, tol=0.0, loss_type=4,
                           gp_dispersion_mode=0)
    assert res["status"] == 0, res.get("error")
    assert res["iter"] == ref.iter
    # IRLS amplifies rounding (weights 1/mu with mu near the 1e-4 floor in the first passes): same bars as the NB fit
    assert abs(res["loss"] - ref.loss) / abs(ref.loss) < 2e-5
    assert np.abs(W - ref.W_T).max() < 1e-4 and np.abs(H - ref.H).max() < 1e-4
    assert np.abs(res["d"] - ref.d).max() / ref.d.max() < 1e-4
    assert np.all(res["theta"] == 0)
    W, H = W0.copy(), H0.copy()
    bad = _abi.nmf_unified(A.p, A.i, A.x, A.rows, A.cols, k, W, H, entry="double", max_iter=2, tol=0.0, loss_type=4,
                           gp_dispersion_mode=3)
    assert bad["status"] == -1                       # per-column dispersion is handed back
    from rcppml_amd.data import CSC
    Ap = CSC((A.rows, A.cols), A.p, A.i, A.x)
    model = N.nmf(Ap, k, loss="gp", dispersion="none", seed=3, maxit=5, tol=0.0)
    assert model.misc["loss_type"] == "gp" and np.isfinite(model.misc["loss"]) and model.w.min() >= 0 and model.h.min() >= 0
    pc = N.nmf(Ap, k, loss="gp", dispersion="per_col", seed=3, maxit=2)            # round 5: one theta per column (tests/test_gpu_combos.py)
    assert len(pc.misc["theta"]) == Ap.shape[1]
    disp = N.nmf(Ap, k, loss="gp", seed=3, maxit=5, tol=0.0)          # the R default: dispersion = "per_row"
    assert disp.misc["theta"].shape == (A.rows,) and disp.misc["theta"].max() > 0 and disp.misc["theta"].max() <= 5.0


def _positive_problem(m, n, seed):
    """Strictly positive continuous data for the Gamma / inverse-Gaussian / Tweedie deviances."""
    A = _nb_problem(m, n, 3, seed=seed)
    rs = np.random.default_rng(seed)
    A.x[:] = A.x * rs.uniform(0.5, 1.5, size=A.x.shape) + 0.1
    return A


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 3e-2)])
@pytest.mark.parametrize("loss_type,power", [(6, 0.0), (7, 0.0), (8, 1.5), (8, 2.7)])
def test_irls_power_family_half_update_and_loss(env, dtype, tol, loss_type, power):
    """Gamma (6), inverse Gaussian (7), Tweedie (8): power-variance weights min(1/mu^p, 1e6) and deviance terms vs the
    oracle (per-element pieces pinned to the reference's math/loss.hpp)."""
    torch, _abi, ctx = env
    k = 12
    A = _positive_problem(150, 220, seed=loss_type * 10 + int(power * 10))
    rng = np.random.default_rng(loss_type)
    dt = _abi.F32 if dtype == np.float32 else _abi.F64
    tt = torch.float32 if dtype == np.float32 else torch.float64
    F = rng.uniform(0.05, 1.0, size=(A.rows, k)).astype(dtype)
    F /= F.sum(axis=0, keepdims=True)
    F *= 30.0
    G = O.gram(F)
    ref = O.irls(loss_type, A, F, G, k, L1=0.0, L2=1e-3, dtype=dtype, power=power)
    dX = torch.full((A.cols, k), 3.0, dtype=tt, device="cuda")
    ctx.solve_irls(dt, loss_type, _dev(torch, A.p), _dev(torch, A.i), _dev(torch, A.values(dtype)), A.cols, _dev(torch, F),
                   _dev(torch, G), dX, k, l1=0.0, l2=1e-3, loss_param=power)
    X = dX.cpu().numpy()
    assert X.min() >= 0 and np.all(np.isfinite(X))
    assert np.abs(X - ref).max() / np.abs(ref).max() < tol
    d = rng.uniform(0.5, 2.0, size=k).astype(dtype)
    theta = np.ones(A.rows, dtype)
    lref = O.irls_loss(loss_type, A, F, d, X.astype(dtype), theta, dtype=dtype, power=power)
    out = torch.zeros((1,), dtype=torch.float64, device="cuda")
    ctx.irls_loss(dt, loss_type, _dev(torch, A.p), _dev(torch, A.i), _dev(torch, A.values(dtype)), A.cols, _dev(torch, F),
                  _dev(torch, d), _dev(torch, X.astype(dtype)), _dev(torch, theta), k, out, loss_param=power)
    assert abs(float(out.item()) - lref) <= (1e-9 if dtype == np.float64 else 2e-3) * abs(lref)


@pytest.mark.parametrize("loss,loss_type,power", [("gamma", 6, 1.5), ("inverse_gaussian", 7, 1.5), ("tweedie", 8, 1.3)])
def test_power_family_fit_through_plugin(loss, loss_type, power):
    from rcppml_amd import _abi, nmf as N
    from rcppml_amd.data import CSC
    A = _positive_problem(100, 160, seed=loss_type)
    k = 5
    W0, H0 = O.init_factors(11, k, A.rows, A.cols, np.float64)
    ref = O.nmf_fit(A, W0, H0, np.float64, max_iter=5, tol=0.0, loss_type=loss_type, dispersion_mode=0, threads=1, tweedie_power=power)
    W, H = W0.copy(), H0.copy()
    res = _abi.nmf_unified(A.p, A.i, A.x, A.rows, A.cols, k, W, H, entry="double", max_iter=5, tol=0.0, loss_type=loss_type,
                           gp_dispersion_mode=0, tweedie_power=power)
    assert res["status"] == 0, res.get("error")
    assert res["iter"] == ref.iter
    # weights 1/mu^p start at the 1e6 cap (x = 0 in the first pass) and the alternating map amplifies rounding strongly:
    # the half-updates agree to 1e-6 at kernel level (test above); the 5-iteration fit is held to looser bars
    assert np.all(np.isfinite(W)) and np.all(np.isfinite(H)) and W.min() >= 0 and H.min() >= 0
    if abs(ref.loss) < 1e8:     # inverse Gaussian from a random start drives predictions to the 1e-10 floor (loss ~1e22
        #                         in the oracle too): nothing to compare there beyond finiteness
        assert abs(res["loss"] - ref.loss) / abs(ref.loss) < 1e-3
        assert np.abs(W - ref.W_T).max() < 2e-2 * np.abs(ref.W_T).max() + 1e-3
        assert np.abs(H - ref.H).max() < 2e-2 * np.abs(ref.H).max() + 1e-3
    model = N.nmf(CSC((A.rows, A.cols), A.p, A.i, A.x), k, loss=loss, dispersion="none", seed=3, maxit=4, tol=0.0, tweedie_power=power)
    assert model.misc["loss_type"] == loss and np.isfinite(model.misc["loss"])
    pc = N.nmf(CSC((A.rows, A.cols), A.p, A.i, A.x), k, loss=loss, dispersion="per_col", seed=3, maxit=2)
    assert len(pc.misc["theta"]) == A.cols


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 3e-2)])
@pytest.mark.parametrize("loss_type", [0, 4, 5, 6])
def test_irls_robust_half_update_and_loss(env, dtype, tol, loss_type):
    """robust_delta > 0: every distribution weight is multiplied by the Huber modifier of the Pearson residual
    (nnls_batch_irls.hpp:95-120) and the loss becomes the Huber rho of that residual (math/loss.hpp:549-607); loss_type 0
    (MSE) is routed through the IRLS path.  Oracle pieces pinned to the reference in tests/test_oracle_ref.py."""
    torch, _abi, ctx = env
    k, delta = 12, 1.345
    A = _positive_problem(150, 220, seed=40 + loss_type)
    rng = np.random.default_rng(loss_type + 7)
    dt = _abi.F32 if dtype == np.float32 else _abi.F64
    tt = torch.float32 if dtype == np.float32 else torch.float64
    F = rng.uniform(0.05, 1.0, size=(A.rows, k)).astype(dtype)
    F /= F.sum(axis=0, keepdims=True)
    F *= 30.0
    G = O.gram(F)
    theta = rng.uniform(2.0, 20.0, size=A.rows).astype(dtype)
    th_arg = theta if loss_type == 5 else None
    ref = O.irls(loss_type, A, F, G, k, L1=0.0, L2=1e-3, theta_row=th_arg, dtype=dtype, robust=delta)
    dX = torch.full((A.cols, k), 3.0, dtype=tt, device="cuda")
    ctx.solve_irls(dt, loss_type, _dev(torch, A.p), _dev(torch, A.i), _dev(torch, A.values(dtype)), A.cols, _dev(torch, F),
                   _dev(torch, G), dX, k, l1=0.0, l2=1e-3, theta_row=_dev(torch, theta) if loss_type == 5 else None,
                   robust_delta=delta)
    X = dX.cpu().numpy()
    assert X.min() >= 0 and np.all(np.isfinite(X))
    assert np.abs(X - ref).max() / np.abs(ref).max() < tol
    d = rng.uniform(0.5, 2.0, size=k).astype(dtype)
    lref = O.irls_loss(loss_type, A, F, d, X.astype(dtype), theta, dtype=dtype, robust=delta)
    out = torch.zeros((1,), dtype=torch.float64, device="cuda")
    ctx.irls_loss(dt, loss_type, _dev(torch, A.p), _dev(torch, A.i), _dev(torch, A.values(dtype)), A.cols, _dev(torch, F),
                  _dev(torch, d), _dev(torch, X.astype(dtype)), _dev(torch, theta), k, out, robust_delta=delta)
    assert abs(float(out.item()) - lref) <= (1e-9 if dtype == np.float64 else 2e-3) * abs(lref)


@pytest.mark.parametrize("loss_type", [0, 5])
def test_robust_fit_through_plugin(loss_type):
    from rcppml_amd import _abi
    A = _nb_problem(100, 160, 3, seed=23)
    k = 5
    W0, H0 = O.init_factors(11, k, A.rows, A.cols, np.float64)
    ref = O.nmf_fit(A, W0, H0, np.float64, max_iter=5, tol=0.0, loss_type=loss_type, dispersion_mode=2, threads=1, robust_delta=1.345)
    W, H = W0.copy(), H0.copy()
    res = _abi.nmf_unified(A.p, A.i, A.x, A.rows, A.cols, k, W, H, entry="double", max_iter=5, tol=0.0, loss_type=loss_type,
                           gp_dispersion_mode=2, robust_delta=1.345)
    assert res["status"] == 0, res.get("error")
    assert res["iter"] == ref.iter
    assert abs(res["loss"] - ref.loss) / abs(ref.loss) < 1e-4
    assert np.abs(W - ref.W_T).max() < 1e-3 and np.abs(H - ref.H).max() < 1e-3


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-10), (np.float32, 2e-4)])
@pytest.mark.parametrize("loss_type,power", [(4, 0.0), (6, 0.0), (7, 0.0), (8, 1.4)])
@pytest.mark.parametrize("mode", [2, 1])
def test_dispersion_update(env, dtype, tol, loss_type, power, mode):
    """rcppml_hip_dispersion_update: GP theta by five MM passes (fit_cpu.hpp:914-1008) and Gamma / inverse-Gaussian /
    Tweedie phi by Pearson moments (:1561-1670), PER_ROW and GLOBAL (mean / median), vs the oracle restatement.  The
    per-row sums are fp64 on both sides; the GPU's wave-tree order differs from the oracle's sequential one."""
    torch, _abi, ctx = env
    A = _nb_problem(130, 190, 3, seed=5 + loss_type)
    At = A.transpose()
    k = 8
    rng = np.random.default_rng(loss_type)
    W_T = rng.uniform(size=(A.rows, k)).astype(dtype); W_T /= W_T.sum(axis=0, keepdims=True)
    H = rng.uniform(size=(A.cols, k)).astype(dtype); H /= H.sum(axis=0, keepdims=True)
    d = (rng.uniform(50, 500, size=k) * (1 if loss_type == 4 else 40)).astype(dtype)     # phi: predictions of the size of the data
    th0 = np.full(A.rows, 0.1 if loss_type == 4 else 1.0, dtype)
    lo, hi = (0.0, 5.0) if loss_type == 4 else (1e-6, 1e4)
    ref = O.dispersion_update(loss_type, A, W_T, H, d, th0, dispersion_mode=mode, power=power, lo=lo, hi=hi, dtype=dtype)
    dt = _abi.F32 if dtype == np.float32 else _abi.F64
    dth = _dev(torch, th0)
    ctx.dispersion_update(dt, loss_type, mode, _dev(torch, At.p), _dev(torch, At.i), _dev(torch, At.values(dtype)), A.rows, A.nnz,
                          _dev(torch, W_T), _dev(torch, d), _dev(torch, H), A.cols, k, power, lo, hi, dth)
    got = dth.cpu().numpy()
    assert np.abs(got - ref).max() / np.abs(ref).max() < tol
    assert ref.min() >= lo and ref.max() <= hi and (mode == 1) == (np.unique(ref).size == 1)
    if mode == 2:
        assert np.unique(ref).size > A.rows // 4           # a real per-row estimate, not the clamp everywhere


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_vec_global(env, dtype):
    torch, _abi, ctx = env
    dt = _abi.F32 if dtype == np.float32 else _abi.F64
    for m in (1, 2, 7, 1000, 4097):
        x = np.random.default_rng(m).standard_normal(m).astype(dtype) * 100
        dx = _dev(torch, x)
        ctx.vec_global(dt, 1, dx, m)
        assert np.array_equal(dx.cpu().numpy(), np.full(m, np.sort(x)[m // 2], dtype))      # nth_element at m/2: exact
        dx = _dev(torch, x)
        ctx.vec_global(dt, 0, dx, m)
        assert np.allclose(dx.cpu().numpy(), x.astype(np.float64).mean(), rtol=1e-6 if dtype == np.float32 else 1e-14, atol=1e-4 if dtype == np.float32 else 1e-12)


@pytest.mark.parametrize("loss_type,power", [(4, 1.5), (6, 1.5), (8, 1.3)])
@pytest.mark.parametrize("dispersion", [2, 1])
def test_dispersion_fit_through_plugin(loss_type, power, dispersion):
    """GP / Gamma / Tweedie with dispersion = per_row / global through the 73-pointer entry: theta / phi come back through
    out_theta; for GP theta enters the likelihood (so the loss and the stopping rule depend on it)."""
    from rcppml_amd import _abi
    A = _nb_problem(100, 160, 3, seed=9) if loss_type == 4 else _positive_problem(100, 160, seed=loss_type)
    k = 5
    W0, H0 = O.init_factors(11, k, A.rows, A.cols, np.float64)
    ref = O.nmf_fit(A, W0, H0, np.float64, max_iter=5, tol=0.0, loss_type=loss_type, dispersion_mode=dispersion, threads=1, tweedie_power=power)
    none = O.nmf_fit(A, W0, H0, np.float64, max_iter=5, tol=0.0, loss_type=loss_type, dispersion_mode=0, threads=1, tweedie_power=power)
    W, H = W0.copy(), H0.copy()
    res = _abi.nmf_unified(A.p, A.i, A.x, A.rows, A.cols, k, W, H, entry="double", max_iter=5, tol=0.0, loss_type=loss_type,
                           gp_dispersion_mode=dispersion, tweedie_power=power)
    assert res["status"] == 0, res.get("error")
    assert res["iter"] == ref.iter
    assert abs(res["loss"] - ref.loss) / abs(ref.loss) < 1e-3
    rel = np.abs(res["theta"] - ref.theta) / np.abs(ref.theta)
    print("theta rel err: median %.2e max %.2e" % (np.median(rel), rel.max()))
    assert np.median(rel) < 1e-3 and rel.max() < 0.2
    assert not np.array_equal(ref.theta, none.theta)                      # the estimator ran
    if loss_type == 4:
        assert abs(ref.loss - none.loss) > 1e-3 * abs(none.loss)          # and theta changed the GP likelihood


def _pick_cols(A, cols):
    return O.Csc((A.rows, len(cols)), np.concatenate([[0], np.cumsum(np.diff(A.p)[cols])]).astype(np.int32),
                 np.concatenate([A.i[A.p[c]:A.p[c + 1]] for c in cols]),
                 np.concatenate([A.x[A.p[c]:A.p[c + 1]] for c in cols]))


@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_full_size_c5_nb_properties(env, precision):
    """BASELINE configs[4] at FULL size: loss = 'nb' on 10 000 x 200 000 Poisson-Gamma counts (2 % dense), k = 32, fp32 (what the
    reference computes in) and fp64 (parity mode: the sampled columns must agree with the oracle to 1e-6).
    Two outer iterations driven op by op (IRLS half-update of H, scaling, IRLS half-update of W, scaling, method-of-moments
    size update, NB likelihood): everything finite and non-negative, the likelihood not increasing beyond the reference's
    slack, sizes inside their clamp, and 128 sampled columns of each IRLS half-update recomputed by the oracle's irls_nb from
    the device's own inputs (columns are independent given those; fp32 tolerance of the kernel-level NB tests)."""
    torch, _abi, ctx = env
    from rcppml_amd import als, data
    m, n, k = 10000, 200000, 32
    A, _, _ = data.simulate_nb_counts(m, n, k, density=0.02, size=5.0, seed=123)
    assert 3.0e7 < A.nnz < 4.5e7
    At = A.transpose()
    nd = np.float32 if precision == "f32" else np.float64
    W0, H0 = data.init_factors(42, k, m, n, nd)
    ops = als.HipOps(0, precision)
    W, H = ops.to_device(W0), ops.to_device(H0)
    Ad, Atd = ops.upload_csc(A), ops.upload_csc(At)
    theta = torch.full((m,), 10.0, dtype=ops.tdtype, device="cuda")              # nb_size_init (core/config.hpp)
    d = torch.ones((k,), dtype=ops.tdtype, device="cuda")
    sums = ops.empty((k,))
    out = torch.zeros((2,), dtype=torch.float64, device="cuda")
    losses = []
    for it in range(2):
        for side in ("H", "W"):
            F, X, csc, host = (W, H, Ad, A) if side == "H" else (H, W, Atd, At)
            G = ops.gram(F, 1e-15, 0.0)
            F_host, G_host, th_host = F.cpu().numpy(), G.cpu().numpy(), theta.cpu().numpy()
            ops.ctx.solve_irls_nb(ops.dt, csc["p"], csc["i"], csc["x"], csc["cols"], F, G, X, k, 0.0, 0.0, 1, 100, 5, 1e-4,
                                  theta if side == "H" else None, None if side == "H" else theta)
            X_new = X.cpu().numpy()
            assert np.all(np.isfinite(X_new)) and X_new.min() >= 0
            cols = np.sort(np.random.default_rng(3 * it + (side == "W")).choice(host.cols, size=128, replace=False))
            sub = _pick_cols(host, cols)
            ref = O.irls_nb(sub, F_host, G_host, k, L1=0.0, L2=0.0, theta_row=th_host if side == "H" else None,
                            theta_col=None if side == "H" else th_host[cols], dtype=nd)
            # per-column deviation relative to the column's largest entry.  The W side solves rows of A with thousands of
            # nonzeros whose first-pass weights sit at the 1e6 cap (x = 0): a few such systems are ill-conditioned enough
            # that fp32 summation order moves single entries by several percent -- most columns agree to 1e-4
            errc = np.abs(X_new[cols] - ref).max(axis=1) / (np.abs(ref).max(axis=1) + 1e-30)
            worst = int(np.argmax(errc))
            if precision == "f64":
                assert errc.max() <= 1e-6, (it, side, np.percentile(errc, [50, 90, 99, 100]), "worst column nnz", int(sub.p[worst + 1] - sub.p[worst]))
            else:
                assert np.median(errc) < 2e-3 and np.mean(errc < 3e-2) >= 0.95, (
                    it, side, np.percentile(errc, [50, 90, 99, 100]), "worst column nnz", int(sub.p[worst + 1] - sub.p[worst]),
                    "GPU", X_new[cols][worst][:6], "oracle", ref[worst][:6])
                # the outliers must be columns on which fp32 itself is not trustworthy: the ORACLE's fp32 solve of the same column
                # deviates from its fp64 solve (same inputs widened) by a comparable amount -- not an unexplained 0.2
                outl = np.nonzero(errc > 3e-2)[0]
                if outl.size:
                    sub_o = _pick_cols(sub, outl)
                    ref64 = O.irls_nb(sub_o, F_host.astype(np.float64), G_host.astype(np.float64), k, L1=0.0, L2=0.0,
                                      theta_row=th_host.astype(np.float64) if side == "H" else None,
                                      theta_col=None if side == "H" else th_host[cols][outl].astype(np.float64), dtype=np.float64)
                    dev_ref = np.abs(ref[outl] - ref64).max(axis=1) / (np.abs(ref64).max(axis=1) + 1e-30)
                    assert np.all(errc[outl] <= np.maximum(3e-2, 20.0 * dev_ref)), (it, side, errc[outl], dev_ref)
            ops.row_norms(X, 0, out=sums)
            ops.apply_scaling(X, sums, 0, d)
        ops.ctx.nb_size_update(ops.dt, Atd["p"], Atd["i"], Atd["x"], m, W, d, H, n, k, 0.01, 1e6, theta)
        th = theta.cpu().numpy()
        assert np.all(np.isfinite(th)) and th.min() >= 0.01 and th.max() <= 1e6
        ops.ctx.nb_loss(ops.dt, Ad["p"], Ad["i"], Ad["x"], n, W, d, H, theta, k, out)
        losses.append(float(out[0].item()))
    assert np.all(np.isfinite(losses)) and losses[1] <= losses[0] * (1 + 1e-3)


def test_nb_fit_with_upper_bounds():
    """upper_bound with an IRLS loss: clipped after both half-updates as the reference does (fit_cpu.hpp:636-637, 884-885)."""
    from rcppml_amd import _abi
    A = _nb_problem(90, 130, 3, seed=8)
    k = 4
    W0, H0 = O.init_factors(4, k, A.rows, A.cols, np.float64)
    ref = O.nmf_fit(A, W0, H0, np.float64, max_iter=5, tol=0.0, loss_type=5, ub=(0.05, 0.03))
    W, H = W0.copy(), H0.copy()
    res = _abi.nmf_unified(A.p, A.i, A.x, A.rows, A.cols, k, W, H, entry="ex", max_iter=5, tol=0.0, loss_type=5, precision=1,
                           ub_W=0.05, ub_H=0.03)
    assert res["status"] == 0, res.get("error")
    assert abs(res["loss"] - ref.loss) / abs(ref.loss) < 1e-6
    assert np.abs(W - ref.W_T).max() < 1e-6 and np.abs(H - ref.H).max() < 1e-6


@pytest.mark.parametrize("k", [8, 20, 32])
@pytest.mark.parametrize("loss,opts", [(5, dict()), (5, dict(l1=0.05)), (5, dict(nonneg=0)), (4, dict()), (6, dict(robust_delta=1.5)),
                                       (5, dict(cd_maxit=3, irls_max_iter=2))])
def test_quad_equals_single(env, k, loss, opts):
    """Four columns per wavefront (irls_nb_mfma32q_kernel, what many-column sides run) against one (irls_nb_mfma32_kernel), forced
    through RCPPML_OPT_IRLS_COLUMNS_PER_WAVE: the arithmetic of a column is the same, so the results are bit-identical --
    ragged and empty columns, a column count that is no multiple of 16, theta by row and by column, L1, no clamp, robust
    weights, truncated sweeps."""
    torch, _abi, ctx = env
    rng = np.random.default_rng(100 * k + loss)
    A = _nb_problem(180, 1003, 4, seed=k + loss)
    # empty and near-empty columns, one dense column
    x = A.x.copy()
    p = A.p
    for j in (0, 5, 17, 500, 1002):
        x[p[j]:p[j + 1]] = 0
    import scipy.sparse as sp
    S = sp.csc_matrix((x, A.i, A.p), shape=(A.rows, A.cols)).tolil()
    S[:, 33] = rng.integers(1, 9, size=(A.rows, 1))
    S = S.tocsc()
    S.eliminate_zeros()
    S.sort_indices()
    M = O.Csc(S.shape, S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data.astype(np.float64))
    for by_row in (True, False):
        F = rng.uniform(0.05, 1.0, size=(M.rows, k)).astype(np.float32)
        F /= F.sum(axis=0, keepdims=True)
        F *= 30.0
        G = O.gram(F)
        theta = rng.uniform(2.0, 20.0, size=(M.rows if by_row else M.cols)).astype(np.float32)
        out = {}
        for cpw in (1, 4):
            ctx.set_option(_abi.OPT_IRLS_COLUMNS_PER_WAVE, cpw)
            dX = torch.full((M.cols, k), 3.0, dtype=torch.float32, device="cuda")
            ctx.solve_irls(_abi.F32, loss, _dev(torch, M.p), _dev(torch, M.i), _dev(torch, M.values(np.float32)), M.cols,
                           _dev(torch, F), _dev(torch, G), dX, k, l2=1e-3, theta_row=_dev(torch, theta) if by_row else None,
                           theta_col=None if by_row else _dev(torch, theta), **opts)
            out[cpw] = dX.cpu().numpy()
        ctx.set_option(_abi.OPT_IRLS_COLUMNS_PER_WAVE, 0)
        assert np.all(np.isfinite(out[4]))
        assert np.array_equal(out[1], out[4]), float(np.abs(out[1] - out[4]).max())


@pytest.mark.parametrize("k", [4, 16, 32])
def test_irls_nb_half_update_four_columns_per_wave(env, k):
    """The four-columns-per-wavefront kernel against the oracle (same check as test_irls_nb_half_updates, fp32 rows)."""
    torch, _abi, ctx = env
    A = _nb_problem(150, 220, 4, seed=k)
    rng = np.random.default_rng(k)
    F = rng.uniform(0.05, 1.0, size=(A.rows, k)).astype(np.float32)
    F /= F.sum(axis=0, keepdims=True)
    F *= 30.0
    G = O.gram(F)
    theta = rng.uniform(2.0, 20.0, size=A.rows).astype(np.float32)
    ref = O.irls_nb(A, F, G, k, L1=0.0, L2=1e-3, theta_row=theta, theta_col=None, dtype=np.float32)
    dX = torch.full((A.cols, k), 3.0, dtype=torch.float32, device="cuda")
    ctx.set_option(_abi.OPT_IRLS_COLUMNS_PER_WAVE, 4)
    try:
        ctx.solve_irls_nb(_abi.F32, _dev(torch, A.p), _dev(torch, A.i), _dev(torch, A.values(np.float32)), A.cols, _dev(torch, F),
                          _dev(torch, G), dX, k, l1=0.0, l2=1e-3, theta_row=_dev(torch, theta), theta_col=None)
    finally:
        ctx.set_option(_abi.OPT_IRLS_COLUMNS_PER_WAVE, 0)
    X = dX.cpu().numpy()
    assert X.min() >= 0 and np.all(np.isfinite(X))
    assert np.abs(X - ref).max() / np.abs(ref).max() < 3e-2


def test_full_size_c5_whole_fit_parity_fp64():
    """BASELINE configs[4] at FULL size through the plugin boundary (rcppml_gpu_nmf_ex, fp64, loss_type = 5, per-row dispersion, the
    reference's defaults) against the CPU oracle's fit of the same matrix from the same starting factors: the NB likelihood after
    each of three outer iterations.  NB-IRLS amplifies rounding by 3-4 orders of magnitude per outer iteration in the reference's own
    arithmetic (every column starts at x = 0 with its weights at the 1e6 cap; the method-of-moments size divides by a difference of
    large sums), so the bar of an iteration is tied to what the CPU fit does to ITSELF when its starting factors move by 1e-14
    relative: iterations one and two must agree to 1e-9 / 1e-8 (measured 7e-12 / 1.5e-10), the third to within 10x the oracle's own
    sensitivity (measured 3.4e-5 against 8e-6).  bench.py --config c5 prints the same numbers (c5_parity_leg)."""
    import argparse
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    from rcppml_amd import data
    m, n, k = 10000, 200000, 32
    A, _, _ = data.simulate_nb_counts(m, n, k, density=0.02, size=5.0, seed=123)
    leg = bench.c5_parity_leg(A, m, n, k, argparse.Namespace(seed=42, cd_maxit=100), iters=3)
    dev, self_dev = leg["loss_rel_dev_by_iteration"], leg["cpu_self_dev_by_iteration"]
    print("C5 whole-fit parity: GPU vs CPU %s; CPU vs itself (1e-14 perturbation) %s; d %.2e" % (dev, self_dev, leg["d_rel_dev"]))
    assert len(dev) == 3
    assert dev[0] <= 1e-9 and dev[1] <= 1e-8, dev
    assert dev[2] <= 10.0 * max(self_dev[2], 1e-7), (dev, self_dev)
