"""GPU parity, cross-validation path (reference nmf/fit_cv.hpp): the speckled-mask half-update and the
held-out error through the device-level C ABI, and the CV fit through the plugin's CV entry, against the oracle's
restatement (whose mask hash is pinned to the reference's rng/rng.hpp in tests/test_oracle_ref.py)."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import lowrank_csc, random_csc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    from rcppml_amd import _abi
    return torch, _abi, _abi.Context(0)


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-8), (np.float32, 2e-3)])
@pytest.mark.parametrize("k", [5, 32, 40, 72, 128])
@pytest.mark.parametrize("mask_zeros", [0, 1])
@pytest.mark.parametrize("solver", [0, 1])
def test_cv_half_updates(env, dtype, tol, k, mask_zeros, solver):
    """H side (A, F = W_T) and W side (A^T, transposed mask arguments), both solvers, zeros held out or not."""
    torch, _abi, ctx = env
    A = lowrank_csc(130, 170, 4, 0.2, seed=k)
    At = A.transpose()
    dt = _abi.F32 if dtype == np.float32 else _abi.F64
    rng = np.random.default_rng(k + mask_zeros)
    frac, cv_seed = 0.1, 77
    for (M, transposed) in ((A, 0), (At, 1)):
        F = rng.uniform(0.05, 1.0, size=(M.rows, k)).astype(dtype)
        G = O.gram(F)
        G[np.diag_indices(k)] += dtype(0.3)            # L2-like ridge keeps G - sum_test f f^T well conditioned
        X0 = rng.uniform(0.0, 0.2, size=(M.cols, k)).astype(dtype)
        ref = O.cv_half_update(M, F, G, X0, k, frac, cv_seed, mask_zeros=bool(mask_zeros), transposed=bool(transposed), L1=0.01,
                               cd_maxit=20, solver_mode=solver, dtype=dtype)
        dX = _dev(torch, X0.copy())
        ctx.solve_cv(dt, _dev(torch, M.p), _dev(torch, M.i), _dev(torch, M.values(dtype)), M.cols, M.rows, _dev(torch, F),
                     _dev(torch, G), dX, k, frac, cv_seed, mask_zeros=mask_zeros, transposed=transposed, l1=0.01, cd_maxit=20,
                     solver_mode=solver)
        X = dX.cpu().numpy()
        assert np.all(np.isfinite(X)) and X.min() >= 0
        assert np.abs(X - ref).max() / max(np.abs(ref).max(), 1e-30) < tol, (k, mask_zeros, solver, transposed)


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-11), (np.float32, 1e-4)])
@pytest.mark.parametrize("mask_zeros", [0, 1])
@pytest.mark.parametrize("k", [7, 100])
def test_cv_test_error(env, dtype, tol, mask_zeros, k):
    torch, _abi, ctx = env
    A = lowrank_csc(90, 140, 3, 0.25, seed=3)
    dt = _abi.F32 if dtype == np.float32 else _abi.F64
    rng = np.random.default_rng(5)
    W = rng.uniform(size=(A.rows, k)).astype(dtype)
    H = rng.uniform(size=(A.cols, k)).astype(dtype)
    d = rng.uniform(0.5, 2.0, size=k).astype(dtype)
    sq_ref, n_ref = O.cv_test_error(A, W, d, H, 0.2, 5, mask_zeros=bool(mask_zeros), dtype=dtype)
    out = torch.zeros((2,), dtype=torch.float64, device="cuda")
    ctx.cv_test_error(dt, _dev(torch, A.p), _dev(torch, A.i), _dev(torch, A.values(dtype)), A.cols, A.rows, _dev(torch, W),
                      _dev(torch, d), _dev(torch, H), k, 0.2, 5, mask_zeros, out)
    sq, cnt = out.cpu().numpy()
    assert int(cnt) == n_ref and n_ref > 0
    assert abs(sq - sq_ref) <= tol * abs(sq_ref)
    # the mask holds out about `fraction` of the entries (all entries, or the nonzeros with mask_zeros)
    total = A.nnz if mask_zeros else A.rows * A.cols
    assert abs(n_ref / total - 0.2) < 0.03


@pytest.mark.parametrize("precision,tol_loss,tol_fac", [("f64", 1e-7, 1e-6), ("f32", 2e-3, 5e-3)])
@pytest.mark.parametrize("solver", [1, 0])
@pytest.mark.parametrize("mask_zeros", [0, 1])
def test_cv_fit_through_plugin(precision, tol_loss, tol_fac, solver, mask_zeros):
    """rcppml_gpu_nmf_cv_* (the reference's CV plugin boundary) vs the oracle's nmf_fit_cv: same iteration count, same
    early-stopping decision, train / test / best-test losses, factors (H returned with d absorbed)."""
    from rcppml_amd import _abi
    A = lowrank_csc(80, 110, 3, 0.3, seed=2)
    k = 4
    dtype = np.float64 if precision == "f64" else np.float32
    W0, H0 = O.init_factors(5, k, A.rows, A.cols, np.float64)
    ref = O.nmf_fit_cv(A, W0, H0, dtype, max_iter=12, tol=1e-6, L1=(0.0, 0.01), L2=(0.02, 0.0), solver_mode=solver,
                       holdout_fraction=0.1, cv_seed=3, mask_zeros=bool(mask_zeros), cv_patience=5)
    W, H = W0.copy(), H0.copy()
    res = _abi.nmf_cv(A.p, A.i, A.x, A.rows, A.cols, k, W, H, entry="ex", max_iter=12, tol=1e-6, L1_H=0.01, L2_W=0.02,
                      solver_mode=solver, holdout_fraction=0.1, cv_seed=3, mask_zeros=mask_zeros, sort_model=0,
                      precision=_abi.F64 if precision == "f64" else _abi.F32, cv_patience=5)
    assert res["status"] == 0, res.get("error")
    assert res["iter"] == ref.iter and res["converged"] == ref.converged and res["best_iter"] == ref.best_iter
    assert np.allclose(res["test_history"], ref.test_history, rtol=tol_loss, atol=0)
    assert np.allclose(res["train_history"], ref.train_history, rtol=tol_loss * 10, atol=0)
    assert abs(res["best_test_loss"] - ref.best_test_loss) <= tol_loss * abs(ref.best_test_loss)
    assert np.abs(res["d"] - ref.d).max() <= tol_fac * np.abs(ref.d).max()
    assert np.abs(W - ref.W_T).max() < tol_fac and np.abs(H - ref.H).max() < tol_fac * max(1.0, np.abs(ref.H).max())


def test_cv_reference_entry_points_and_rejections():
    """The two reference symbols (51 pointers, cv_seed = 0 falls back to seed) and the refusals (out_status = -1)."""
    from rcppml_amd import _abi
    A = lowrank_csc(60, 70, 3, 0.3, seed=4)
    k = 3
    W0, H0 = O.init_factors(9, k, A.rows, A.cols, np.float64)
    ref = O.nmf_fit_cv(A, W0, H0, np.float64, max_iter=8, tol=1e-6, solver_mode=1, holdout_fraction=0.2, cv_seed=17, cv_patience=5,
                       )
    for entry, kw in (("double", dict(cv_seed=17, seed=99)), ("double", dict(cv_seed=0, seed=17))):
        W, H = W0.copy(), H0.copy()
        res = _abi.nmf_cv(A.p, A.i, A.x, A.rows, A.cols, k, W, H, entry=entry, max_iter=8, tol=1e-6, solver_mode=1,
                          holdout_fraction=0.2, **kw)
        assert res["status"] == 0 and res["iter"] == ref.iter
        assert abs(res["test_loss"] - ref.test_loss) <= 1e-7 * abs(ref.test_loss)
        assert abs(res["train_loss"] - ref.train_loss) <= 1e-6 * abs(ref.train_loss)
    W, H = W0.copy(), H0.copy()
    res32 = _abi.nmf_cv(A.p, A.i, A.x, A.rows, A.cols, k, W, H, entry="float", max_iter=8, tol=1e-6, solver_mode=1,
                        holdout_fraction=0.2, cv_seed=17)
    assert res32["status"] == 0 and abs(res32["test_loss"] - ref.test_loss) <= 5e-3 * abs(ref.test_loss)
    for kw in (dict(loss_type=1), dict(loss_type=3), dict(projective=1), dict(symmetric=1), dict(graph_W_nnz=3), dict(solver_mode=2),
               dict(holdout_fraction=0.0), dict(holdout_fraction=1.5)):
        W, H = W0.copy(), H0.copy()
        r = _abi.nmf_cv(A.p, A.i, A.x, A.rows, A.cols, k, W, H, entry="double", max_iter=2, **kw)
        assert r["status"] == -1 and r["error"], kw

@pytest.mark.parametrize("loss_type", [0, 6])
def test_cv_fit_through_plugin_at_rank_above_64(loss_type):
    """The CV entry at k = 80 (MSE: per-column Gram correction; Gamma: per-column weighted Grams) in fp64 vs the oracle."""
    from rcppml_amd import _abi
    A = _counts_csc(60, 85, 0.3, seed=6)
    k = 80
    rng = np.random.default_rng(2)
    W0 = rng.uniform(0.2, 1.0, size=(A.rows, k)); H0 = rng.uniform(0.2, 1.0, size=(A.cols, k))
    kw = dict(max_iter=3, tol=1e-12, solver_mode=0, holdout_fraction=0.15, cv_seed=5, cv_patience=5)
    ref = O.nmf_fit_cv(A, W0, H0, np.float64, L2=(0.05, 0.05), loss_type=loss_type, irls_max_iter=3, **kw)
    W, H = W0.copy(), H0.copy()
    res = _abi.nmf_cv(A.p, A.i, A.x, A.rows, A.cols, k, W, H, entry="irls_ex" if loss_type else "ex", L2_H=0.05, L2_W=0.05, sort_model=0,
                      precision=_abi.F64, **(dict(loss_type=loss_type, irls_max_iter=3) if loss_type else {}), **kw)
    assert res["status"] == 0, res.get("error")
    assert res["iter"] == ref.iter
    assert np.allclose(res["test_history"], ref.test_history, rtol=1e-6) and np.allclose(res["train_history"], ref.train_history, rtol=1e-6)
    assert np.abs(W - ref.W_T).max() < 1e-6 * max(1.0, np.abs(ref.W_T).max()) and np.abs(H - ref.H).max() < 1e-6 * max(1.0, np.abs(ref.H).max())



def test_cv_python_surface():
    """nmf(test_fraction = ...) mirrors the R call: test / train losses and histories in misc, patience-based stop."""
    from rcppml_amd import nmf as N
    from rcppml_amd.data import CSC
    A = lowrank_csc(80, 110, 3, 0.3, seed=2)
    Ap = CSC((A.rows, A.cols), A.p, A.i, A.x)
    mod = N.nmf(Ap, 4, test_fraction=0.1, seed=3, maxit=15, tol=1e-6, patience=3)
    assert mod.misc["test_fraction"] == 0.1 and np.isfinite(mod.misc["test_loss"]) and np.isfinite(mod.misc["loss"])
    assert len(mod.misc["test_loss_history"]) == mod.misc["iter"] <= 15
    assert mod.misc["best_test_loss"] <= mod.misc["test_loss_history"].min() * (1 + 1e-6)
    assert mod.w.min() >= 0 and mod.h.min() >= 0
    mz = N.nmf(Ap, 4, test_fraction=0.1, seed=3, maxit=5, mask="zeros")
    assert np.isfinite(mz.misc["test_loss"])
    mm = N.nmf(Ap, 4, test_fraction=0.1, seed=3, maxit=3, mask=(np.arange(80 * 110).reshape(80, 110) % 7 == 0).astype(float))   # round 5: runs
    assert np.isfinite(mm.misc["test_loss"])


@pytest.mark.parametrize("loss", ["gp", "gamma", "tweedie"])
def test_cv_python_surface_with_irls_losses(loss):
    """nmf(test_fraction = ..., loss = "gp" / ...): the surface forwards the loss, IRLS and dispersion settings to the CV entry and
    reproduces the oracle's fit from the same R-stream initialisation (fp64)."""
    from rcppml_amd import nmf as N
    from rcppml_amd.data import CSC
    from rcppml_amd.data import r_runif, splitmix64_uniform
    A = _counts_csc(40, 55, 0.35, seed=9)
    Ap = CSC((A.rows, A.cols), A.p, A.i, A.x)
    k, seed = 3, 11
    mod = N.nmf(Ap, k, test_fraction=0.2, seed=seed, maxit=4, tol=1e-9, loss=loss, precision="fp64", irls_max_iter=3, sort_model=False,
                theta_init=0.2)
    lt = {"gp": 4, "gamma": 6, "tweedie": 8}[loss]
    W0 = r_runif(seed, A.rows * k).reshape(k, A.rows).T.copy()
    H0 = splitmix64_uniform(seed, 0, k * A.cols, np.float64).reshape(A.cols, k)
    ref = O.nmf_fit_cv(A, W0, H0, np.float64, max_iter=4, tol=1e-9, holdout_fraction=0.2, cv_seed=seed, loss_type=lt, irls_max_iter=3,
                       gp_theta=(0.2, 5.0), solver_mode=mod.misc["solver_mode"])
    assert mod.misc["iter"] == ref.iter and mod.misc["loss_type"] == loss
    assert np.allclose(mod.misc["test_loss_history"], ref.test_history, rtol=1e-7)
    assert np.allclose(mod.misc["loss_history"], ref.train_history, rtol=1e-7)
    if loss == "gp":
        assert np.allclose(mod.misc["theta"], ref.theta, rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("precision,tol_loss,tol_fac", [("f64", 1e-8, 1e-6), ("f32", 2e-3, 5e-3)])
def test_cv_fit_with_graph_regularisation(precision, tol_loss, tol_fac):
    """apply_cv_features (variant_helpers.hpp:174-189): L2 then the graph Laplacian term on the full Gram, before the
    per-column held-out correction; through the CV entry's graph_W_* / graph_H_* slots, vs the oracle."""
    import scipy.sparse as sp
    from rcppml_amd import _abi
    A = lowrank_csc(80, 110, 3, 0.3, seed=2)
    k = 4
    dtype = np.float64 if precision == "f64" else np.float32

    def chain(dim):
        Adj = sp.diags([np.ones(dim - 1), np.ones(dim - 1)], [-1, 1], format="csc")
        L = sp.csc_matrix(sp.diags(np.asarray(Adj.sum(axis=0)).ravel()) - Adj)
        L.sort_indices()
        return O.Csc(L.shape, L.indptr, L.indices, L.data)
    LW, LH = chain(A.rows), chain(A.cols)
    W0, H0 = O.init_factors(5, k, A.rows, A.cols, np.float64)
    ref = O.nmf_fit_cv(A, W0, H0, dtype, max_iter=8, tol=1e-9, L2=(0.02, 0.0), holdout_fraction=0.1, cv_seed=3,
                       graph_W=(LW, 0.05), graph_H=(LH, 0.08))
    base = O.nmf_fit_cv(A, W0, H0, dtype, max_iter=8, tol=1e-9, L2=(0.02, 0.0), holdout_fraction=0.1, cv_seed=3)
    assert abs(ref.test_loss - base.test_loss) > 1e-4 * abs(base.test_loss)
    W, H = W0.copy(), H0.copy()
    res = _abi.nmf_cv(A.p, A.i, A.x, A.rows, A.cols, k, W, H, entry="ex", max_iter=8, tol=1e-9, L2_W=0.02, holdout_fraction=0.1,
                      cv_seed=3, sort_model=0, precision=_abi.F64 if precision == "f64" else _abi.F32,
                      graph_W=(LW.p, LW.i, LW.x, 0.05), graph_H=(LH.p, LH.i, LH.x, 0.08))
    assert res["status"] == 0, res.get("error")
    assert res["iter"] == ref.iter and res["best_iter"] == ref.best_iter
    assert np.allclose(res["test_history"], ref.test_history, rtol=tol_loss, atol=0)
    assert np.abs(W - ref.W_T).max() < tol_fac and np.abs(H - ref.H).max() < tol_fac * max(1.0, np.abs(ref.H).max())


def _counts_csc(rows, cols, dens, seed):
    """Count-valued sparse matrix (what the IRLS losses are for)."""
    A = lowrank_csc(rows, cols, 4, dens, seed=seed)
    x = np.ceil(A.x / A.x.mean() * 2.0)
    return O.Csc((A.rows, A.cols), A.p, A.i, x)


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-7), (np.float32, 5e-3)])
@pytest.mark.parametrize("loss_type", [4, 5, 6, 7, 8])
@pytest.mark.parametrize("mask_zeros", [0, 1])
@pytest.mark.parametrize("solver", [0, 1])
def test_cv_irls_half_updates(env, dtype, tol, loss_type, mask_zeros, solver):
    """The per-column weighted-Gram half-update of the IRLS CV path (reference nmf/cv_detail.hpp:101-292) on both sides, both
    solvers, zeros held out or not, with an additive feature term, against the oracle's restatement."""
    torch, _abi, ctx = env
    k = {4: 6, 5: 6, 6: 33, 7: 70, 8: 128}[loss_type]          # 70, 128: the two-features-per-lane kernels (kernels_wide.hip.h)
    A = _counts_csc(70, 90, 0.25, seed=loss_type + 10 * mask_zeros)
    At = A.transpose()
    dt = _abi.F32 if dtype == np.float32 else _abi.F64
    rng = np.random.default_rng(loss_type + solver)
    frac, cv_seed = 0.15, 31
    for (M, transposed) in ((A, 0), (At, 1)):
        F = rng.uniform(0.05, 1.0, size=(M.rows, k)).astype(dtype)
        X0 = rng.uniform(0.05, 0.4, size=(M.cols, k)).astype(dtype)
        G_add = (np.eye(k) * 0.05 + 0.01 * np.ones((k, k))).astype(dtype)
        kw = dict(mask_zeros=bool(mask_zeros), transposed=bool(transposed), L1=0.01, cd_maxit=15, solver_mode=solver, irls_max_iter=4,
                  irls_tol=1e-4, power=1.6)
        ref = O.cv_irls_half_update(M, F, X0, k, frac, cv_seed, loss_type, G_add=G_add, dtype=dtype, **kw)
        dX = _dev(torch, X0.copy())
        ctx.solve_cv_irls(dt, loss_type, _dev(torch, M.p), _dev(torch, M.i), _dev(torch, M.values(dtype)), M.cols, M.rows, _dev(torch, F),
                          _dev(torch, G_add), dX, k, frac, cv_seed, mask_zeros=mask_zeros, transposed=transposed, l1=0.01, cd_maxit=15,
                          solver_mode=solver, irls_max_iter=4, irls_tol=1e-4, loss_param=1.6)
        X = dX.cpu().numpy()
        assert np.all(np.isfinite(X)) and X.min() >= 0
        assert not np.array_equal(X, X0)
        # a column whose IRLS passes stop at a different pass (relative change right at irls_tol) may differ in fp32
        err = np.abs(X - ref).max(axis=1) / max(np.abs(ref).max(), 1e-30)
        assert np.median(err) < tol and (err < tol).mean() > (0.999 if dtype == np.float64 else 0.97), (loss_type, mask_zeros, solver, transposed, err.max())


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-11), (np.float32, 2e-4)])
@pytest.mark.parametrize("loss_type", [0, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("mask_zeros", [0, 1])
def test_cv_irls_losses(env, dtype, tol, loss_type, mask_zeros):
    """Per-element train / test losses (reference nmf/fit_cv.hpp:1377-1443): sums and counts vs the oracle."""
    torch, _abi, ctx = env
    k = 9 if loss_type != 6 else 90
    A = _counts_csc(80, 120, 0.2, seed=2)
    dt = _abi.F32 if dtype == np.float32 else _abi.F64
    rng = np.random.default_rng(loss_type)
    W = rng.uniform(0.1, 1.0, size=(A.rows, k)).astype(dtype)
    H = rng.uniform(0.1, 1.0, size=(A.cols, k)).astype(dtype)
    d = rng.uniform(0.5, 2.0, size=k).astype(dtype)
    theta = rng.uniform(0.0, 0.8, size=A.rows).astype(dtype)
    ref = O.cv_explicit_loss(A, W, d, H, 0.2, 9, loss_type, theta=theta, mask_zeros=bool(mask_zeros), power=1.4, dtype=dtype)
    out = torch.zeros((4,), dtype=torch.float64, device="cuda")
    ctx.cv_irls_loss(dt, loss_type, _dev(torch, A.p), _dev(torch, A.i), _dev(torch, A.values(dtype)), A.cols, A.rows, _dev(torch, W),
                     _dev(torch, d), _dev(torch, H), _dev(torch, theta), k, 0.2, 9, mask_zeros, 1.4, out)
    tr, ntr, te, nte = out.cpu().numpy()
    assert (int(ntr), int(nte)) == (ref[1], ref[3]) and ref[3] > 0
    assert ntr + nte == (A.nnz if mask_zeros else A.rows * A.cols)
    assert abs(tr - ref[0]) <= tol * abs(ref[0]) and abs(te - ref[2]) <= tol * abs(ref[2])


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-9), (np.float32, 5e-4)])
@pytest.mark.parametrize("mode", [2, 1])
@pytest.mark.parametrize("frac", [0.2, 0.0])
def test_cv_gp_theta_over_training_entries(env, dtype, tol, mode, frac):
    """GP theta (MM update) over the training entries only (reference nmf/fit_cv.hpp:866-961); frac = 0 = the non-CV update."""
    torch, _abi, ctx = env
    k = 5
    A = _counts_csc(60, 150, 0.3, seed=8)
    At = A.transpose()
    dt = _abi.F32 if dtype == np.float32 else _abi.F64
    rng = np.random.default_rng(mode)
    W = rng.uniform(0.1, 1.0, size=(A.rows, k)).astype(dtype)
    H = rng.uniform(0.1, 1.0, size=(A.cols, k)).astype(dtype)
    d = rng.uniform(0.5, 2.0, size=k).astype(dtype)
    th0 = np.full(A.rows, 0.1, dtype)
    ref = O.cv_gp_theta_update(A, W, d, H, th0, frac, 13, mode=mode, theta_max=5.0, dtype=dtype)
    dth = _dev(torch, th0.copy())
    ctx.cv_gp_theta_update(dt, mode, _dev(torch, At.p), _dev(torch, At.i), _dev(torch, At.values(dtype)), A.rows, A.nnz, _dev(torch, W),
                           _dev(torch, d), _dev(torch, H), A.cols, k, frac, 13, 5.0, dth)
    th = dth.cpu().numpy()
    assert not np.allclose(th, th0)
    assert np.abs(th - ref).max() <= tol * max(np.abs(ref).max(), 1e-30)



@pytest.mark.parametrize("precision,tol_loss,tol_fac", [("f64", 1e-6, 1e-5), ("f32", 5e-3, 2e-2)])
@pytest.mark.parametrize("loss_type", [4, 5, 6, 8])
@pytest.mark.parametrize("mask_zeros", [0, 1])
def test_cv_irls_fit_through_plugin(precision, tol_loss, tol_fac, loss_type, mask_zeros):
    """CV fits with the IRLS losses through the plugin's CV boundary (the build-defined entry: histories, precision, sort off) vs
    the oracle's nmf_fit_cv: iteration count, early-stopping decision, train / test losses per iteration, GP theta, factors."""
    from rcppml_amd import _abi
    A = _counts_csc(50, 70, 0.3, seed=20 + loss_type)
    k = 3
    dtype = np.float64 if precision == "f64" else np.float32
    rng = np.random.default_rng(loss_type)
    W0 = rng.uniform(0.2, 1.0, size=(A.rows, k)); H0 = rng.uniform(0.2, 1.0, size=(A.cols, k))
    solver = 1 if loss_type in (5, 8) else 0
    kw = dict(max_iter=6, tol=1e-9, solver_mode=solver, holdout_fraction=0.15, cv_seed=5, cv_patience=4)
    ref = O.nmf_fit_cv(A, W0, H0, dtype, L1=(0.0, 0.01), L2=(0.02, 0.0), mask_zeros=bool(mask_zeros), loss_type=loss_type, irls_max_iter=3,
                       irls_tol=1e-4, **kw)
    W, H = W0.copy(), H0.copy()
    res = _abi.nmf_cv(A.p, A.i, A.x, A.rows, A.cols, k, W, H, entry="irls_ex", L1_H=0.01, L2_W=0.02, mask_zeros=mask_zeros, sort_model=0,
                      precision=_abi.F64 if precision == "f64" else _abi.F32, loss_type=loss_type, irls_max_iter=3, irls_tol=1e-4, **kw)
    assert res["status"] == 0, res.get("error")
    assert res["iter"] == ref.iter and res["converged"] == ref.converged
    assert np.all(np.isfinite(res["test_history"]))
    if not (precision == "f32" or loss_type == 5):
        assert res["best_iter"] == ref.best_iter
    if precision == "f32" or loss_type == 5:
        # NB: the reference evaluates the CV weights at theta = 0 (r floored at 1e-10), so every per-column Gram is ~1e-10 f f^T plus the
        # 1e-15 ridge -- conditioned so badly that rounding differences grow by orders of magnitude per iteration in EITHER precision
        # (the fp32 oracle itself runs into NaN on some shapes; the GPU's Cholesky guards its pivots).  First iterations tightly, the
        # rest loosely; the other losses in fp64 carry the trajectory parity.
        ok = np.isfinite(ref.test_history)
        n_tight = 2 if loss_type == 5 else 3
        assert np.allclose(res["test_history"][:n_tight], ref.test_history[:n_tight], rtol=tol_loss, atol=0), (res["test_history"], ref.test_history)
        assert np.allclose(res["test_history"][ok], ref.test_history[ok], rtol=0.05, atol=0)
        assert np.allclose(res["train_history"][ok], ref.train_history[ok], rtol=0.05, atol=0)
        return
    assert np.allclose(res["test_history"], ref.test_history, rtol=tol_loss, atol=0), (res["test_history"], ref.test_history)
    assert np.allclose(res["train_history"], ref.train_history, rtol=tol_loss, atol=0)
    assert np.abs(res["d"] - ref.d).max() <= tol_fac * np.abs(ref.d).max()
    assert np.abs(W - ref.W_T).max() < tol_fac and np.abs(H - ref.H).max() < tol_fac * max(1.0, np.abs(ref.H).max())
    if loss_type == 4:
        assert np.abs(res["theta"] - ref.theta).max() <= tol_fac * max(np.abs(ref.theta).max(), 1e-30)
        assert res["theta"].max() > 0


def test_cv_irls_reference_entry_takes_config_defaults():
    """rcppml_gpu_nmf_cv_unified_double with loss_type = GP: the boundary carries no dispersion arguments, so the fit runs with the
    reference's config defaults (per-row theta, init 0.1, max 5) -- equal to the oracle with those defaults."""
    from rcppml_amd import _abi
    A = _counts_csc(40, 60, 0.3, seed=3)
    k = 3
    rng = np.random.default_rng(1)
    W0 = rng.uniform(0.2, 1.0, size=(A.rows, k)); H0 = rng.uniform(0.2, 1.0, size=(A.cols, k))
    ref = O.nmf_fit_cv(A, W0, H0, np.float64, max_iter=4, tol=1e-9, holdout_fraction=0.2, cv_seed=11, cv_patience=5, loss_type=4, sort_model=False) \
        if False else O.nmf_fit_cv(A, W0, H0, np.float64, max_iter=4, tol=1e-9, holdout_fraction=0.2, cv_seed=11, cv_patience=5, loss_type=4)
    W, H = W0.copy(), H0.copy()
    res = _abi.nmf_cv(A.p, A.i, A.x, A.rows, A.cols, k, W, H, entry="double", max_iter=4, tol=1e-9, holdout_fraction=0.2, cv_seed=11,
                      loss_type=4)
    assert res["status"] == 0 and res["iter"] == ref.iter
    assert abs(res["test_loss"] - ref.test_loss) <= 1e-6 * abs(ref.test_loss)
    assert abs(res["train_loss"] - ref.train_loss) <= 1e-6 * abs(ref.train_loss)


# ---------------------------------------------------------------------------------------------------------------------------
# Cross-validation together with a user mask (reference nmf/fit_cv.hpp:327-331, :491-501, :779-790, :1377-1443; cv_detail.hpp:433-505);
# build-defined entry rcppml_gpu_nmf_cv_masked_ex, oracle pieces pinned to brute-force numpy in tests/test_oracle_cv.py
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision,tol_loss,tol_fac", [("f64", 1e-7, 1e-6), ("f32", 2e-3, 5e-3)])
@pytest.mark.parametrize("k", [4, 40, 72])
@pytest.mark.parametrize("solver", [1, 0])
@pytest.mark.parametrize("mask_zeros", [0, 1])
def test_cv_fit_with_user_mask_through_plugin(precision, tol_loss, tol_fac, k, solver, mask_zeros):
    """MSE CV fit with a user mask (masked nonzeros AND masked zeros) vs the oracle: iteration count, early-stopping decision, both
    loss histories (explicit per-element means over the unmasked entries), factors; ranks on the 32-, 64- and wide-tile kernels."""
    from rcppml_amd import _abi
    if k > 4 and solver == 1:
        pytest.skip("Cholesky on the near-singular Grams of a rank far above the data's: no stable trajectory to compare (CD carries these ranks)")
    A = lowrank_csc(90, 120, 3, 0.3, seed=4)
    M = random_csc(90, 120, 0.12, 77)
    dtype = np.float64 if precision == "f64" else np.float32
    W0, H0 = O.init_factors(6, k, A.rows, A.cols, np.float64)
    kw = dict(max_iter=8, tol=1e-7, solver_mode=solver, holdout_fraction=0.12, cv_seed=9, cv_patience=5)
    ref = O.nmf_fit_cv(A, W0, H0, dtype, L1=(0.0, 0.01), L2=(0.02, 0.0), mask_zeros=bool(mask_zeros), mask=M, **kw)
    free = O.nmf_fit_cv(A, W0, H0, dtype, L1=(0.0, 0.01), L2=(0.02, 0.0), mask_zeros=bool(mask_zeros), **kw)
    assert abs(free.test_history[0] - ref.test_history[0]) > 1e-4 * abs(ref.test_history[0])       # the mask matters
    W, H = W0.copy(), H0.copy()
    res = _abi.nmf_cv(A.p, A.i, A.x, A.rows, A.cols, k, W, H, entry="ex", L1_H=0.01, L2_W=0.02, mask_zeros=mask_zeros, sort_model=0,
                      precision=_abi.F64 if precision == "f64" else _abi.F32, mask=(M.p, M.i), **kw)
    assert res["status"] == 0, res.get("error")
    assert res["iter"] == ref.iter and res["converged"] == ref.converged
    if precision == "f64":
        assert res["best_iter"] == ref.best_iter
    tl = tol_loss * (10 if k > 4 else 1)         # (rank above the data's: the per-column Grams are close to singular, rounding grows)
    assert np.allclose(res["test_history"], ref.test_history, rtol=tl, atol=0), (res["test_history"], ref.test_history)
    assert np.allclose(res["train_history"], ref.train_history, rtol=tl, atol=0)
    if k == 4:
        assert np.abs(res["d"] - ref.d).max() <= tol_fac * np.abs(ref.d).max()
        assert np.abs(W - ref.W_T).max() < tol_fac and np.abs(H - ref.H).max() < tol_fac * max(1.0, np.abs(ref.H).max())


@pytest.mark.parametrize("loss_type", [4, 6, 8])
@pytest.mark.parametrize("mask_zeros", [0, 1])
def test_cv_irls_fit_with_user_mask_through_plugin(loss_type, mask_zeros):
    """IRLS CV fit with a user mask, fp64: the masked entries leave the weighted Grams of both half-updates and both losses."""
    from rcppml_amd import _abi
    A = _counts_csc(50, 70, 0.3, seed=30 + loss_type)
    M = random_csc(50, 70, 0.15, 5)
    k = 3
    rng = np.random.default_rng(loss_type)
    W0 = rng.uniform(0.2, 1.0, size=(A.rows, k)); H0 = rng.uniform(0.2, 1.0, size=(A.cols, k))
    kw = dict(max_iter=5, tol=1e-9, solver_mode=1 if loss_type == 8 else 0, holdout_fraction=0.15, cv_seed=5, cv_patience=4)
    ref = O.nmf_fit_cv(A, W0, H0, np.float64, L1=(0.0, 0.01), L2=(0.02, 0.0), mask_zeros=bool(mask_zeros), loss_type=loss_type, irls_max_iter=3,
                       irls_tol=1e-4, mask=M, **kw)
    W, H = W0.copy(), H0.copy()
    res = _abi.nmf_cv(A.p, A.i, A.x, A.rows, A.cols, k, W, H, entry="irls_ex", L1_H=0.01, L2_W=0.02, mask_zeros=mask_zeros, sort_model=0,
                      precision=_abi.F64, loss_type=loss_type, irls_max_iter=3, irls_tol=1e-4, mask=(M.p, M.i), **kw)
    assert res["status"] == 0, res.get("error")
    assert res["iter"] == ref.iter and res["converged"] == ref.converged and res["best_iter"] == ref.best_iter
    assert np.allclose(res["test_history"], ref.test_history, rtol=1e-6, atol=0), (res["test_history"], ref.test_history)
    assert np.allclose(res["train_history"], ref.train_history, rtol=1e-6, atol=0)
    assert np.abs(res["d"] - ref.d).max() <= 1e-5 * np.abs(ref.d).max()
    assert np.abs(W - ref.W_T).max() < 1e-5 and np.abs(H - ref.H).max() < 1e-5 * max(1.0, np.abs(ref.H).max())
    if loss_type == 4:
        assert np.abs(res["theta"] - ref.theta).max() <= 1e-5 * max(np.abs(ref.theta).max(), 1e-30)


def test_cv_user_mask_edge_cases_and_surface():
    """An empty mask is the unmasked fit bit for bit (same kernels route aside: the generic solve kernel is what runs either way at
    k = 5, fp64 uses the MFMA form without a mask -> compared to rounding); a mask that covers a whole column leaves that column at
    alone with an empty right-hand side; malformed masks are refused; nmf(mask = <matrix>, test_fraction > 0) reaches
    the entry; the context's mask is cleared after the fit (the next unmasked fit is the unmasked fit)."""
    import scipy.sparse as sp
    from rcppml_amd import _abi, nmf as N
    A = lowrank_csc(60, 45, 3, 0.35, seed=12)
    k = 5
    W0, H0 = O.init_factors(3, k, A.rows, A.cols, np.float64)
    kw = dict(max_iter=4, tol=0.0, holdout_fraction=0.2, cv_seed=2, cv_patience=0, sort_model=0, precision=_abi.F64)
    W1, H1 = W0.copy(), H0.copy()
    plain = _abi.nmf_cv(A.p, A.i, A.x, A.rows, A.cols, k, W1, H1, entry="ex", **kw)
    W2, H2 = W0.copy(), H0.copy()
    empty = _abi.nmf_cv(A.p, A.i, A.x, A.rows, A.cols, k, W2, H2, entry="ex", mask=(np.zeros(A.cols + 1, np.int32), np.zeros(0, np.int32)), **kw)
    assert plain["status"] == 0 and empty["status"] == 0
    assert np.allclose(W1, W2, rtol=0, atol=1e-10) and np.allclose(H1, H2, rtol=0, atol=1e-10)
    assert np.allclose(plain["test_history"], empty["test_history"], rtol=1e-9, atol=0)
    # every nonzero of column 7 masked: its right-hand side is empty, its Gram loses those rows
    nz7 = A.i[A.p[7]:A.p[8]].astype(np.int32)
    mp = np.zeros(A.cols + 1, np.int32); mp[8:] = nz7.shape[0]
    Mcol = O.Csc((A.rows, A.cols), mp, nz7, np.ones(nz7.shape[0]))
    ref = O.nmf_fit_cv(A, W0, H0, np.float64, max_iter=4, tol=0.0, holdout_fraction=0.2, cv_seed=2, cv_patience=0, mask=Mcol)
    W3, H3 = W0.copy(), H0.copy()
    full = _abi.nmf_cv(A.p, A.i, A.x, A.rows, A.cols, k, W3, H3, entry="ex", mask=(mp, nz7), **kw)
    assert full["status"] == 0 and np.allclose(full["test_history"], ref.test_history, rtol=1e-7, atol=0)
    assert np.abs(H3 - ref.H).max() < 1e-6 and np.abs(H3[7]).max() < 0.2 * np.abs(H1[7]).max()      # nothing pulls the column up
    # malformed: descending rows, row out of range
    bad = _abi.nmf_cv(A.p, A.i, A.x, A.rows, A.cols, k, W0.copy(), H0.copy(), entry="ex",
                      mask=(np.r_[0, 2, np.full(A.cols - 1, 2)].astype(np.int32), np.array([5, 3], np.int32)), **kw)
    assert bad["status"] == -1 and "mask" in bad["error"]
    bad = _abi.nmf_cv(A.p, A.i, A.x, A.rows, A.cols, k, W0.copy(), H0.copy(), entry="ex",
                      mask=(np.r_[0, 1, np.full(A.cols - 1, 1)].astype(np.int32), np.array([A.rows], np.int32)), **kw)
    assert bad["status"] == -1
    # the next fit without a mask is the plain one again
    W4, H4 = W0.copy(), H0.copy()
    again = _abi.nmf_cv(A.p, A.i, A.x, A.rows, A.cols, k, W4, H4, entry="ex", **kw)
    assert np.array_equal(W4, W1) and np.array_equal(H4, H1) and np.array_equal(again["test_history"], plain["test_history"])
    # R surface
    Msp = sp.random(A.rows, A.cols, density=0.1, format="csc", random_state=3)
    Asp = sp.csc_matrix((A.x, A.i, A.p), shape=(A.rows, A.cols))
    mod = N.nmf(Asp, 3, test_fraction=0.2, mask=Msp, seed=5, maxit=4, tol=0.0, precision="fp64")
    nomask = N.nmf(Asp, 3, test_fraction=0.2, seed=5, maxit=4, tol=0.0, precision="fp64")
    assert np.isfinite(mod.misc["test_loss"]) and mod.misc["test_loss"] != nomask.misc["test_loss"]
