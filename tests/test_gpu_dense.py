"""Dense-input NMF through rcppml_gpu_nmf_dense_unified_* (the reference's dense plugin boundary, bridge_nmf.hpp:537-690)
vs the oracle's restatement of fit_cpu.hpp's STANDARD (unfused) path on the same matrix -- the oracle gets the dense
matrix as a CSC that stores every entry -- plus the GEMM right-hand-side op against numpy."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _dense_problem(m, n, r, seed, zero_frac=0.3):
    rng = np.random.default_rng(seed)
    M = rng.uniform(0, 1, (m, r)) @ rng.uniform(0, 1, (r, n)) + 0.05 * rng.standard_normal((m, n))
    M = np.maximum(M, 0.0)
    M[rng.random((m, n)) < zero_frac] = 0.0
    return M


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("k,m,n", [(5, 37, 53), (16, 200, 130), (64, 300, 257), (100, 64, 90)])
def test_rhs_dense(dtype, k, m, n):
    """B = F A and B = F A^T (column-major, leading dimension k) against a float64 numpy product."""
    import torch
    from rcppml_amd import _abi
    ctx = _abi.Context(0)
    rng = np.random.default_rng(k + m)
    A = rng.standard_normal((m, n)).astype(dtype)
    dA = torch.from_numpy(np.asfortranarray(A).T.copy()).cuda()          # memory = column-major m x n
    tol = 2e-5 if dtype == np.float32 else 1e-12
    dt = _abi.F32 if dtype == np.float32 else _abi.F64
    for transposed, rows in ((0, m), (1, n)):
        F = rng.standard_normal((rows, k)).astype(dtype)                 # (rows, k) array = column-major k x rows
        out_cols = n if transposed == 0 else m
        dB = torch.zeros((out_cols, k), dtype=dA.dtype, device="cuda")
        ctx.rhs_dense(dt, dA, m, n, transposed, torch.from_numpy(F).cuda(), k, dB)
        ref = (A.astype(np.float64).T @ F.astype(np.float64)) if transposed == 0 else (A.astype(np.float64) @ F.astype(np.float64))
        assert np.abs(dB.cpu().numpy() - ref).max() <= tol * np.abs(ref).max() * np.sqrt(rows)


@pytest.mark.parametrize("entry,tol_loss,tol_fac", [("double", 1e-6, 1e-6), ("float", 5e-4, 5e-3)])
def test_dense_fit_through_plugin(entry, tol_loss, tol_fac):
    from rcppml_amd import _abi
    M = _dense_problem(70, 95, 4, seed=3)
    A = O.dense_as_csc(M)
    k = 6
    dtype = np.float64 if entry == "double" else np.float32
    W0, H0 = O.init_factors(9, k, 70, 95, np.float64)
    cases = ((0, {}, {}), (1, {}, {}),
             (0, dict(L1=(0.01, 0.02), L2=(0.02, 0.01), L21=(0.03, 0.0)), dict(L1_W=0.01, L1_H=0.02, L2_W=0.02, L2_H=0.01, L21_W=0.03)),
             (0, dict(ub=(0.0, 0.08), angular=(0.02, 0.01), norm_type=1), dict(ub_H=0.08, ortho_W=0.02, ortho_H=0.01, norm_type=1)),
             (0, dict(projective=True), dict(projective=1)))
    for solver, kw_o, kw_g in cases:
        ref = O.nmf_fit(A, W0, H0, dtype, max_iter=8, tol=0.0, solver_mode=solver, unfused=True, sort_model=True, **kw_o)
        W, H = W0.copy(), H0.copy()
        res = _abi.nmf_dense(M, k, W, H, entry=entry, max_iter=8, tol=0.0, solver_mode=solver, **kw_g)
        assert res["status"] == 0, res["error"]
        assert res["iter"] == ref.iter
        assert abs(res["loss"] - ref.loss) / abs(ref.loss) < tol_loss, (solver, kw_g)
        assert np.abs(res["d"] - ref.d).max() / np.abs(ref.d).max() < tol_fac
        assert np.abs(W - ref.W_T).max() < tol_fac and np.abs(H - ref.H).max() < tol_fac
    # the dense path is NOT the sparse path on the same numbers: nnls_batch starts iteration 0 from zero
    fused = O.nmf_fit(A, W0, H0, dtype, max_iter=1, tol=0.0, solver_mode=0)
    unfused = O.nmf_fit(A, W0, H0, dtype, max_iter=1, tol=0.0, solver_mode=0, unfused=True)
    assert abs(fused.loss - unfused.loss) > 1e-3 * abs(unfused.loss)


def test_dense_symmetric_and_rejections():
    from rcppml_amd import _abi
    rng = np.random.default_rng(2)
    F = rng.uniform(0, 1, (60, 3))
    S = F @ F.T
    A = O.dense_as_csc(S)
    k = 4
    W0, H0 = O.init_factors(5, k, 60, 60, np.float64)
    ref = O.nmf_fit(A, W0, H0, np.float64, max_iter=8, tol=0.0, symmetric=True, unfused=True)
    W, H = W0.copy(), H0.copy()
    res = _abi.nmf_dense(S, k, W, H, entry="double", max_iter=8, tol=0.0, symmetric=1)
    assert res["status"] == 0 and abs(res["loss"] - ref.loss) / abs(ref.loss) < 1e-6
    assert np.abs(W - ref.W_T).max() < 1e-6 and np.array_equal(W, H)
    M = _dense_problem(30, 40, 3, seed=1)
    for kw in (dict(loss_type=5), dict(robust_delta=1.0), dict(symmetric=1), dict(solver_mode=2), dict(ortho_H=-0.1)):
        W1, H1 = O.init_factors(1, 3, 30, 40, np.float64)
        r = _abi.nmf_dense(M, 3, W1, H1, entry="double", max_iter=2, **kw)
        assert r["status"] == -1 and r["error"], kw


def test_nmf_surface_dense_input():
    """nmf(<base matrix>) goes through the dense entry (R: a dense matrix takes the dense path); a scipy / CSC input of the
    same numbers goes through the sparse entry -- both decrease the loss, and differ (iteration-0 start)."""
    import scipy.sparse as sp
    from rcppml_amd import nmf as N
    M = _dense_problem(60, 80, 3, seed=8)
    md = N.nmf(M, 4, seed=11, maxit=10, tol=0.0, precision="fp64")
    ms = N.nmf(sp.csc_matrix(M), 4, seed=11, maxit=10, tol=0.0, precision="fp64")
    assert md.misc.get("input") == "dense" and "input" not in ms.misc
    W0 = N.r_runif(11, 60 * 4).reshape(4, 60).T.copy()
    from rcppml_amd.data import splitmix64_uniform
    H0 = splitmix64_uniform(11, 0, 4 * 80, np.float64).reshape(80, 4)
    ref = O.nmf_fit(O.dense_as_csc(M), W0, H0, np.float64, max_iter=10, tol=0.0, unfused=True)
    assert abs(md.misc["loss"] - ref.loss) / ref.loss < 1e-6 and np.abs(md.w - ref.W_T).max() < 1e-6
    assert abs(ms.misc["loss"] - md.misc["loss"]) / md.misc["loss"] < 0.2
