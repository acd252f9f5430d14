"""GPU parity tests, kernel level: each device-level op of include/rcppml_gpu.h (called through the
C-ABI via rcppml_amd._abi) against the CPU oracle on the same seeded inputs.

Tolerances (stated per test): fp64 differences come only from FMA contraction and summation order
(the gather splits a column's nonzeros over lane groups; reductions are tree-shaped) -> <= 1e-11
relative; fp32 the same effects at fp32 epsilon -> <= 5e-5 relative.  CD results are compared after
the same number of sweeps with the same early-exit rule, so they inherit those bounds times a small
amplification; integer outputs (none here) would be exact.
"""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.util import load_fixture, lowrank_csc, random_csc, rel_err

pytestmark = pytest.mark.gpu

TOL = {np.float32: 5e-5, np.float64: 1e-11}


@pytest.fixture(scope="module")
def env():
    import torch
    from rcppml_amd import _abi
    ctx = _abi.Context(0)
    return torch, _abi, ctx


def _dt(_abi, dtype):
    return _abi.F32 if dtype == np.float32 else _abi.F64


def _tt(torch, dtype):
    return torch.float32 if dtype == np.float32 else torch.float64


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _csc_dev(torch, A, dtype):
    return _dev(torch, A.p), _dev(torch, A.i), _dev(torch, A.values(dtype))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("k,r", [(10, 183), (16, 64), (32, 1000), (64, 5000), (64, 7), (100, 333), (128, 2049)])
def test_gram(env, dtype, k, r):
    torch, _abi, ctx = env
    rng = np.random.default_rng(k * 1000 + r)
    # asymmetric, sign-varying input catches row/col mapping errors of the MFMA C/D layout
    F = (rng.standard_normal((r, k)) * (1 + np.arange(k))[None, :]).astype(dtype)
    G_ref = O.gram(F)
    dF = _dev(torch, F)
    dG = torch.empty((k, k), dtype=_tt(torch, dtype), device="cuda")
    ctx.gram(_dt(_abi, dtype), dF, k, r, 1e-15, 0.0, dG)
    G = dG.cpu().numpy()
    assert np.array_equal(G, G.T), "Gram must be bitwise symmetric"
    assert rel_err(G, G_ref) < TOL[dtype] * 4
    # eps / L2 on the diagonal
    ctx.gram(_dt(_abi, dtype), dF, k, r, 1e-15, 0.5, dG)
    G2 = dG.cpu().numpy()
    assert np.all(np.abs(np.diag(G2).astype(np.float64) - np.diag(G) - 0.5) <= 2 * np.spacing(np.abs(np.diag(G2))))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("k", [1, 3, 10, 16, 32, 48, 64, 128])
def test_rhs(env, dtype, k):
    torch, _abi, ctx = env
    A = random_csc(300, 257, 0.08, seed=k)
    # empty and heavy columns
    F = np.random.default_rng(k).standard_normal((A.rows, k)).astype(dtype)
    B_ref = O.rhs(A, F, dtype)
    dp, di, dx = _csc_dev(torch, A, dtype)
    dB = torch.full((A.cols, k), 7.0, dtype=_tt(torch, dtype), device="cuda")
    ctx.rhs(_dt(_abi, dtype), dp, di, dx, A.cols, _dev(torch, F), k, dB)
    assert rel_err(dB.cpu().numpy(), B_ref) < TOL[dtype]


def test_rhs_empty_columns_and_fixture(env):
    torch, _abi, ctx = env
    A = load_fixture("movielens")
    At = A.transpose()          # many short/empty columns
    k = 32
    F = np.random.default_rng(0).uniform(size=(At.rows, k))
    B_ref = O.rhs(At, F, np.float64)
    dp, di, dx = _csc_dev(torch, At, np.float64)
    dB = torch.empty((At.cols, k), dtype=torch.float64, device="cuda")
    ctx.rhs(_abi.F64, dp, di, dx, At.cols, _dev(torch, F), k, dB)
    assert rel_err(dB.cpu().numpy(), B_ref) < 1e-12
    empties = np.nonzero(np.diff(At.p) == 0)[0]
    if len(empties):
        assert np.all(dB.cpu().numpy()[empties] == 0)


def _cd_problem(k, n, dtype, seed):
    rng = np.random.default_rng(seed)
    Fm = rng.uniform(size=(4 * k + 5, k))
    G = (Fm.T @ Fm).astype(dtype)
    G[np.diag_indices(k)] += dtype(1e-15)
    B = (rng.standard_normal((n, k)) * 3 + 1).astype(dtype)
    X0 = rng.uniform(size=(n, k)).astype(dtype)
    return G, B, X0


VARIANTS = ["lane", "wave", "group", "mfma", "mfma16", "lmf"]     # mfma: fp32 k <= 64 (falls back to group otherwise); lmf: fp32 k <= 64, plain non-negative steps (AUTO otherwise)


def _var(_abi, name):
    return dict(lane=_abi.CD_LANE, wave=_abi.CD_WAVE, group=_abi.CD_GROUP, mfma=_abi.CD_MFMA, mfma16=_abi.CD_MFMA16, lmf=_abi.CD_LMF)[name]


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("k", [2, 10, 16, 32, 64, 80, 100, 128])
def test_cd_cold(env, dtype, k, variant):
    """nnls_batch cold start (X = 0), reference nnls_batch.hpp:150-225, incl. early exit on cd_tol."""
    torch, _abi, ctx = env
    n = 131
    G, B, _ = _cd_problem(k, n, dtype, k)
    X_ref = O.nnls_batch(G, B, maxit=100, tol=1e-8)
    dX = torch.full((n, k), 5.0, dtype=_tt(torch, dtype), device="cuda")
    ctx.solve_cd(_dt(_abi, dtype), _dev(torch, G), _dev(torch, B), dX, k, n, zero_init=1, maxit=100, tol=1e-8,
                 variant=_var(_abi, variant))
    X = dX.cpu().numpy()
    assert X.min() >= 0
    scale = np.abs(X_ref).max()
    assert np.abs(X - X_ref).max() / scale < (2e-4 if dtype == np.float32 else 1e-9)


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_cd_variants_options(env, dtype, variant):
    """L1 before the solve, warm start (b -= G x), the iteration-0 quirk (start from X without correction),
    L1 inside CD (nnls()/predict() semantics), upper bounds, nonneg=False; fixed sweep count (tol=0)."""
    torch, _abi, ctx = env
    k, n = 24, 77
    G, B, X0 = _cd_problem(k, n, dtype, 99)
    var = _var(_abi, variant)
    dt = _dt(_abi, dtype)
    tol = 3e-4 if dtype == np.float32 else 1e-9

    def run(**kw):
        dX = _dev(torch, X0.copy())
        ctx.solve_cd(dt, _dev(torch, G), _dev(torch, B), dX, k, n, variant=var, **kw)
        return dX.cpu().numpy()

    def ref(l1_pre=0.0, warm=False, zero=False, l1_cd=0.0, nonneg=True, maxit=7, tolr=0.0, ub=0.0):
        out = np.empty_like(X0)
        for j in range(n):
            b = B[j].copy()
            if l1_pre > 0:
                b -= dtype(l1_pre)
            x = np.zeros(k, dtype) if zero else X0[j].copy()
            if warm:
                b = b - G @ x
            x, _, _ = O.cd_col(G, b, x, L1=l1_cd, nonneg=nonneg, maxit=maxit, ub=ub, tol=tolr)
            out[j] = x
        return out

    s = np.abs(ref(warm=True)).max()
    assert np.abs(run(warm=1, maxit=7, tol=0.0) - ref(warm=True)).max() / s < tol
    assert np.abs(run(warm=0, maxit=7, tol=0.0) - ref()).max() / s < tol          # iteration-0 quirk
    assert np.abs(run(l1_pre=0.7, warm=1, maxit=7, tol=0.0) - ref(l1_pre=0.7, warm=True)).max() / s < tol
    assert np.abs(run(zero_init=1, l1_cd=0.05, maxit=7, tol=0.0) - ref(zero=True, l1_cd=0.05)).max() / s < tol
    assert np.abs(run(zero_init=1, nonneg=0, maxit=7, tol=0.0) - ref(zero=True, nonneg=False)).max() / s < tol * 10
    r_ub = run(zero_init=1, ub_cd=0.05, maxit=7, tol=0.0)
    assert r_ub.max() <= 0.05 + 1e-7
    assert np.abs(r_ub - ref(zero=True, ub=0.05)).max() / s < tol
    r_post = run(zero_init=1, ub_post=0.02, maxit=7, tol=0.0)
    assert np.abs(r_post - np.minimum(ref(zero=True), dtype(0.02))).max() / s < tol


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_cd_lane_equals_wave(env, dtype):
    """Both mappings execute the reference's sequential sweep exactly (same fma chain, same order):
    they must agree to the last bit."""
    torch, _abi, ctx = env
    k, n = 32, 200
    G, B, X0 = _cd_problem(k, n, dtype, 5)
    outs = []
    for var in (_abi.CD_LANE, _abi.CD_WAVE, _abi.CD_GROUP, _abi.CD_MFMA, _abi.CD_MFMA16, _abi.CD_LMF):
        dX = _dev(torch, X0.copy())
        ctx.solve_cd(_dt(_abi, dtype), _dev(torch, G), _dev(torch, B), dX, k, n, warm=1, maxit=30, tol=1e-8, variant=var)
        outs.append(dX.cpu().numpy())
    if dtype == np.float64:
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
        # the fp64 MFMA kernel multiplies by 1/G_ii and refines a hardware reciprocal in the tolerance term
        assert np.abs(outs[0] - outs[3]).max() < 1e-11 * np.abs(outs[0]).max()
        assert np.abs(outs[0] - outs[4]).max() < 1e-11 * np.abs(outs[0]).max()
    else:  # fp32 variants use rcp for the tolerance term / reassociate it (mfma): same iterates up to the exit sweep
        for o in outs[1:]:
            assert np.abs(outs[0] - o).max() < 1e-5 * np.abs(outs[0]).max()


def test_cd_known_answer(env):
    """reference tests/cpp/test_nnls.cpp:65-84: G=[[2,1],[1,2]] (+1e-10 I), b=[3,3] -> x=[1,1] (1e-4)."""
    torch, _abi, ctx = env
    G = np.array([[2.0, 1.0], [1.0, 2.0]]) + 1e-10 * np.eye(2)
    B = np.array([[3.0, 3.0]])
    for var in (_abi.CD_LANE, _abi.CD_WAVE, _abi.CD_GROUP, _abi.CD_MFMA):
        dX = torch.zeros((1, 2), dtype=torch.float64, device="cuda")
        ctx.solve_cd(_abi.F64, _dev(torch, G), _dev(torch, B), dX, 2, 1, zero_init=1, maxit=100, tol=1e-8, variant=var)
        assert np.allclose(dX.cpu().numpy(), [[1.0, 1.0]], atol=1e-4)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("k", [3, 10, 16, 32, 64])
def test_chol_clip(env, dtype, k):
    """fused_rhs_cholesky_sparse solve part (reference fused_nnls.hpp:185-219) vs the oracle LLT restatement."""
    torch, _abi, ctx = env
    n = 97
    G, B, _ = _cd_problem(k, n, dtype, 3 * k)
    G = G + np.eye(k, dtype=dtype) * dtype(0.5)     # well conditioned
    X_ref = O.chol_clip_batch(G, B - dtype(0.1))
    dX = torch.empty((n, k), dtype=_tt(torch, dtype), device="cuda")
    ctx.solve_chol(_dt(_abi, dtype), _dev(torch, G), _dev(torch, B), dX, k, n, l1_pre=0.1, nonneg=1)
    X = dX.cpu().numpy()
    assert X.min() >= 0
    assert np.abs(X - X_ref).max() / np.abs(X_ref).max() < (5e-4 if dtype == np.float32 else 1e-10)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("norm_type", [0, 1, 2])
@pytest.mark.parametrize("k,c", [(10, 1183), (64, 3000), (7, 5)])
def test_scaling(env, dtype, norm_type, k, c):
    torch, _abi, ctx = env
    X = np.random.default_rng(c).uniform(size=(c, k)).astype(dtype)
    X[:, 0] = 0   # dead factor -> d = 1e-15
    Xr, dr = O.extract_scaling(X, norm_type)
    dX = _dev(torch, X)
    sums = torch.empty(k, dtype=_tt(torch, dtype), device="cuda")
    d = torch.empty(k, dtype=_tt(torch, dtype), device="cuda")
    ctx.row_norms(_dt(_abi, dtype), dX, k, c, norm_type, sums)
    ctx.apply_scaling(_dt(_abi, dtype), dX, k, c, norm_type, sums, d)
    assert rel_err(d.cpu().numpy(), dr) < TOL[dtype]
    assert rel_err(dX.cpu().numpy(), Xr) < TOL[dtype] * 2


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_loss_mse_terms(env, dtype):
    """trace_AtA + cross term + recon (reference fit_cpu.hpp:1729-1753) against the oracle's explicit pass."""
    torch, _abi, ctx = env
    A = lowrank_csc(150, 220, 6, 0.1, seed=1)
    At = A.transpose()
    k = 12
    rng = np.random.default_rng(2)
    W_T = rng.uniform(size=(A.rows, k)).astype(dtype)
    H = rng.uniform(size=(A.cols, k)).astype(dtype)
    d = rng.uniform(0.5, 2.0, size=k).astype(dtype)
    cross_ref = O.loss_cross(At, W_T.astype(np.float64), H.astype(np.float64), d.astype(np.float64))
    Gs = O.gram(H.astype(np.float64))
    Gw = O.gram(W_T.astype(np.float64))
    recon_ref = float(np.sum(np.outer(d, d).astype(np.float64) * Gw * Gs))
    tr_ref = O.trace_AtA(A)
    dt = _dt(_abi, dtype)
    tt = _tt(torch, dtype)
    tp, ti, tx = _csc_dev(torch, At, dtype)
    dH, dW, dd = _dev(torch, H), _dev(torch, W_T), _dev(torch, d)
    dBw = torch.empty((A.rows, k), dtype=tt, device="cuda")
    ctx.rhs(dt, tp, ti, tx, At.cols, dH, k, dBw)
    dGs = torch.empty((k, k), dtype=tt, device="cuda")
    dGw = torch.empty((k, k), dtype=tt, device="cuda")
    ctx.gram(dt, dH, k, A.cols, 1e-15, 0.0, dGs)
    ctx.gram(dt, dW, k, A.rows, 1e-15, 0.0, dGw)
    tr = torch.empty(1, dtype=torch.float64, device="cuda")
    ctx.sumsq(dt, _dev(torch, A.values(dtype)), A.nnz, tr)
    out = torch.empty(4, dtype=torch.float64, device="cuda")
    ctx.loss_mse(dt, tr, dd, dW, dBw, k, A.rows, dGw, dGs, out)
    o = out.cpu().numpy()
    tol = 2e-5 if dtype == np.float32 else 1e-12
    assert abs(tr.item() - tr_ref) / tr_ref < tol
    assert abs(o[1] - cross_ref) / abs(cross_ref) < tol
    assert abs(o[2] - recon_ref) / abs(recon_ref) < tol
    assert abs(o[0] - (tr_ref - 2 * cross_ref + recon_ref)) / abs(tr_ref) < tol


def test_ctx_stats_counts_column_sweeps(env):
    """rcppml_hip_ctx_stats: the CD kernels count the sweeps they run (what cd_nnls_col_fixed returns, summed over
    columns); must agree with the per-column sweeps_out array for both AUTO kernels (MFMA fp32, lane-group fp64)."""
    torch, _abi, ctx = env
    k, n = 24, 300
    for dtype in (np.float32, np.float64):
        G, B, _ = _cd_problem(k, n, dtype, 3)
        dX = torch.zeros((n, k), dtype=_tt(torch, dtype), device="cuda")
        sw = torch.zeros((n,), dtype=torch.int32, device="cuda")
        ctx.stats(reset=True)
        ctx.solve_cd(_dt(_abi, dtype), _dev(torch, G), _dev(torch, B), dX, k, n, zero_init=1, maxit=100, tol=1e-8,
                     sweeps_out=sw)
        st = ctx.stats()
        assert st["cd_columns"] == n
        assert st["cd_column_sweeps"] == int(sw.sum().item()) and st["cd_column_sweeps"] >= n


def test_transpose_csc_and_cast(env):
    """rcppml_hip_transpose_csc (stable radix sort by row): identical to the host transpose -- integer arrays and the
    moved values bit for bit, column indices ascending inside every row; pattern-only (mask) form; empty matrix."""
    torch, _abi, ctx = env
    for (m, n, dens, seed) in ((300, 257, 0.08, 1), (5, 900, 0.4, 2), (1000, 3, 0.5, 3), (64, 64, 0.0, 4)):
        A = random_csc(m, n, dens, seed=seed)
        At = A.transpose()
        for dtype, dt, tt in ((np.float32, _abi.F32, torch.float32), (np.float64, _abi.F64, torch.float64)):
            dp, di, dx = _csc_dev(torch, A, dtype)
            tp = torch.full((m + 1,), -1, dtype=torch.int32, device="cuda")
            ti = torch.full((max(A.nnz, 1),), -1, dtype=torch.int32, device="cuda")
            tx = torch.zeros((max(A.nnz, 1),), dtype=tt, device="cuda")
            ctx.transpose_csc(dt, m, n, dp, di, dx, tp, ti, tx)
            assert np.array_equal(tp.cpu().numpy(), At.p.astype(np.int32))
            assert np.array_equal(ti.cpu().numpy()[:A.nnz], At.i.astype(np.int32))
            assert np.array_equal(tx.cpu().numpy()[:A.nnz], At.values(dtype))
        ti2 = torch.full((max(A.nnz, 1),), -1, dtype=torch.int32, device="cuda")
        ctx.transpose_csc(_abi.F64, m, n, dp, di, None, tp, ti2, None)
        assert np.array_equal(ti2.cpu().numpy()[:A.nnz], At.i.astype(np.int32))
    src = torch.from_numpy(np.random.default_rng(0).standard_normal(1000)).cuda()
    dst = torch.empty((1000,), dtype=torch.float32, device="cuda")
    ctx.cast(_abi.F64, src, _abi.F32, dst, 1000)
    assert np.array_equal(dst.cpu().numpy(), src.cpu().numpy().astype(np.float32))
    back = torch.empty((1000,), dtype=torch.float64, device="cuda")
    ctx.cast(_abi.F32, dst, _abi.F64, back, 1000)
    assert np.array_equal(back.cpu().numpy(), dst.cpu().numpy().astype(np.float64))


def test_transpose_csc_skewed_and_tall(env):
    """The transpose's chunks are cut by nonzeros on column boundaries (round 5): matrices whose nonzeros sit in a few columns, with
    empty columns between them and at both ends, more columns than chunks (512), and more rows than one LDS pass holds (32 768)."""
    import scipy.sparse as sp
    torch, _abi, ctx = env
    from oracle.oracle import Csc
    rs = np.random.default_rng(12)
    cases = []
    # (a) 3 000 columns, 40 of them dense-ish, the rest empty or with one entry; (b) tall: 70 000 rows, 900 columns, skewed
    for (m, n, heavy, hd, ld) in ((500, 3000, 40, 0.6, 0.0005), (70000, 900, 12, 0.05, 0.00005)):
        dens = np.full(n, ld)
        dens[rs.choice(n, size=heavy, replace=False)] = hd
        dens[:5] = 0.0; dens[-7:] = 0.0                          # empty columns at both ends
        cols = []
        for j in range(n):
            cnt = rs.binomial(m, dens[j])
            rows = np.sort(rs.choice(m, size=cnt, replace=False))
            cols.append(sp.csc_matrix((rs.uniform(0.1, 1.0, cnt), (rows, np.zeros(cnt, int))), shape=(m, 1)))
        M = sp.hstack(cols, format="csc"); M.sort_indices()
        cases.append(Csc((m, n), M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data.astype(np.float64)))
    for A in cases:
        m, n = A.rows, A.cols
        At = A.transpose()
        dp, di, dx = _csc_dev(torch, A, np.float64)
        tp = torch.full((m + 1,), -1, dtype=torch.int32, device="cuda")
        ti = torch.full((max(A.nnz, 1),), -1, dtype=torch.int32, device="cuda")
        tx = torch.zeros((max(A.nnz, 1),), dtype=torch.float64, device="cuda")
        ctx.transpose_csc(_abi.F64, m, n, dp, di, dx, tp, ti, tx)
        assert np.array_equal(tp.cpu().numpy(), At.p.astype(np.int32))
        assert np.array_equal(ti.cpu().numpy()[:A.nnz], At.i.astype(np.int32))
        assert np.array_equal(tx.cpu().numpy()[:A.nnz], At.values(np.float64))


@pytest.mark.parametrize("m,n,density", [(40000, 700, 0.002), (65536, 300, 0.004), (65537, 300, 0.004), (300000, 2000, 0.001),
                                         (1 << 20, 64, 0.0005), (2500000, 40, 0.0004)])
def test_transpose_csc_tall_radix_passes(env, m, n, density):
    """Inputs with more than 32 768 rows take the stable LSD radix sort of the nonzero positions by row index (8 bits per pass: 2
    passes up to 65 536 rows, 3 above; ADVICE r4: the LDS-counter form re-read every chunk once per 32 768 rows): row pointers, column
    indices (ascending inside every row, as the reference's `A.transpose()` leaves them, nmf/fit_cpu.hpp:251-253) and values equal
    scipy's transpose; also through the two-step entry the plugin uses (sort on the row indices, gather when the values arrive)."""
    import scipy.sparse as sp
    torch, _abi, ctx = env
    rs = np.random.default_rng(m % 1000 + n)
    M = sp.random(m, n, density=density, format="csc", random_state=rs, dtype=np.float64)
    M.sort_indices()
    T = sp.csc_matrix(M.T)
    T.sort_indices()
    nnz = M.nnz
    dp = torch.from_numpy(M.indptr.astype(np.int32)).cuda(); di = torch.from_numpy(M.indices.astype(np.int32)).cuda()
    dx = torch.from_numpy(M.data).cuda()
    tp = torch.full((m + 1,), -1, dtype=torch.int32, device="cuda")
    ti = torch.full((nnz,), -1, dtype=torch.int32, device="cuda")
    tx = torch.zeros((nnz,), dtype=torch.float64, device="cuda")
    ctx.transpose_csc(_abi.F64, m, n, dp, di, dx, tp, ti, tx)
    assert np.array_equal(tp.cpu().numpy(), T.indptr.astype(np.int32))
    assert np.array_equal(ti.cpu().numpy(), T.indices.astype(np.int32))
    assert np.array_equal(tx.cpu().numpy(), T.data)
    lib = _abi.lib()
    import ctypes as C
    tp2 = torch.full((m + 1,), -1, dtype=torch.int32, device="cuda"); pos = torch.full((nnz,), -1, dtype=torch.int32, device="cuda")
    ti2 = torch.full((nnz,), -1, dtype=torch.int32, device="cuda"); tx2 = torch.zeros((nnz,), dtype=torch.float32, device="cuda")
    x32 = dx.float()
    vp = lambda t: C.c_void_p(t.data_ptr())
    assert lib.rcppml_hip_transpose_csc_sort(ctx._h, C.c_int(m), C.c_int(n), C.c_int64(nnz), vp(dp), vp(di), vp(tp2), vp(pos)) == 0
    assert lib.rcppml_hip_transpose_csc_gather(ctx._h, C.c_int(_abi.F32), C.c_int(n), C.c_int64(nnz), vp(dp), vp(pos), vp(x32), vp(ti2), vp(tx2)) == 0
    ctx.sync()
    assert torch.equal(tp2, tp) and torch.equal(ti2, ti) and np.array_equal(tx2.cpu().numpy(), T.data.astype(np.float32))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("k,dim", [(3, 37), (10, 500), (16, 64), (33, 1200), (64, 3001)])
def test_apply_graph_reg(env, dtype, k, dim):
    """rcppml_hip_apply_graph_reg: G += lambda (X L) X^T (features/graph_reg.hpp:52-66) with a non-symmetric sparse L
    (catches a transposed product), against a float64 dense evaluation of the same expression."""
    torch, _abi, ctx = env
    rng = np.random.default_rng(k * 7 + dim)
    L = random_csc(dim, dim, min(1.0, 6.0 / dim), seed=dim + k)
    L.x[:] = L.x * rng.choice([-1.0, 1.0], size=L.x.shape[0])
    X = rng.uniform(0, 1, (dim, k)).astype(dtype)               # (dim, k) = column-major k x dim
    G0 = rng.standard_normal((k, k)).astype(dtype)
    lam = 0.37
    Ld = np.zeros((dim, dim))
    for j in range(dim):
        Ld[L.i[L.p[j]:L.p[j + 1]], j] += L.x[L.p[j]:L.p[j + 1]].astype(dtype).astype(np.float64)
    Xd = X.astype(np.float64).T                                   # k x dim
    upd = lam * (Xd @ Ld) @ Xd.T                                  # element (a, b)
    dG = _dev(torch, G0)
    lp, li, lx = _csc_dev(torch, L, dtype)
    ctx.apply_graph_reg(_dt(_abi, dtype), dG, lp, li, lx, _dev(torch, X), k, dim, lam)
    got = dG.cpu().numpy().astype(np.float64) - G0.astype(np.float64)   # memory [b*k + a] = column-major (a, b)
    assert rel_err(got.T, upd) < TOL[dtype] * 20
    # lambda = 0 leaves G untouched
    dG2 = _dev(torch, G0)
    ctx.apply_graph_reg(_dt(_abi, dtype), dG2, lp, li, lx, _dev(torch, X), k, dim, 0.0)
    assert np.array_equal(dG2.cpu().numpy(), G0)


@pytest.mark.parametrize("n", [1, 127, 2048, 2049, 5000, 40001])
def test_order_columns_permutation_and_order_invariance(env, n):
    """rcppml_hip_order_columns: the work order is a permutation of the columns, longest-first up to the serpentine
    layout (every group of 16 x 128 slots holds exactly the columns the descending sort puts there), and the CD kernels
    (32- and 16-column MFMA tiles) return bit-identical solutions with and without it -- columns are independent
    (reference nnls_batch.hpp:70-132 solves them one by one).  Equal keys keep their column order (stable sort): the
    layout is a function of the sweep counts alone."""
    torch, _abi, ctx = env
    rs = np.random.default_rng(n)
    sw = rs.integers(0, 140, size=n).astype(np.int32)            # counts above 127 share the last bin
    d_sw = _dev(torch, sw)
    d_order = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    ctx.order_columns(d_sw, n, d_order)
    order = d_order.cpu().numpy()
    assert np.array_equal(np.sort(order), np.arange(n))
    key = np.minimum(sw, 127)
    want = np.sort(key)[::-1]
    got = key[order]
    for g0 in range(0, n, 2048):
        assert np.array_equal(np.sort(got[g0:g0 + 2048]), np.sort(want[g0:g0 + 2048])), g0
    assert np.all(np.diff(got[:128]) <= 0)                        # inside a block: still descending
    # the sort is stable (equal keys keep their ascending column index), so the layout is a function of `sweeps` alone: the
    # stable descending sort, then every second FULL group of 16 blocks x 128 slots with its blocks in reverse
    pos = np.arange(n)
    blk, grp = pos >> 7, pos >> 11
    rev = ((grp & 1) == 1) & ((grp + 1) * 16 <= (n >> 7))
    dest = np.where(rev, (((grp << 4) + 15 - (blk & 15)) << 7) | (pos & 127), pos)
    expect = np.empty(n, np.int64)
    expect[dest] = np.argsort(-key.astype(np.int64), kind="stable")
    assert np.array_equal(order, expect)
    k = 24
    Fm = rs.uniform(size=(4 * k, k))
    G = (Fm.T @ Fm).astype(np.float32)
    B = (rs.standard_normal((n, k)) * 2 + 1).astype(np.float32)
    for variant in (_abi.CD_MFMA, _abi.CD_MFMA16, _abi.CD_LMF):
        outs = []
        for use in (False, True):
            dX = torch.zeros((n, k), dtype=torch.float32, device="cuda")
            ctx.solve_cd(_abi.F32, _dev(torch, G), _dev(torch, B), dX, k, n, zero_init=1, maxit=30, tol=1e-6, variant=variant,
                           col_order=d_order if use else None)
            outs.append(dX.cpu().numpy())
        assert np.array_equal(outs[0], outs[1]), variant



@pytest.mark.parametrize("lg", [1, 2, 4])
@pytest.mark.parametrize("k", [7, 20, 32, 33, 48, 64])
def test_cd_lmf_geometries_and_refill(env, lg, k):
    """The persistent lane = column kernel (kernels_cd_lmf.hip.h) in every geometry: 1, 2, 4 lane groups per column, both
    paddings (32 / 64), ONE resident wave per SIMD forced so that most columns arrive by refill through the ticket queue.
    With a fixed sweep count its residual chains are the reference's sequential single-rounded fma chains, i.e. the lane-group
    VALU kernel's: bit-identical results (the 32-column MFMA kernel's v_mfma_f32_32x32x2 differs from that chain in the last
    bit); with the early exit on, sweep counts and iterates against the oracle's cd_nnls_col_fixed restatement
    (nnls_batch.hpp:70-132)."""
    torch, _abi, ctx = env
    n = 70000 if lg == 1 else (36000 if lg == 2 else 20000)       # > 1024 waves x 64 / lg slots: refills happen
    rs = np.random.default_rng(100 * lg + k)
    Fm = rs.uniform(size=(4 * k + 5, k))
    G = (Fm.T @ Fm).astype(np.float32)
    G[np.diag_indices(k)] += np.float32(1e-15)
    B = (rs.standard_normal((n, k)) * 3 + 1).astype(np.float32)
    X0 = rs.uniform(size=(n, k)).astype(np.float32)
    dG, dB = _dev(torch, G), _dev(torch, B)
    ctx.set_option(_abi.OPT_CD_LMF_LANE_GROUPS, lg)
    ctx.set_option(_abi.OPT_CD_LMF_WAVES_PER_SIMD, 1)
    try:
        for kw in (dict(warm=1), dict(warm=0), dict(zero_init=1), dict(warm=1, l1_pre=0.3, ub_post=0.4)):
            outs = []
            for var in (_abi.CD_GROUP, _abi.CD_LMF):
                dX = _dev(torch, X0.copy())
                ctx.solve_cd(_abi.F32, dG, dB, dX, k, n, maxit=6, tol=0.0, variant=var, **kw)
                outs.append(dX.cpu().numpy())
            assert np.array_equal(outs[0], outs[1]) and not np.signbit(outs[1]).any(), kw
        # early exit: per-column sweep counts and iterates vs the oracle on a sample of columns; the work counters
        ctx.stats(reset=True)
        dX = _dev(torch, X0.copy())
        d_sw = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        ctx.solve_cd(_abi.F32, dG, dB, dX, k, n, warm=1, maxit=60, tol=1e-4, variant=_abi.CD_LMF, sweeps_out=d_sw)
        X, sw = dX.cpu().numpy(), d_sw.cpu().numpy()
        st = ctx.stats(reset=True)
        assert sw.min() >= 1 and sw.max() <= 60
        assert st["cd_columns"] == n and st["cd_column_sweeps"] == int(sw.sum())
        assert st["cd_slot_sweeps"] >= st["cd_column_sweeps"] + n       # + one correction sweep per column
        nbad = 0
        for j in rs.choice(n, 48, replace=False):
            b = B[j] - G @ X0[j]
            xr, _, it = O.cd_col(G, b.astype(np.float32), X0[j].copy(), maxit=60, tol=1e-4)
            assert np.abs(X[j] - xr).max() <= 3e-4 * max(1.0, np.abs(xr).max()), j
            nbad += int(abs(int(sw[j]) - int(it)) > 1)       # v_rcp_f32 in the tolerance term: the exit sweep may differ by one
        assert nbad <= 2
    finally:
        ctx.set_option(_abi.OPT_CD_LMF_LANE_GROUPS, 0)
        ctx.set_option(_abi.OPT_CD_LMF_WAVES_PER_SIMD, 0)



@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("norm_type", [0, 1, 2])
@pytest.mark.parametrize("k,c,ranked", [(64, 100003, True), (64, 20000, True), (10, 1183, False), (7, 5, False), (100, 17000, True),
                                        (128, 40000, True), (30, 16384, True)])
def test_scale_order_equals_separate_ops(env, dtype, norm_type, k, c, ranked):
    """rcppml_hip_scale_order (the iteration's tail in three launches: row sums beside the work-order histogram, their final sum,
    scaling beside the scatter) returns the row sums, d, X and work order of rcppml_hip_row_norms + rcppml_hip_apply_scaling +
    rcppml_hip_order_columns BIT FOR BIT -- same bodies, grids and summation orders.  Reference for the scaling:
    nmf/variant_helpers.hpp:286-305."""
    torch, _abi, ctx = env
    rs = np.random.default_rng(k * 7 + c)
    X = (rs.uniform(size=(c, k)) * (rs.uniform(size=(c, k)) < 0.6)).astype(dtype)
    X[:, 0] = 0                                                    # dead factor -> d = 1e-15
    sw = rs.integers(0, 110, size=c).astype(np.int32)
    dt, tt = _dt(_abi, dtype), _tt(torch, dtype)
    d_sw = _dev(torch, sw)
    # separate ops
    X1 = _dev(torch, X)
    s1 = torch.empty(k, dtype=tt, device="cuda"); d1 = torch.empty(k, dtype=tt, device="cuda")
    o1 = torch.full((c,), -1, dtype=torch.int32, device="cuda")
    ctx.row_norms(dt, X1, k, c, norm_type, s1)
    ctx.apply_scaling(dt, X1, k, c, norm_type, s1, d1)
    if ranked:
        ctx.order_columns(d_sw, c, o1)
    for rep in range(2):
        X2 = _dev(torch, X)
        s2 = torch.full((k,), -7.0, dtype=tt, device="cuda"); d2 = torch.full((k,), -7.0, dtype=tt, device="cuda")
        o2 = torch.full((c,), -1, dtype=torch.int32, device="cuda")
        ctx.scale_order(dt, X2, k, c, norm_type, s2, d2, d_sw if ranked else None, o2 if ranked else None)
        ctx.sync()
        assert torch.equal(s1, s2), (rep, "row sums")
        assert torch.equal(d1, d2), (rep, "d")
        assert torch.equal(X1, X2), (rep, "X")
        assert torch.equal(o1, o2), (rep, "order")
    Xr, dr = O.extract_scaling(X, norm_type)
    assert rel_err(d2.cpu().numpy(), dr) < TOL[dtype]
    assert rel_err(X2.cpu().numpy(), Xr) < TOL[dtype] * 2


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("k,m", [(64, 20000), (10, 183), (32, 3867), (100, 5000), (128, 30000), (5, 3)])
def test_gram_loss_mse_equals_separate_ops(env, dtype, k, m):
    """rcppml_hip_gram_loss_mse (Gram partials; ONE launch in which the cross-term partials run beside the Gram's final sum; the
    loss's final sum) = rcppml_hip_gram + rcppml_hip_loss_mse bit for bit: G_wt and out[0..2].  Reference: nmf/fit_cpu.hpp:1729-1753."""
    torch, _abi, ctx = env
    rs = np.random.default_rng(k + m)
    dt, tt = _dt(_abi, dtype), _tt(torch, dtype)
    W = _dev(torch, (rs.uniform(size=(m, k)) * (rs.uniform(size=(m, k)) < 0.7)).astype(dtype))
    Bw = _dev(torch, rs.normal(size=(m, k)).astype(dtype))
    d = _dev(torch, rs.uniform(0.5, 2.0, size=k).astype(dtype))
    Gs = _dev(torch, rs.uniform(size=(k, k)).astype(dtype))
    tr = torch.tensor([123.456], dtype=torch.float64, device="cuda")
    G1 = torch.empty((k, k), dtype=tt, device="cuda"); out1 = torch.zeros(4, dtype=torch.float64, device="cuda")
    ctx.gram(dt, W, k, m, 1e-15, 0.0, G1)
    ctx.loss_mse(dt, tr, d, W, Bw, k, m, G1, Gs, out1)
    for rep in range(2):
        G2 = torch.full((k, k), -3.0, dtype=tt, device="cuda"); out2 = torch.full((4,), -3.0, dtype=torch.float64, device="cuda")
        ctx.gram_loss_mse(dt, W, k, m, 1e-15, tr, d, Bw, Gs, G2, out2)
        ctx.sync()
        assert torch.equal(G1, G2), rep
        assert torch.equal(out1[:3], out2[:3]), (rep, out1.tolist(), out2.tolist())
    # oracle values of the three terms
    Wd = W.cpu().numpy().astype(np.float64); dd = d.cpu().numpy().astype(np.float64)
    cross = float(np.sum(Wd * dd[None, :] * Bw.cpu().numpy().astype(np.float64)))
    recon = float(np.sum(np.outer(dd, dd) * (Wd.T @ Wd + 1e-15 * np.eye(k)) * Gs.cpu().numpy().astype(np.float64).T))
    got = out2.cpu().numpy()
    assert abs(got[1] - cross) <= 1e-5 * max(1.0, abs(cross)) and abs(got[2] - recon) <= (1e-4 if dtype == np.float32 else 1e-9) * max(1.0, abs(recon))


def test_fused_tail_under_graph_replay(env):
    """Both fused-tail ops inside one captured hipGraph, replayed: every replay reproduces the eager results bit for bit (the
    plugin's steady-state iteration and bench.py's timed region run them this way)."""
    torch, _abi, _ = env
    k, c = 64, 50000
    rs = np.random.default_rng(5)
    X0 = _dev(torch, rs.uniform(size=(c, k)).astype(np.float32))
    sw = _dev(torch, rs.integers(0, 100, size=c).astype(np.int32))
    Bw = _dev(torch, rs.normal(size=(c, k)).astype(np.float32))
    Gs = _dev(torch, rs.uniform(size=(k, k)).astype(np.float32))
    tr = torch.tensor([9.0], dtype=torch.float64, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        ctx = _abi.Context(torch.cuda.current_device())
        X = X0.clone()
        s = torch.empty(k, device="cuda"); d = torch.empty(k, device="cuda"); o = torch.empty(c, dtype=torch.int32, device="cuda")
        G = torch.empty((k, k), device="cuda"); out = torch.zeros(4, dtype=torch.float64, device="cuda")

        def body():
            X.copy_(X0)
            ctx.scale_order(_abi.F32, X, k, c, 0, s, d, sw, o)
            ctx.gram_loss_mse(_abi.F32, X, k, c, 1e-15, tr, d, Bw, Gs, G, out)
            X.copy_(X0)
            ctx.tail_scale_gram(_abi.F32, X, k, c, 0, s, d, sw, o, 1e-15, 0.0, G)        # (the k = 64 form: scaling inside the Gram kernel)
            X.copy_(X0)
            ctx.tail_scale_gram_loss(_abi.F32, X, k, c, 0, s, d, sw, o, 1e-15, tr, Bw, Gs, G, out)
        body(); body()
        torch.cuda.synchronize()
        want = [t.clone() for t in (X, s, d, o, G, out[:3])]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            body()
        for rep in range(4):
            for t in (s, d, G, out):
                t.fill_(-1)
            g.replay()
            torch.cuda.synchronize()
            for a, b in zip(want, (X, s, d, o, G, out[:3])):
                assert torch.equal(a, b), rep


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("norm_type", [0, 1, 2])
@pytest.mark.parametrize("k,c,ranked", [(64, 100003, True), (64, 20000, True), (64, 777, False), (64, 1, False), (32, 20000, True), (10, 1183, False),
                                        (60, 20000, True), (49, 777, False)])
def test_tail_ops_equal_separate_ops(env, dtype, norm_type, k, c, ranked):
    """rcppml_hip_tail_scale_gram / rcppml_hip_tail_scale_gram_loss (a half-update's whole tail in one call; fp32 with k = 64 scales
    the factor INSIDE the Gram's partial-tile kernel -- the lane that loads an element divides it, stores it back and feeds the scaled
    value to the matrix cores) against the separate ops row_norms + apply_scaling + order_columns + gram (+ loss_mse): row sums, d, the
    scaled factor, the work order, the Gram and the three loss terms BIT FOR BIT, every shape (the others take the unfused calls).
    Reference: nmf/variant_helpers.hpp:286-305, primitives/cpu/gram.hpp:37-67, nmf/fit_cpu.hpp:1729-1753."""
    torch, _abi, ctx = env
    rs = np.random.default_rng(k * 11 + c)
    X = (rs.uniform(size=(c, k)) * (rs.uniform(size=(c, k)) < 0.6)).astype(dtype)
    X[:, 0] = 0
    sw = _dev(torch, rs.integers(0, 110, size=c).astype(np.int32))
    dt, tt = _dt(_abi, dtype), _tt(torch, dtype)
    Bw = _dev(torch, rs.normal(size=(c, k)).astype(dtype))
    Gs = _dev(torch, rs.uniform(size=(k, k)).astype(dtype))
    tr = torch.tensor([77.0], dtype=torch.float64, device="cuda")
    X1 = _dev(torch, X)
    s1 = torch.empty(k, dtype=tt, device="cuda"); d1 = torch.empty(k, dtype=tt, device="cuda")
    o1 = torch.full((c,), -1, dtype=torch.int32, device="cuda")
    G1 = torch.empty((k, k), dtype=tt, device="cuda"); Gl1 = torch.empty((k, k), dtype=tt, device="cuda")
    out1 = torch.zeros(4, dtype=torch.float64, device="cuda")
    ctx.row_norms(dt, X1, k, c, norm_type, s1)
    ctx.apply_scaling(dt, X1, k, c, norm_type, s1, d1)
    if ranked:
        ctx.order_columns(sw, c, o1)
    ctx.gram(dt, X1, k, c, 1e-15, 0.25, G1)
    ctx.gram(dt, X1, k, c, 1e-15, 0.0, Gl1)
    ctx.loss_mse(dt, tr, d1, X1, Bw, k, c, Gl1, Gs, out1)
    for which in ("gram", "loss"):
        X2 = _dev(torch, X)
        s2 = torch.full((k,), -7.0, dtype=tt, device="cuda"); d2 = torch.full((k,), -7.0, dtype=tt, device="cuda")
        o2 = torch.full((c,), -1, dtype=torch.int32, device="cuda")
        G2 = torch.full((k, k), -3.0, dtype=tt, device="cuda"); out2 = torch.full((4,), -3.0, dtype=torch.float64, device="cuda")
        if which == "gram":
            ctx.tail_scale_gram(dt, X2, k, c, norm_type, s2, d2, sw if ranked else None, o2 if ranked else None, 1e-15, 0.25, G2)
        else:
            ctx.tail_scale_gram_loss(dt, X2, k, c, norm_type, s2, d2, sw if ranked else None, o2 if ranked else None, 1e-15, tr, Bw, Gs, G2, out2)
        ctx.sync()
        assert torch.equal(s1, s2) and torch.equal(d1, d2), which
        assert torch.equal(X1, X2), which
        assert torch.equal(o1, o2), which
        assert torch.equal(G1 if which == "gram" else Gl1, G2), which
        if which == "loss":
            assert torch.equal(out1[:3], out2[:3]), (out1.tolist(), out2.tolist())
