"""Randomised parity sweep (GPU vs oracle, through the device-level C ABI) over awkward ranks and shapes: ranks that are
not multiples of the vector width (scalar-load fallbacks), ranks straddling the MFMA tile sizes (16/32/48/64/96/128),
single columns, empty columns, columns longer than one staged chunk.  Seeds are fixed: failures reproduce."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import random_csc, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    from rcppml_amd import _abi
    return torch, _abi, _abi.Context(0)


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


RANKS = [1, 2, 5, 7, 12, 15, 16, 17, 24, 31, 32, 33, 40, 47, 48, 49, 63, 64, 65, 80, 96, 97, 127, 128, 129, 160, 200, 256]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_fuzz_rhs_gram_scaling(env, dtype):
    torch, _abi, ctx = env
    dt = _abi.F32 if dtype == np.float32 else _abi.F64
    tt = torch.float32 if dtype == np.float32 else torch.float64
    tol = 5e-5 if dtype == np.float32 else 1e-11
    rs = np.random.default_rng(2024)
    for k in RANKS:
        rows, cols = int(rs.integers(1, 400)), int(rs.integers(1, 300))
        dens = float(rs.choice([0.01, 0.1, 0.6]))
        A = random_csc(rows, cols, dens, seed=k + 17)
        F = rs.standard_normal((rows, k)).astype(dtype)
        B_ref = O.rhs(A, F, dtype)
        dB = torch.full((cols, k), -3.0, dtype=tt, device="cuda")
        ctx.rhs(dt, _dev(torch, A.p), _dev(torch, A.i), _dev(torch, A.values(dtype)), cols, _dev(torch, F), k, dB)
        assert rel_err(dB.cpu().numpy(), B_ref) < tol, ("rhs", k, rows, cols, dens)
        G_ref = O.gram(F)
        dG = torch.empty((k, k), dtype=tt, device="cuda")
        ctx.gram(dt, _dev(torch, F), k, rows, 1e-15, 0.0, dG)
        G = dG.cpu().numpy()
        assert np.array_equal(G, G.T) and rel_err(G, G_ref) < tol * 4, ("gram", k, rows)
        for norm_type in (0, 1):
            sums = torch.empty((k,), dtype=tt, device="cuda")
            ctx.row_norms(dt, _dev(torch, np.abs(F)), k, rows, norm_type, sums)
            ref = np.abs(F).sum(axis=0) if norm_type == 0 else (F.astype(np.float64) ** 2).sum(axis=0)
            assert rel_err(sums.cpu().numpy().astype(np.float64), ref.astype(np.float64)) < tol * 10, ("norms", k, rows, norm_type)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_fuzz_cd_auto(env, dtype):
    """The AUTO solve (MFMA kernels for k <= 128, the general-rank wave-per-column kernel up to 256) on odd ranks and
    column counts, cold and warm, with and without the early exit."""
    torch, _abi, ctx = env
    dt = _abi.F32 if dtype == np.float32 else _abi.F64
    tol = 3e-4 if dtype == np.float32 else 1e-9
    rs = np.random.default_rng(7)
    for k in RANKS:
        n = int(rs.choice([1, 15, 16, 17, 31, 33, 64, 65, 130]))
        Fm = rs.uniform(size=(4 * k + 5, k))
        G = (Fm.T @ Fm).astype(dtype)
        G[np.diag_indices(k)] += dtype(1e-15)
        B = (rs.standard_normal((n, k)) * 3 + 1).astype(dtype)
        X0 = rs.uniform(size=(n, k)).astype(dtype)
        for warm, maxit, cdtol in ((False, 100, 1e-8), (True, 9, 0.0)):
            ref = O.nnls_batch(G, B, X=X0 if warm else None, maxit=maxit, tol=cdtol, warm=warm)
            dX = _dev(torch, X0.copy())
            ctx.solve_cd(dt, _dev(torch, G), _dev(torch, B), dX, k, n, warm=1 if warm else 0, zero_init=0 if warm else 1,
                         maxit=maxit, tol=cdtol)
            X = dX.cpu().numpy()
            assert X.min() >= 0
            # fp32 at k > 64: 4k+5 uniform samples give a Gram with condition ~1e4, a few sweeps amplify rounding
            tk = tol * ((16 if k > 128 else 4) if (dtype == np.float32 and k > 64) else 1)
            assert np.abs(X - ref).max() / max(np.abs(ref).max(), 1e-30) < tk, ("cd", k, n, warm)


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-7), (np.float32, 3e-2)])
def test_fuzz_irls_nb(env, dtype, tol):
    """NB-IRLS half-update on ranks around the MFMA kernel's domain (fp32, k<=32, k%4==0) and outside it, with columns
    shorter and longer than one 32-nonzero chunk, empty columns included."""
    torch, _abi, ctx = env
    dt = _abi.F32 if dtype == np.float32 else _abi.F64
    tt = torch.float32 if dtype == np.float32 else torch.float64
    rs = np.random.default_rng(99)
    for k in (3, 4, 8, 12, 20, 28, 32, 33, 40, 64):
        rows, cols = int(rs.integers(40, 200)), int(rs.integers(5, 90))
        dens = float(rs.choice([0.02, 0.3, 0.9]))
        A = random_csc(rows, cols, dens, seed=k)
        A.x[:] = np.ceil(np.abs(A.x) * 6.0)            # counts >= 1
        F = rs.uniform(0.05, 1.0, size=(rows, k)).astype(dtype)
        F /= F.sum(axis=0, keepdims=True)
        F *= 30.0
        G = O.gram(F)
        theta = rs.uniform(2.0, 20.0, size=rows).astype(dtype)
        ref = O.irls_nb(A, F, G, k, L1=0.0, L2=1e-3, theta_row=theta, dtype=dtype)
        dX = torch.full((cols, k), 3.0, dtype=tt, device="cuda")
        ctx.solve_irls_nb(dt, _dev(torch, A.p), _dev(torch, A.i), _dev(torch, A.values(dtype)), cols, _dev(torch, F),
                          _dev(torch, G), dX, k, l1=0.0, l2=1e-3, theta_row=_dev(torch, theta), theta_col=None)
        X = dX.cpu().numpy()
        assert np.all(np.isfinite(X)) and X.min() >= 0
        assert np.abs(X - ref).max() / max(np.abs(ref).max(), 1e-30) < tol, ("irls", k, rows, cols, dens)


def test_fuzz_plugin_fits_fp64():
    """73-pointer entry (fp64) vs the oracle's nmf_fit on random small problems with random options: odd ranks,
    L1/L2 on either side, upper bounds, L2 normalisation, both solvers, early stopping.  Five iterations: with k above
    the data's rank, bounds and Cholesky+clip the ALS map is not contractive and rounding differences grow ~30x per
    iteration (1e-14 after one iteration, 1e-9 after five, 1e-3 after fifteen -- tools/probe/dbg_fuzz.py)."""
    from rcppml_amd import _abi
    from tests.util import lowrank_csc
    rs = np.random.default_rng(31337)
    for trial in range(10):
        k = int(rs.choice([2, 3, 7, 9, 16, 17, 33]))
        m, n = int(rs.integers(k + 5, 120)), int(rs.integers(k + 5, 150))
        A = lowrank_csc(m, n, max(2, k // 2), float(rs.choice([0.15, 0.5])), seed=trial)
        W0, H0 = O.init_factors(int(rs.integers(1, 1000)), k, m, n, np.float64)
        solver = int(rs.integers(0, 2))
        L1 = (float(rs.choice([0.0, 0.01])), float(rs.choice([0.0, 0.05])))
        L2 = (float(rs.choice([0.0, 0.1])), float(rs.choice([0.0, 0.01])))
        ub = (0.0, float(rs.choice([0.0, 0.0, 0.2])))
        norm_type = int(rs.choice([0, 0, 1]))
        tol = float(rs.choice([0.0, 1e-4]))
        ref = O.nmf_fit(A, W0, H0, np.float64, max_iter=5, tol=tol, solver_mode=solver, L1=L1, L2=L2, ub=ub, norm_type=norm_type)
        W, H = W0.copy(), H0.copy()
        res = _abi.nmf_unified(A.p, A.i, A.x, m, n, k, W, H, entry="double", max_iter=5, tol=tol, solver_mode=solver,
                               L1_W=L1[0], L1_H=L1[1], L2_W=L2[0], L2_H=L2[1], ub_W=ub[0], ub_H=ub[1], norm_type=norm_type)
        cfg = (trial, k, m, n, solver, L1, L2, ub, norm_type, tol)
        assert res["status"] == 0, cfg
        assert res["iter"] == ref.iter, cfg
        assert abs(res["loss"] - ref.loss) <= 1e-6 * abs(ref.loss) + 1e-12, cfg
        assert np.abs(res["d"] - ref.d).max() <= 1e-6 * np.abs(ref.d).max(), cfg
        assert np.abs(W - ref.W_T).max() < 1e-6 and np.abs(H - ref.H).max() < 1e-6, cfg
