"""CPU tests of the oracle (oracle/): pinned against every known answer the reference's own tests hold
for this path and the properties its R test-suite asserts, so the GPU
parity tests compare against something that is itself checked.  No GPU needed."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import load_fixture, lowrank_csc, random_csc


# ---------------------------------------------------------------- known answers held by the reference
def test_splitmix64_published_vectors():
    # SplitMix64 reference implementation (Vigna) test vector for seed 1234567
    assert O.splitmix_stream(1234567, 5) == [6457827717110365317, 3203168211198807973, 9817491932198370423,
                                             4593380528125082431, 16408922859458223821]


def test_splitmix64_zero_seed_remapped():
    # reference tests/cpp/test_rng.cpp:36-39
    assert O.splitmix_state(0) == 12345
    assert O.splitmix_stream(0, 3) == O.splitmix_stream(12345, 3)


def test_rng_deterministic_and_uniform_range():
    # reference tests/cpp/test_rng.cpp:9-34,75-85
    assert O.splitmix_stream(42, 100) == O.splitmix_stream(42, 100)
    assert sum(a == b for a, b in zip(O.splitmix_stream(42, 100), O.splitmix_stream(99, 100))) < 5
    M = O.fill_uniform(42, 5, 10)
    assert M.min() >= 0.0 and M.max() < 1.0 and M.max() - M.min() > 0.1


def test_nnls_known_solution_2x2():
    # reference tests/cpp/test_nnls.cpp:65-84
    G = np.array([[2.0, 1.0], [1.0, 2.0]]) + 1e-10 * np.eye(2)
    X = O.nnls_batch(G, np.array([[3.0, 3.0]]), maxit=100, tol=1e-8)
    assert np.allclose(X, [[1.0, 1.0]], atol=1e-4)


def test_nnls_identity_gram():
    # reference tests/cpp/test_nnls.cpp:11-29
    rng = np.random.default_rng(0)
    G = np.eye(3) + 1e-10 * np.eye(3)
    B = np.abs(rng.standard_normal((5, 3)))
    assert np.allclose(O.nnls_batch(G, B), B, atol=1e-4)
    Bn = rng.standard_normal((5, 3))
    Bn[0, 0] = -1.0
    assert np.allclose(O.nnls_batch(G, Bn, nonneg=False)[0, 0], -1.0, atol=1e-4)   # :48-63 unconstrained


def test_nnls_nonnegativity_l1_warmstart():
    # reference tests/cpp/test_nnls.cpp:31-46,86-109,111-140
    rng = np.random.default_rng(1)
    H = np.abs(rng.standard_normal((50, 4)))
    G = O.gram(H)
    B = rng.standard_normal((10, 4))
    X = O.nnls_batch(G, B)
    assert X.min() >= 0.0
    Bp = np.abs(B)
    assert np.abs(O.nnls_batch(G, Bp)).sum() >= np.abs(O.nnls_batch(G, Bp, L1=1.0)).sum()
    Xc = O.nnls_batch(G, Bp)
    Xw = O.nnls_batch(G, Bp, X=Xc, warm=True)
    assert np.allclose(Xc, Xw, atol=1e-4)


def test_gram_matches_product():
    # reference tests/cpp/test_gram.cpp:11-80
    rng = np.random.default_rng(2)
    H = rng.standard_normal((40, 6))            # (cols, k)
    G = O.gram(H)
    assert np.allclose(G, H.T @ H + 1e-15 * np.eye(6), atol=1e-8)
    assert np.array_equal(G, G.T)
    assert np.all(np.diag(G) > 0)
    assert np.allclose(O.gram(2 * H), 4 * G, atol=1e-6)
    assert np.allclose(O.gram(np.eye(3)), np.eye(3), atol=1e-6)


def test_reconstruct_known_value():
    # reference tests/cpp/test_nmf.cpp:14-27: W = ones(10,3), d = (2,3,1), H = ones(3,20) -> W diag(d) H = 6
    W_T = np.ones((10, 3))
    H = np.ones((20, 3))
    d = np.array([2.0, 3.0, 1.0])
    A6 = O.Csc.from_dense(np.full((10, 20), 6.0))
    assert O.evaluate_mse(W_T, d, H, A6) < 1e-20
    A7 = O.Csc.from_dense(np.full((10, 20), 7.0))
    assert abs(O.evaluate_mse(W_T, d, H, A7) - 1.0) < 1e-12


# ---------------------------------------------------------------- algorithmic properties
def _kkt_violation(G, b, x):
    g = G @ x - b
    return max(np.max(np.abs(g[x > 0]), initial=0.0), np.max(-g[x == 0], initial=0.0))


def test_cd_satisfies_kkt():
    rng = np.random.default_rng(3)
    F = rng.uniform(size=(60, 12))
    G = O.gram(F)
    B = rng.standard_normal((30, 12)) * 2 + 1
    X = O.nnls_batch(G, B, maxit=2000, tol=1e-14)
    for j in range(30):
        assert _kkt_violation(G, B[j], X[j]) < 1e-6


def test_llt_restatement():
    rng = np.random.default_rng(4)
    F = rng.uniform(size=(50, 9))
    G = O.gram(F) + 0.1 * np.eye(9)
    L = O.llt(G)
    assert np.allclose(L @ L.T, G, atol=1e-12)
    assert np.allclose(L, np.linalg.cholesky(G), atol=1e-12)
    B = rng.standard_normal((7, 9))
    X = O.chol_clip_batch(G, B, nonneg=False)
    assert np.allclose(X, np.linalg.solve(G, B.T).T, atol=1e-10)
    assert O.chol_clip_batch(G, B).min() >= 0


def test_fused_equals_unfused_and_iter0_quirk():
    """fused_rhs_nnls_sparse == rhs -> nnls_batch(warm); at iter 0 it starts from X WITHOUT residual correction (F7)."""
    A = random_csc(80, 50, 0.15, seed=5)
    rng = np.random.default_rng(5)
    F = rng.uniform(size=(80, 6))
    X0 = rng.uniform(size=(50, 6))
    G = O.gram(F)
    B = O.rhs(A, F)
    warm = O.fused_cd(A, F, G, X0, maxit=50, tol=1e-8, L1=0.05, warm=True)
    assert np.allclose(warm, O.nnls_batch(G, B - 0.05, X=X0, maxit=50, tol=1e-8, warm=True), atol=1e-12)
    quirk = O.fused_cd(A, F, G, X0, maxit=3, tol=0.0, warm=False)
    ref = np.stack([O.cd_col(G, B[j], X0[j], maxit=3)[0] for j in range(50)])
    assert np.allclose(quirk, ref, atol=1e-13)
    cold = O.nnls_batch(G, B, maxit=3, tol=0.0)
    assert not np.allclose(quirk, cold)


def test_transpose_and_cross_term():
    A = lowrank_csc(40, 30, 3, 0.3, seed=6)
    At = A.transpose()
    assert np.array_equal(At.toarray(), A.toarray().T)
    rng = np.random.default_rng(6)
    W_T, H, d = rng.uniform(size=(40, 5)), rng.uniform(size=(30, 5)), rng.uniform(1, 2, size=5)
    cross = O.loss_cross(At, W_T, H, d)
    dense = np.sum(A.toarray() * ((W_T * d) @ H.T))
    assert abs(cross - dense) / abs(dense) < 1e-12
    assert abs(O.trace_AtA(A) - np.sum(A.x ** 2)) < 1e-9


# ---------------------------------------------------------------- fit-level properties (reference R testthat)
@pytest.mark.parametrize("solver", [0, 1])
def test_fit_loss_is_gram_trick_of_true_loss(solver):
    A = lowrank_csc(60, 90, 4, 0.25, seed=7)
    W0, H0 = O.init_factors(1, 5, A.rows, A.cols)
    r = O.nmf_fit(A, W0, H0, max_iter=15, tol=0.0, solver_mode=solver)
    dense = A.toarray()
    true = np.sum((dense - (r.W_T * r.d) @ r.H.T) ** 2)
    assert abs(r.loss - true) / true < 1e-9
    assert abs(O.evaluate_mse(r.W_T, r.d, r.H, A) - true / dense.size) / (true / dense.size) < 1e-9
    assert r.W_T.min() >= 0 and r.H.min() >= 0 and np.all(r.d > 0)           # test_rcpp_bridge_roundtrip.R:26-46
    assert np.all(np.diff(r.d) <= 0)                                         # sorted, test_nmf.cpp:45-67
    assert np.allclose(r.W_T.sum(axis=0), 1.0) and np.allclose(r.H.sum(axis=0), 1.0)


@pytest.mark.parametrize("kw,slack", [(dict(), 1e-5), (dict(L1=(0.01, 0.01)), 1e-3), (dict(L1=(0.01, 0.01), L2=(0.01, 0.01)), 1e-3)])
def test_loss_monotonicity(kw, slack):
    # reference tests/testthat/test_loss_monotonicity.R:6-87
    A = lowrank_csc(100, 120, 5, 0.2, seed=8)
    W0, H0 = O.init_factors(3, 6, A.rows, A.cols)
    h = O.nmf_fit(A, W0, H0, max_iter=25, tol=0.0, **kw).loss_history
    assert np.all(np.diff(h) <= slack * h[:-1])


def test_l1_increases_sparsity_and_rank_lowers_loss():
    # reference tests/testthat/test_nmf.R:40-51, test_convergence.R:158-173
    A = lowrank_csc(80, 100, 6, 0.3, seed=9)
    W0, H0 = O.init_factors(4, 6, A.rows, A.cols)
    r0 = O.nmf_fit(A, W0, H0, max_iter=20, tol=0.0)
    r1 = O.nmf_fit(A, W0, H0, max_iter=20, tol=0.0, L1=(0.5, 0.5))
    assert (r1.H == 0).mean() >= (r0.H == 0).mean()
    losses = []
    for k in (2, 4, 8):
        Wk, Hk = O.init_factors(4, k, A.rows, A.cols)
        losses.append(O.nmf_fit(A, Wk, Hk, max_iter=30, tol=0.0).loss)
    assert losses[0] >= losses[1] >= losses[2]


def test_same_seed_bitwise_and_threads_agree():
    # reference tests/testthat/test_nmf.R:58-71 (threads=1 bitwise), test_thread_edge_cases.R:31-42 (d to 1e-4)
    A = load_fixture("hawaiibirds")
    W0, H0 = O.init_factors(42, 10, A.rows, A.cols, np.float32)
    a = O.nmf_fit(A, W0, H0, np.float32, max_iter=8, tol=0.0, threads=1)
    b = O.nmf_fit(A, W0, H0, np.float32, max_iter=8, tol=0.0, threads=1)
    assert np.array_equal(a.W_T, b.W_T) and np.array_equal(a.H, b.H) and a.loss == b.loss
    c = O.nmf_fit(A, W0, H0, np.float32, max_iter=8, tol=0.0, threads=4)
    assert np.allclose(a.d, c.d, rtol=1e-4)


def test_convergence_patience_and_upper_bound():
    A = lowrank_csc(60, 70, 3, 0.3, seed=10)
    W0, H0 = O.init_factors(5, 3, A.rows, A.cols)
    r = O.nmf_fit(A, W0, H0, max_iter=200, tol=1e-3, patience=5)
    assert r.converged and 6 <= r.iter < 200 and r.tol < 1e-3
    h = r.loss_history
    rel = np.abs(np.diff(h)) / (np.abs(h[:-1]) + 1e-15)
    assert np.all(rel[-5:] < 1e-3)                                           # `patience` consecutive hits
    rb = O.nmf_fit(A, W0, H0, max_iter=5, tol=0.0, ub=(0.02, 0.03), norm_type=2, sort_model=False)
    assert rb.W_T.max() <= 0.02 + 1e-12 and rb.H.max() <= 0.03 + 1e-12      # test_upper_bound.R:9-71


@pytest.mark.parametrize("case", ["k1", "single_col", "single_row", "zero_rows", "tiny", "near_zero", "identical_cols"])
def test_degenerate_inputs_finite(case):
    # reference tests/testthat/test_degenerate_inputs.R:5-126: must not crash, factors finite
    rng = np.random.default_rng(11)
    k = 2
    if case == "k1":
        D, k = rng.uniform(size=(20, 15)), 1
    elif case == "single_col":
        D = rng.uniform(size=(20, 1))
    elif case == "single_row":
        D = rng.uniform(size=(1, 20))
    elif case == "zero_rows":
        D = rng.uniform(size=(20, 15)); D[3] = 0; D[:, 4] = 0
    elif case == "tiny":
        D = np.array([[1.0, 2.0], [3.0, 4.0]])
    elif case == "near_zero":
        D = np.full((10, 10), 1e-15)
    else:
        col = rng.uniform(size=(20, 1)); D = np.repeat(col, 10, axis=1)
    A = O.Csc.from_dense(D)
    W0, H0 = O.init_factors(7, k, A.rows, A.cols)
    for solver in (0, 1):
        r = O.nmf_fit(A, W0, H0, max_iter=10, tol=0.0, solver_mode=solver)
        assert np.all(np.isfinite(r.W_T)) and np.all(np.isfinite(r.H)) and np.all(np.isfinite(r.d))
    if case == "identical_cols":
        r = O.nmf_fit(A, W0[:, :1].copy(), H0[:, :1].copy(), max_iter=30, tol=0.0)
        assert np.sqrt(r.loss / np.sum(D ** 2)) < 0.01                        # rank-1 relative error < 1 %


def test_masked_fit_properties():
    """Explicit mask (reference nmf/masked_nnls.hpp): masked entries do not influence the fit."""
    A = lowrank_csc(50, 60, 3, 0.4, seed=12)
    M = random_csc(50, 60, 0.1, seed=13)
    W0, H0 = O.init_factors(2, 4, A.rows, A.cols)
    r = O.nmf_fit(A, W0, H0, max_iter=10, tol=0.0, mask=M)
    # perturb A at masked positions: result must be identical
    D = A.toarray(); Md = M.toarray() != 0
    D2 = D.copy(); D2[Md & (D != 0)] *= 3.0
    r2 = O.nmf_fit(O.Csc.from_dense(D2), W0, H0, max_iter=10, tol=0.0, mask=M)
    assert np.allclose(r.W_T, r2.W_T, atol=1e-12) and np.allclose(r.H, r2.H, atol=1e-12)
    pred = (r.W_T * r.d) @ r.H.T
    true = np.sum(((D - pred) ** 2)[(D != 0) & ~Md])
    assert abs(r.loss - true) / true < 1e-10


def test_c_nnls_and_predict_semantics():
    """L1 inside CD for nnls()/predict() (F9); warm start keeps the fixed point; recovery > 0.9 (test_nnls.R:2-33)."""
    rng = np.random.default_rng(14)
    w = rng.uniform(size=(100, 5))
    h_true = rng.gamma(1.0, 1.0, size=(40, 5))
    D = w @ h_true.T
    A = O.Csc.from_dense(D)
    h = O.c_nnls(w, A)
    assert np.corrcoef(h.ravel(), h_true.ravel())[0, 1] > 0.9
    h_l1 = O.c_nnls(w, A, L1=0.5)
    assert h_l1.sum() < h.sum()
    hw = O.c_nnls(w, A, h0=h)
    assert np.allclose(hw, h, atol=1e-6)


def test_proj_adv_gram_matches_numpy_eigh():
    """PROJ_ADV (nmf/variant_helpers.hpp:112-146): G - |lambda| (tr G / tr TG) TG with eigenvalues below 1e-8 raised to 1e-8.
    The reference uses Eigen's SelfAdjointEigenSolver (third-party, absent); the clipped matrix is a function of G alone, so
    the oracle's Jacobi iteration is pinned against LAPACK (numpy.linalg.eigh): 1e-11, clipped and unclipped cases."""
    rng = np.random.default_rng(0)
    for k in (3, 12, 40):
        F = rng.standard_normal((3 * k + 5, k))
        G = F.T @ F
        T = rng.standard_normal((2 * k, k))
        TG = T.T @ T / (2 * k)
        for lam in (1e-3, 0.3, 2.5):
            got = O.proj_adv(G, TG, lam)
            G2 = G - lam * (np.trace(G) / np.trace(TG)) * TG
            w, V = np.linalg.eigh(G2)
            ref = G2 if w.min() >= 1e-8 else (V * np.maximum(w, 1e-8)) @ V.T
            assert np.abs(got - ref).max() <= 1e-11 * np.abs(ref).max()
            assert np.allclose(got, got.T, atol=1e-12 * np.abs(got).max())


def test_corrected_reciprocal_quotient_equals_division():
    """The product's fp64 coordinate sweeps keep 1 / G_cc per column and form the step's quotient as q0 = b ginv,
    q = q0 + (b - q0 g) ginv (cd_quotient in rcppml_amd/csrc/kernels.hip.h): with a correctly rounded reciprocal and an exact
    residual (one fma) that IS the reference's `b[i] / g_diag` (nnls_batch.hpp:100).  2e8 random operand pairs, host fma (IEEE,
    as the device's): not one differs from the division operator."""
    import ctypes as C
    L = O.lib()
    L.oracle_corrected_quotient_mismatches.restype = C.c_longlong
    b, g = C.c_double(0), C.c_double(0)
    bad = 0
    for seed in (1, 2, 3, 4):
        bad += L.oracle_corrected_quotient_mismatches(C.c_uint64(seed), C.c_longlong(50_000_000), C.byref(b), C.byref(g))
    assert bad == 0, (bad, b.value, g.value)
