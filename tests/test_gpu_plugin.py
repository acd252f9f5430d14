"""GPU parity tests, plugin level: the 73-pointer entry points of RcppML_gpu.so (exactly the call the
reference bridge makes, inst/include/FactorNet/gpu/bridge_nmf.hpp:310-342) against the CPU oracle's
restatement of nmf_fit<CPU> on identical inputs (same CSC, same W_init/H_init, explicit solver).

Tolerances: fp64 entry vs fp64 oracle -- loss relative <= 1e-6 (north star) and in practice ~1e-10;
factors max-abs <= 1e-6 after L1 normalisation.  fp32 entry vs fp32 oracle -- both sides carry fp32
rounding in different orders, loss relative <= 2e-4, factors <= 2e-3.
"""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import load_fixture, lowrank_csc, random_csc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def abi():
    from rcppml_amd import _abi
    return _abi


def _run_gpu(abi, A, W0, H0, entry, **kw):
    W = W0.astype(np.float64).copy()
    H = H0.astype(np.float64).copy()
    res = abi.nmf_unified(A.p, A.i, A.x, A.rows, A.cols, W.shape[1], W, H, entry=entry, **kw)
    assert res["status"] == 0, res.get("error")
    res["W_T"], res["H"] = W, H
    return res


def _compare(res, ref, tol_loss, tol_fac):
    assert res["iter"] == ref.iter
    assert abs(res["loss"] - ref.loss) / abs(ref.loss) < tol_loss
    assert np.abs(res["d"] - ref.d).max() / np.abs(ref.d).max() < tol_fac
    assert np.abs(res["W_T"] - ref.W_T).max() < tol_fac
    assert np.abs(res["H"] - ref.H).max() < tol_fac


def test_detect(abi):
    devs = abi.detect()
    assert len(devs) >= 1 and devs[0][0] > 1000.0 and 0 < devs[0][1] <= devs[0][0]


@pytest.mark.parametrize("solver", [1, 0])
def test_hawaiibirds_fp64(abi, solver):
    """BASELINE config C1: hawaiibirds k=10 (auto solver on CPU = Cholesky+clip; also CD)."""
    A = load_fixture("hawaiibirds")
    W0, H0 = O.init_factors(42, 10, A.rows, A.cols, np.float64)
    ref = O.nmf_fit(A, W0, H0, np.float64, max_iter=30, tol=1e-4, solver_mode=solver)
    res = _run_gpu(abi, A, W0, H0, "double", max_iter=30, tol=1e-4, solver_mode=solver)
    _compare(res, ref, 1e-6, 1e-6)
    assert res["converged"] == ref.converged


def test_hawaiibirds_fp32(abi):
    A = load_fixture("hawaiibirds")
    W0, H0 = O.init_factors(42, 10, A.rows, A.cols, np.float32)
    ref = O.nmf_fit(A, W0, H0, np.float32, max_iter=20, tol=0.0, solver_mode=1)
    res = _run_gpu(abi, A, W0, H0, "float", max_iter=20, tol=0.0, solver_mode=1)
    _compare(res, ref, 2e-4, 2e-3)


def test_movielens_l1(abi):
    """BASELINE config C3: movielens k=32, L1=c(0,0.1) -> L1_W=0, L1_H=0.1, CD (mask='zeros' is a fit-time no-op)."""
    A = load_fixture("movielens")
    k = 32
    W0, H0 = O.init_factors(7, k, A.rows, A.cols, np.float64)
    ref = O.nmf_fit(A, W0, H0, np.float64, max_iter=15, tol=1e-4, L1=(0.0, 0.1), solver_mode=0)
    res = _run_gpu(abi, A, W0, H0, "double", max_iter=15, tol=1e-4, L1_H=0.1, L1_W=0.0, solver_mode=0)
    _compare(res, ref, 1e-6, 1e-6)
    assert (res["H"] == 0).mean() > 0.05     # L1 produces exact zeros


@pytest.mark.parametrize("k", [5, 16, 64])
def test_synthetic_cd_history(abi, k):
    A = lowrank_csc(400, 700, 8, 0.06, seed=k)
    W0, H0 = O.init_factors(123, k, A.rows, A.cols, np.float64)
    ref = O.nmf_fit(A, W0, H0, np.float64, max_iter=12, tol=0.0, L2=(0.01, 0.02), solver_mode=0)
    res = _run_gpu(abi, A, W0, H0, "ex", max_iter=12, tol=0.0, L2_W=0.01, L2_H=0.02, solver_mode=0, precision=1,
                   want_history=True)
    _compare(res, ref, 1e-6, 1e-6)
    h = res["loss_history"]
    assert np.abs(h - ref.loss_history).max() / ref.loss_history.max() < 1e-8
    assert np.all(np.diff(h) <= 1e-4 * h[:-1])          # loss non-increasing (reference test_loss_monotonicity.R)
    assert res["W_T"].min() >= 0 and res["H"].min() >= 0 and np.all(res["d"] > 0)
    assert np.allclose(res["H"].sum(axis=0), 1.0, atol=1e-9)    # rows of H sum to 1 under L1 norm
    assert np.all(np.diff(res["d"]) <= 0)                       # sorted by descending d


@pytest.mark.parametrize("k,precision,tol", [(128, 1, 1e-6), (100, 1, 1e-6), (128, 0, 2e-3), (100, 0, 2e-3)])
def test_fit_at_c4_ranks(abi, k, precision, tol):
    """Whole-fit parity at BASELINE configs[3]'s rank (k = 128) and at a rank that is not a multiple of the MFMA tile
    (k = 100), CD, vs the oracle's fp64 nmf_fit: loss, W, H, d, iteration count.  The fp32 mode is held against the fp64
    oracle as well: at k = 128 the fp32 ORACLE itself lands 0.35 away from its fp64 self after 8 iterations (two nearly
    equal d swap places in the final sort), while the GPU's fp32 result stays within 2e-5 of the fp64 one
    (tools/probe/c4rank_probe.py)."""
    A = lowrank_csc(500, 900, 12, 0.08, seed=k)
    W0, H0 = O.init_factors(77, k, A.rows, A.cols, np.float64)
    ref = O.nmf_fit(A, W0, H0, np.float64, max_iter=8, tol=0.0, solver_mode=0)
    res = _run_gpu(abi, A, W0, H0, "ex", max_iter=8, tol=0.0, solver_mode=0, precision=precision)
    _compare(res, ref, tol if precision == 1 else 2e-4, tol)


@pytest.mark.parametrize("k", [129, 160, 256])
def test_fit_above_rank_128(abi, k):
    """Ranks the MFMA tiles do not cover (128 < k <= 256) are not handed back: general-rank CD kernel (one wavefront per
    column, Gram from L2), generic rhs / Gram / scaling kernels.  fp64 entry vs the oracle's fp64 fit; fp32 entry vs the
    fp64 oracle as in test_fit_at_c4_ranks.  k = 257 is rejected loudly."""
    A = lowrank_csc(420, 640, 10, 0.1, seed=k)
    W0, H0 = O.init_factors(5, k, A.rows, A.cols, np.float64)
    ref = O.nmf_fit(A, W0, H0, np.float64, max_iter=5, tol=0.0, solver_mode=0, L1=(0.0, 0.01))
    res = _run_gpu(abi, A, W0, H0, "ex", max_iter=5, tol=0.0, solver_mode=0, L1_H=0.01, precision=1)
    _compare(res, ref, 1e-6, 1e-6)
    res32 = _run_gpu(abi, A, W0, H0, "ex", max_iter=5, tol=0.0, solver_mode=0, L1_H=0.01, precision=0)
    _compare(res32, ref, 5e-4, 5e-3)
    if k == 256:
        W1, H1 = O.init_factors(5, 257, A.rows, A.cols, np.float64)
        r = abi.nmf_unified(A.p, A.i, A.x, A.rows, A.cols, 257, W1, H1, entry="double", max_iter=2)
        assert r["status"] == -1 and "256" in r["error"]


@pytest.mark.parametrize("solver", [0, 1])
@pytest.mark.parametrize("lam_w,lam_h", [(0.5, 0.0), (0.0, 0.8), (0.3, 0.4), (-0.4, 0.0), (0.0, -0.6), (-0.3, 0.5)])
def test_target_regularisation(abi, solver, lam_w, lam_h):
    """Target regularisation (nmf/variant_helpers.hpp:107-146) through the build-defined
    rcppml_gpu_nmf_target entry: enrichment (lambda > 0: ridge toward the target) and PROJ_ADV (lambda < 0: trace-scaled
    target Gram removed from G, eigenvalues clipped at 1e-8) on either or both factors, standard (unfused) path, fp64,
    against the oracle's restatement: loss <= 1e-6, factors <= 1e-6."""
    if solver == 1:
        # Cholesky on a Gram whose small eigenvalues were clipped to 1e-8 amplifies rounding by 1e8 (the reference's own
        # result is unstable there): PROJ_ADV is exercised with the Cholesky solver at a strength that keeps G well conditioned
        lam_w, lam_h = (lam_w * 0.1 if lam_w < 0 else lam_w), (lam_h * 0.1 if lam_h < 0 else lam_h)
    A = lowrank_csc(140, 190, 5, 0.15, seed=21)
    k = 7
    rng = np.random.default_rng(4)
    W0, H0 = O.init_factors(13, k, A.rows, A.cols, np.float64)
    TW = rng.uniform(0, 2.0 / A.rows, size=(A.rows, k))
    TH = rng.uniform(0, 2.0 / A.cols, size=(A.cols, k))
    tw = (TW, lam_w) if lam_w != 0 else None
    th = (TH, lam_h) if lam_h != 0 else None
    ref = O.nmf_fit(A, W0, H0, np.float64, max_iter=8, tol=0.0, solver_mode=solver, L1=(0.001, 0.002), L2=(0.01, 0.0),
                    target_W=tw, target_H=th)
    res = _run_gpu(abi, A, W0, H0, "ex", max_iter=8, tol=0.0, solver_mode=solver, precision=1, L1_W=0.001, L1_H=0.002,
                   L2_W=0.01, L2_H=0.0, target_W=tw, target_H=th)
    _compare(res, ref, 1e-6, 1e-6)
    # the target changes the fit (the test would pass trivially if it were dropped)
    plain = O.nmf_fit(A, W0, H0, np.float64, max_iter=8, tol=0.0, solver_mode=solver, L1=(0.001, 0.002), L2=(0.01, 0.0), unfused=True)
    assert np.abs(plain.W_T - ref.W_T).max() > 1e-5 or np.abs(plain.H - ref.H).max() > 1e-5


def test_target_regularisation_fp32_and_refusals(abi):
    A = lowrank_csc(140, 190, 5, 0.15, seed=22)
    k = 6
    rng = np.random.default_rng(5)
    W0, H0 = O.init_factors(3, k, A.rows, A.cols, np.float64)
    TH = rng.uniform(0, 2.0 / A.cols, size=(A.cols, k))
    ref = O.nmf_fit(A, W0, H0, np.float64, max_iter=6, tol=0.0, target_H=(TH, 0.7))
    res = _run_gpu(abi, A, W0, H0, "ex", max_iter=6, tol=0.0, precision=0, target_H=(TH, 0.7))
    _compare(res, ref, 2e-4, 2e-3)
    # not combinable with IRLS losses / masks: rejected (status -1), never silently dropped
    W, H = W0.copy(), H0.copy()
    bad = abi.nmf_unified(A.p, A.i, A.x, A.rows, A.cols, k, W, H, entry="ex", max_iter=2, loss_type=5, target_H=(TH, 0.7))
    assert bad["status"] != 0 and "target" in bad["error"]


@pytest.mark.parametrize("solver", [0, 1])
def test_upper_bound_with_explicit_mask(abi, solver):
    """upper_bound on the explicit-mask path: the reference clips after every half-update branch (fit_cpu.hpp:636-637,
    884-885); an earlier build ignored ub_W there and refused ub_H."""
    A = lowrank_csc(120, 160, 4, 0.15, seed=31)
    M = random_csc(120, 160, 0.05, seed=32)
    k = 5
    W0, H0 = O.init_factors(9, k, A.rows, A.cols, np.float64)
    ref = O.nmf_fit(A, W0, H0, np.float64, max_iter=7, tol=0.0, mask=M, solver_mode=solver, ub=(0.02, 0.015))
    res = _run_gpu(abi, A, W0, H0, "ex", max_iter=7, tol=0.0, mask=(M.p, M.i), solver_mode=solver, precision=1, ub_W=0.02, ub_H=0.015)
    _compare(res, ref, 1e-6, 1e-6)
    raw = O.nmf_fit(A, W0, H0, np.float64, max_iter=7, tol=0.0, mask=M, solver_mode=solver)
    assert np.abs(raw.W_T - ref.W_T).max() > 1e-4          # the bounds bite


def test_upper_bound_and_norms(abi):
    A = lowrank_csc(200, 300, 4, 0.1, seed=3)
    k = 6
    W0, H0 = O.init_factors(5, k, A.rows, A.cols, np.float64)
    for norm_type in (1, 2):
        ref = O.nmf_fit(A, W0, H0, np.float64, max_iter=8, tol=0.0, norm_type=norm_type, ub=(0.5, 0.4))
        res = _run_gpu(abi, A, W0, H0, "double", max_iter=8, tol=0.0, norm_type=norm_type, ub_W=0.5, ub_H=0.4)
        _compare(res, ref, 1e-6, 1e-6)


@pytest.mark.parametrize("k", [8, 40, 96])
def test_explicit_mask(abi, k):
    """Explicit-mask path (reference nmf/masked_nnls.hpp) through the build-defined rcppml_gpu_nmf_ex entry
    (k = 8: 32-wide kernel instantiation, k = 40: 64-wide, k = 96: two features per lane, kernels_wide.hip.h)."""
    A = lowrank_csc(120, 160, 4, 0.15, seed=11)
    M = random_csc(120, 160, 0.05, seed=12)
    W0, H0 = O.init_factors(9, k, A.rows, A.cols, np.float64)
    for solver in (0, 1):
        ref = O.nmf_fit(A, W0, H0, np.float64, max_iter=10, tol=0.0, mask=M, solver_mode=solver, L1=(0.01, 0.02), L2=(0.03, 0.0))
        res = _run_gpu(abi, A, W0, H0, "ex", max_iter=10, tol=0.0, mask=(M.p, M.i), solver_mode=solver, precision=1,
                       L1_W=0.01, L1_H=0.02, L2_W=0.03, L2_H=0.0)
        _compare(res, ref, 1e-6, 1e-6)


def test_unsupported_features_are_rejected(abi):
    """The plugin must set out_status=-1 (-> caller's CPU fallback) instead of silently dropping a feature."""
    A = lowrank_csc(50, 60, 3, 0.2, seed=1)
    W0, H0 = O.init_factors(1, 4, A.rows, A.cols, np.float64)
    for kw in (dict(L21_H=0.1, loss_type=5), dict(ortho_W=-0.1), dict(projective=1, loss_type=5), dict(symmetric=1), dict(symmetric=1, projective=1), dict(loss_type=3),
               dict(loss_type=1), dict(loss_type=4, gp_dispersion_mode=3), dict(loss_type=6, gp_dispersion_mode=3), dict(loss_type=5, solver_mode=1),
               dict(graph_W_nnz=5, loss_type=5), dict(guide_H_count=1), dict(solver_mode=2)):
        W, H = W0.copy(), H0.copy()
        r = abi.nmf_unified(A.p, A.i, A.x, A.rows, A.cols, 4, W, H, entry="double", max_iter=2, **kw)
        assert r["status"] == -1 and r["error"], kw


def test_nnls_predict_evaluate(abi):
    """fp64 nnls()/predict()/evaluate() entries vs src/RcppFunctions_utils.cpp restatement."""
    A = lowrank_csc(150, 90, 5, 0.2, seed=21)
    k = 7
    w_T = np.random.default_rng(0).uniform(size=(A.rows, k))
    h_ref = O.c_nnls(w_T, A, L1=0.01, L2=0.02)
    h = np.zeros((A.cols, k))
    abi.nnls_double(A.p, A.i, A.x, A.rows, A.cols, k, w_T, h, L1=0.01, L2=0.02)
    assert np.abs(h - h_ref).max() < 1e-9
    h_warm_ref = O.c_nnls(w_T, A, h0=h_ref * 0.9, cd_maxit=5)
    h2 = h_ref * 0.9
    abi.nnls_double(A.p, A.i, A.x, A.rows, A.cols, k, w_T, h2, cd_maxit=5, warm=1)
    assert np.abs(h2 - h_warm_ref).max() < 1e-9
    d = np.random.default_rng(1).uniform(0.5, 2, size=k)
    for mz in (False, True):
        ref = O.evaluate_mse(w_T, d, h_ref, A, mask_zeros=mz)
        got = abi.evaluate_mse_double(A.p, A.i, A.x, A.rows, A.cols, k, w_T, d, h_ref, mask_zeros=mz)
        assert abs(got - ref) / ref < 1e-10


def _csc_from_dense(D):
    from oracle.oracle import Csc
    D = np.asarray(D, np.float64)
    p, ii, xx = [0], [], []
    for j in range(D.shape[1]):
        nz = np.nonzero(D[:, j])[0]
        ii.extend(nz.tolist())
        xx.extend(D[nz, j].tolist())
        p.append(len(ii))
    return Csc(D.shape, np.array(p, np.int64), np.array(ii, np.int64), np.array(xx, np.float64))


@pytest.mark.parametrize("entry", ["double", "float"])
def test_degenerate_inputs(abi, entry):
    """reference tests/testthat/test_degenerate_inputs.R:5-126: k=1, single row / column, all-zero rows and columns,
    2x2, identical columns (rank-1 relative error < 1 %), near-zero matrix, > 99 % sparse; k > min(m, n) may be refused
    but must not crash.  Factors must be finite and non-negative; where the oracle runs, the plugin must agree."""
    rs = np.random.default_rng(5)
    dense = {
        "k1": (rs.uniform(size=(12, 9)), 1),
        "single_row": (rs.uniform(0.1, 1, size=(1, 15)), 1),
        "single_col": (rs.uniform(0.1, 1, size=(14, 1)), 1),
        "zero_rows_cols": (np.pad(rs.uniform(size=(6, 7)), ((2, 3), (1, 2))), 3),
        "two_by_two": (np.array([[1.0, 2.0], [3.0, 4.0]]), 2),
        "identical_cols": (np.outer(rs.uniform(0.5, 1, size=10), np.ones(8)), 2),
        "near_zero": (rs.uniform(size=(9, 11)) * 1e-15, 2),
        "very_sparse": ((rs.uniform(size=(60, 70)) > 0.995) * rs.uniform(0.5, 1, size=(60, 70)), 3),
        "all_zero": (np.zeros((5, 6)), 2),
    }
    dtype = np.float64 if entry == "double" else np.float32
    for name, (D, k) in dense.items():
        A = _csc_from_dense(D)
        m, n = D.shape
        W0, H0 = O.init_factors(7, k, m, n, np.float64)
        W, H = W0.copy(), H0.copy()
        res = abi.nmf_unified(A.p, A.i, A.x, m, n, k, W, H, entry=entry, max_iter=8, tol=0.0, solver_mode=0)
        assert res["status"] == 0, (name, res.get("error"))
        assert np.all(np.isfinite(W)) and np.all(np.isfinite(H)) and np.all(np.isfinite(res["d"])), name
        assert W.min() >= 0 and H.min() >= 0, name
        if name == "near_zero":
            # entries ~1e-15 sit at the 1e-15 floors of the algorithm (Gram eps, d = sum + 1e-15, CD tolerance): the
            # iteration is rounding noise in ANY implementation (the fp32 oracle itself jumps between 2e-29 and 4e-17
            # from one iteration to the next, tools/probe/dbg_nearzero.py); the reference only asks for finite factors
            continue
        ref = O.nmf_fit(A, W0, H0, dtype, max_iter=8, tol=0.0, solver_mode=0)
        tol = 1e-6 if entry == "double" else 5e-3
        # exact fits leave the Gram-trick loss at the cancellation floor of tr(A'A): compare on that scale too
        floor = (1e-12 if entry == "double" else 1e-5) * float(np.sum(A.x ** 2))
        # (absolute floor: with an all-zero matrix the fp32 kernels, which multiply by 1/G_ii where the reference divides, may leave
        #  factors of ~1e-9 instead of exact zeros -- a loss of 1e-18 where the oracle's is 0)
        assert abs(res["loss"] - ref.loss) <= tol * abs(ref.loss) + floor + (1e-18 if entry == "double" else 1e-15), (name, res["loss"], ref.loss)
        if name == "identical_cols":
            R = (W * res["d"][None, :]) @ H.T
            assert np.linalg.norm(R - D) / np.linalg.norm(D) < 1e-2
    # k > min(m, n): refused or solved, never a crash
    D = rs.uniform(size=(3, 4))
    A = _csc_from_dense(D)
    W0, H0 = O.init_factors(7, 5, 3, 4, np.float64)
    W, H = W0.copy(), H0.copy()
    res = abi.nmf_unified(A.p, A.i, A.x, 3, 4, 5, W, H, entry=entry, max_iter=3, tol=0.0, solver_mode=0)
    assert res["status"] in (0, -1)
    if res["status"] == 0:
        assert np.all(np.isfinite(W)) and np.all(np.isfinite(H))


def test_reference_properties_on_gpu(abi):
    """Properties the reference's own test-suite asserts, on the GPU path:
    same inputs -> bitwise identical factors (test_nmf.R:58-71; here: run-to-run determinism of the kernels, including
    the sweep-sorted work order whose scatter uses atomics), loss non-increasing within +1e-5 relative without
    regularisation (test_loss_monotonicity.R:6-25), higher rank -> lower-or-equal loss (test_convergence.R:158-173),
    non-negativity (test_nmf.R:13-18)."""
    A = load_fixture("movielens")
    losses = {}
    for k in (4, 12):
        W0, H0 = O.init_factors(3, k, A.rows, A.cols, np.float64)
        runs = []
        for rep in range(2):
            W, H = W0.copy(), H0.copy()
            res = abi.nmf_unified(A.p, A.i, A.x, A.rows, A.cols, k, W, H, entry="ex", max_iter=25, tol=0.0, solver_mode=0,
                                  precision=0, want_history=True)
            assert res["status"] == 0
            runs.append((W, H, res))
        assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])
        assert np.array_equal(runs[0][2]["d"], runs[1][2]["d"]) and runs[0][2]["loss"] == runs[1][2]["loss"]
        hist = np.asarray(runs[0][2]["loss_history"][:25])
        assert np.all(hist[2:] <= hist[1:-1] * (1 + 1e-5)), hist      # iteration 1 is the F7 quirk's starting point
        assert runs[0][0].min() >= 0 and runs[0][1].min() >= 0
        losses[k] = runs[0][2]["loss"]
    assert losses[12] <= losses[4] * (1 + 1e-6)


def test_run_to_run_determinism_with_work_order(abi):
    """Large enough (> 32768 columns) for the sweep-sorted column order to be active: two fits, identical bits."""
    A = lowrank_csc(300, 40000, 6, 0.02, seed=21)
    k = 16
    W0, H0 = O.init_factors(5, k, A.rows, A.cols, np.float64)
    outs = []
    for rep in range(2):
        W, H = W0.copy(), H0.copy()
        res = abi.nmf_unified(A.p, A.i, A.x, A.rows, A.cols, k, W, H, entry="float", max_iter=6, tol=0.0, solver_mode=0)
        assert res["status"] == 0
        outs.append((W, H, res["d"].copy(), res["loss"]))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][2], outs[1][2]) and outs[0][3] == outs[1][3]


@pytest.mark.parametrize("entry,tol_loss,tol_fac", [("double", 1e-6, 1e-6), ("float", 5e-4, 5e-3)])
def test_l21_and_angular_features(abi, entry, tol_loss, tol_fac):
    """k x k feature layer: L21 (features/L21.hpp) on the Gram before the solve and the post-hoc
    angular decorrelation (features/angular.hpp) after it, on either side, vs the oracle fit."""
    A = lowrank_csc(90, 140, 5, 0.25, seed=31)
    k = 7
    dtype = np.float64 if entry == "double" else np.float32
    W0, H0 = O.init_factors(13, k, A.rows, A.cols, np.float64)
    for L21, ang in (((0.05, 0.0), (0.0, 0.0)), ((0.0, 0.2), (0.0, 0.0)), ((0.0, 0.0), (0.03, 0.0)), ((0.0, 0.0), (0.0, 0.05)),
                     ((0.02, 0.04), (0.01, 0.02))):
        ref = O.nmf_fit(A, W0, H0, dtype, max_iter=6, tol=0.0, solver_mode=0, L21=L21, angular=ang)
        res = _run_gpu(abi, A, W0, H0, entry, max_iter=6, tol=0.0, solver_mode=0, L21_W=L21[0], L21_H=L21[1],
                       ortho_W=ang[0], ortho_H=ang[1])
        _compare(res, ref, tol_loss, tol_fac)
    # features change the fit (guards against silently ignored arguments)
    base = O.nmf_fit(A, W0, H0, dtype, max_iter=6, tol=0.0, solver_mode=0)
    pen = O.nmf_fit(A, W0, H0, dtype, max_iter=6, tol=0.0, solver_mode=0, L21=(0.05, 0.2), angular=(0.03, 0.05))
    assert abs(pen.loss - base.loss) > 1e-3 * abs(base.loss)


@pytest.mark.parametrize("k", [65, 100, 128])
def test_features_at_ranks_above_64(abi, k):
    """L21, angular and graph-Laplacian features for 64 < k <= 128 (two features per lane in the angular kernel, 128-wide
    tiles in the cross-Gram of the graph term), fp64 entry vs the oracle fit; k = 129 with these features is rejected."""
    A = lowrank_csc(300, 420, 9, 0.12, seed=k)
    W0, H0 = O.init_factors(21, k, A.rows, A.cols, np.float64)
    LW, LH = _ring_laplacian(A.rows, 1), _ring_laplacian(A.cols, 2, hops=3)
    ref = O.nmf_fit(A, W0, H0, np.float64, max_iter=4, tol=0.0, solver_mode=0, L21=(0.02, 0.04), angular=(0.01, 0.02),
                    graph_W=(LW, 0.03), graph_H=(LH, 0.02))
    res = _run_gpu(abi, A, W0, H0, "double", max_iter=4, tol=0.0, solver_mode=0, L21_W=0.02, L21_H=0.04, ortho_W=0.01, ortho_H=0.02,
                   graph_W=(LW.p, LW.i, LW.x, 0.03), graph_H=(LH.p, LH.i, LH.x, 0.02))
    _compare(res, ref, 1e-6, 1e-6)
    if k == 128:
        W1, H1 = O.init_factors(21, 129, A.rows, A.cols, np.float64)
        for kw in (dict(ortho_H=0.02), dict(graph_W=(LW.p, LW.i, LW.x, 0.03))):
            r = abi.nmf_unified(A.p, A.i, A.x, A.rows, A.cols, 129, W1.copy(), H1.copy(), entry="double", max_iter=2, **kw)
            assert r["status"] == -1 and "128" in r["error"], kw


@pytest.mark.parametrize("entry,tol_loss,tol_fac", [("double", 1e-6, 1e-6), ("float", 5e-4, 5e-3)])
def test_symmetric_nmf(abi, entry, tol_loss, tol_fac):
    """symmetric = TRUE (A ~ W diag(d) W^T, fit_cpu.hpp:659-704): H is never solved, W is solved against its own Gram and
    W_T A with nnls_batch semantics (zero start, then residual-corrected warm start), H = W_T; the loss pairs the new W_T
    with the Gram / RHS of the old one.  Both solvers and the k x k features, vs the oracle."""
    import scipy.sparse as sp
    rng = np.random.default_rng(5)
    n, k = 120, 6
    F = rng.uniform(0, 1, (n, 4)) * (rng.random((n, 4)) < 0.4)
    S = sp.csc_matrix((F @ F.T) * (rng.random((n, n)) < 0.4))
    S = sp.csc_matrix((S + S.T) / 2)
    S.sort_indices()
    A = O.Csc(S.shape, S.indptr, S.indices, S.data)
    dtype = np.float64 if entry == "double" else np.float32
    W0, H0 = O.init_factors(7, k, n, n, np.float64)
    for solver, kw_o, kw_g in ((0, {}, {}), (1, {}, {}), (0, dict(L1=(0.01, 0.0), L2=(0.02, 0.0), L21=(0.03, 0.0)), dict(L1_W=0.01, L2_W=0.02, L21_W=0.03)),
                               (0, dict(ub=(0.05, 0.0), angular=(0.02, 0.0)), dict(ub_W=0.05, ortho_W=0.02))):
        ref = O.nmf_fit(A, W0, H0, dtype, max_iter=8, tol=0.0, solver_mode=solver, symmetric=True, **kw_o)
        res = _run_gpu(abi, A, W0, H0, entry, max_iter=8, tol=0.0, solver_mode=solver, symmetric=1, **kw_g)
        _compare(res, ref, tol_loss, tol_fac)
        assert np.array_equal(res["W_T"], res["H"])
    std = O.nmf_fit(A, W0, H0, dtype, max_iter=8, tol=0.0, solver_mode=0)
    sym = O.nmf_fit(A, W0, H0, dtype, max_iter=8, tol=0.0, solver_mode=0, symmetric=True)
    assert abs(sym.loss - std.loss) > 1e-6 * abs(std.loss) and sym.loss_history[-1] < sym.loss_history[0]
    # a non-square matrix is handed back
    B = lowrank_csc(50, 60, 3, 0.2, seed=1)
    W1, H1 = O.init_factors(1, 4, B.rows, B.cols, np.float64)
    r = abi.nmf_unified(B.p, B.i, B.x, B.rows, B.cols, 4, W1, H1, entry="double", max_iter=2, symmetric=1)
    assert r["status"] == -1 and "square" in r["error"]


def _ring_laplacian(dim, seed, hops=2):
    """Symmetric graph Laplacian L = D - A of a weighted ring with `hops` neighbours on each side (CSC)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    rows, cols, vals = [], [], []
    for h in range(1, hops + 1):
        w = rng.uniform(0.2, 1.0, dim)
        i = np.arange(dim); j = (i + h) % dim
        rows += [i, j]; cols += [j, i]; vals += [w, w]
    Adj = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(dim, dim))
    Lap = sp.csc_matrix(sp.diags(np.asarray(Adj.sum(axis=0)).ravel()) - Adj)
    Lap.sort_indices()
    return O.Csc((dim, dim), Lap.indptr.astype(np.int32), Lap.indices.astype(np.int32), Lap.data.astype(np.float64))


@pytest.mark.parametrize("entry,tol_loss,tol_fac", [("double", 1e-6, 1e-6), ("float", 5e-4, 5e-3)])
def test_graph_regularisation(abi, entry, tol_loss, tol_fac):
    """Graph-Laplacian smoothness (features/graph_reg.hpp:52-66, fit_cpu.hpp:508-509 / :740-741): G += lambda (F L) F^T on
    either side, through the 73-pointer entry's graph_W_* / graph_H_* slots, vs the oracle fit."""
    A = lowrank_csc(90, 140, 5, 0.25, seed=51)
    k = 7
    dtype = np.float64 if entry == "double" else np.float32
    W0, H0 = O.init_factors(17, k, A.rows, A.cols, np.float64)
    LW, LH = _ring_laplacian(A.rows, 1), _ring_laplacian(A.cols, 2, hops=3)
    for lw, lh in ((0.05, 0.0), (0.0, 0.08), (0.03, 0.02)):
        for solver in (0, 1):
            ref = O.nmf_fit(A, W0, H0, dtype, max_iter=6, tol=0.0, solver_mode=solver, L2=(0.01, 0.0),
                            graph_W=(LW, lw) if lw else None, graph_H=(LH, lh) if lh else None)
            res = _run_gpu(abi, A, W0, H0, entry, max_iter=6, tol=0.0, solver_mode=solver, L2_W=0.01,
                           graph_W=(LW.p, LW.i, LW.x, lw) if lw else None, graph_H=(LH.p, LH.i, LH.x, lh) if lh else None)
            _compare(res, ref, tol_loss, tol_fac)
    base = O.nmf_fit(A, W0, H0, dtype, max_iter=6, tol=0.0, solver_mode=0)
    pen = O.nmf_fit(A, W0, H0, dtype, max_iter=6, tol=0.0, solver_mode=0, graph_W=(LW, 0.05), graph_H=(LH, 0.08))
    assert abs(pen.loss - base.loss) > 1e-3 * abs(base.loss)
    # dimension mismatch is rejected, not read out of bounds
    W, H = W0.copy(), H0.copy()
    r = abi.nmf_unified(A.p, A.i, A.x, A.rows, A.cols, k, W, H, entry="double", max_iter=2, graph_W=(LH.p, LH.i, LH.x, 0.1))
    assert r["status"] == -1 and "graph" in r["error"]


@pytest.mark.parametrize("entry,tol_loss,tol_fac", [("double", 1e-6, 1e-6), ("float", 5e-4, 5e-3)])
def test_projective_nmf(abi, entry, tol_loss, tol_fac):
    """projective = TRUE: H = (diag(d) W_T) A instead of an NNLS half-update (variant_helpers.hpp:308-325), W standard."""
    A = lowrank_csc(90, 140, 5, 0.25, seed=41)
    k = 6
    dtype = np.float64 if entry == "double" else np.float32
    W0, H0 = O.init_factors(13, k, A.rows, A.cols, np.float64)
    for solver in (0, 1):
        ref = O.nmf_fit(A, W0, H0, dtype, max_iter=8, tol=0.0, solver_mode=solver, L1=(0.01, 0.0), projective=True)
        res = _run_gpu(abi, A, W0, H0, entry, max_iter=8, tol=0.0, solver_mode=solver, L1_W=0.01, projective=1)
        _compare(res, ref, tol_loss, tol_fac)
    base = O.nmf_fit(A, W0, H0, dtype, max_iter=8, tol=0.0, solver_mode=0)
    proj = O.nmf_fit(A, W0, H0, dtype, max_iter=8, tol=0.0, solver_mode=0, projective=True)
    assert abs(proj.loss - base.loss) > 1e-6 * abs(base.loss)


def test_zerocopy_entry_matches_host_entry(abi):
    """rcppml_gpu_nmf_zerocopy_double (reference src/gpu_bridge_nmf.cu:879-967): the CSC is handed over as device
    pointers encoded in doubles; the fit must be bit-identical to the 73-pointer fp64 entry on the same inputs."""
    import torch
    A = load_fixture("hawaiibirds")
    k = 8
    W0, H0 = O.init_factors(21, k, A.rows, A.cols, np.float64)
    W1, H1 = W0.copy(), H0.copy()
    r1 = abi.nmf_unified(A.p, A.i, A.x, A.rows, A.cols, k, W1, H1, entry="double", max_iter=12, tol=0.0, solver_mode=0, L1_H=0.01)
    dp = torch.from_numpy(A.p.astype(np.int32)).cuda()
    di = torch.from_numpy(A.i.astype(np.int32)).cuda()
    dx = torch.from_numpy(A.x.astype(np.float64)).cuda()
    W2, H2 = W0.copy(), H0.copy()
    r2 = abi.nmf_zerocopy(dp, di, dx, A.rows, A.cols, A.nnz, k, W2, H2, max_iter=12, tol=0.0, L1_H=0.01)
    assert r1["status"] == 0 and r2["status"] == 0, (r1.get("error"), r2.get("error"))
    assert r2["iter"] == r1["iter"] and r2["loss"] == r1["loss"]
    assert np.array_equal(W1, W2) and np.array_equal(H1, H2) and np.array_equal(r1["d"], r2["d"])
    # the caller's device arrays are untouched
    assert np.array_equal(dx.cpu().numpy(), A.x) and np.array_equal(di.cpu().numpy(), A.i.astype(np.int32))
    bad = abi.nmf_zerocopy(dp, di, dx, A.rows, A.cols, A.nnz, k, W2, H2, max_iter=2, loss_type=2)      # Huber: not a loss of this build
    assert bad["status"] == -1


def test_profile_entry_phases_and_stopping_rule(abi):
    """rcppml_gpu_nmf_profile_double (reference src/gpu_bridge_utils.cu:48-57, the aux entry SURVEY.md section 5 lists): eleven
    HIP-event phase times per iteration, and the reference's stopping rule on the loss.  The protocol (:96-99 SplitMix64(seed)
    factors, :131-142 untimed warm-up iteration, fixed cd_maxit sweeps from max(B, 0) / the previous factor, L1 scaling, rel < tol
    from the second timed iteration) is restated here with the oracle's column CD, and the number of iterations the entry runs
    under a tolerance chosen in the widest gap of the oracle's loss sequence has to match."""
    A = lowrank_csc(70, 55, 4, 0.5, seed=9)
    m, n, k, cd_maxit, seed = A.rows, A.cols, 5, 7, 11
    D = np.zeros((m, n))
    for j in range(n):
        D[A.i[A.p[j]:A.p[j + 1]], j] = A.x[A.p[j]:A.p[j + 1]]
    W, H = O.init_factors(seed, k, m, n, np.float64)
    d = np.ones(k)

    def half(F, Bm, X, cold):
        G = F.T @ F + 1e-15 * np.eye(k)
        B = Bm @ F
        out = np.empty_like(X)
        for j in range(B.shape[0]):
            x0 = np.maximum(B[j], 0) if cold else X[j]
            out[j], _, _ = O.cd_col(G, B[j] - G @ x0, x0, maxit=cd_maxit, tol=0.0)
        s = np.abs(out).sum(axis=0)
        return out / (s + 1e-15), s + 1e-15

    H, d = half(W, D.T, H, True)
    W, d = half(H, D, W, True)
    rels, prev = [], None
    for it in range(12):
        H, d = half(W, D.T, H, it == 0)
        W, d = half(H, D, W, it == 0)
        loss = ((D - (W * d) @ H.T) ** 2).sum()
        if it > 0:
            rels.append(abs(prev - loss) / (abs(prev) + 1e-15))
        prev = loss
    # stop at the first rel < tol: pick tol in the widest relative gap between consecutive (decreasing) rels
    best = max(range(len(rels) - 1), key=lambda t: rels[t] / max(rels[t + 1], 1e-300) if rels[t + 1] < rels[t] and all(r > rels[t + 1] for r in rels[:t + 1]) else 0)
    assert rels[best] / rels[best + 1] > 1.3
    tol = float(np.sqrt(rels[best] * rels[best + 1]))
    res = abi.nmf_profile_double(A.p, A.i, A.x, m, n, k, max_iter=12, tol=tol, cd_maxit=cd_maxit, seed=seed)
    assert res["iters"] == best + 3                      # timed iteration 0 has no check; rels[t] belongs to timed iteration t + 1
    per, tot = res["per_iter_ms"], res["total_ms"]
    assert set(per) == set(abi.PROFILE_PHASES) and all(v > 0 for v in per.values())
    assert all(abs(tot[p] - per[p] * res["iters"]) <= 1e-9 * tot[p] for p in per)
    assert sum(v for p, v in per.items() if p != "total") <= per["total"] * 1.05
    full = abi.nmf_profile_double(A.p, A.i, A.x, m, n, k, max_iter=4, tol=0.0, cd_maxit=cd_maxit, seed=seed)
    assert full["iters"] == 4
    with pytest.raises(abi.BackendError):
        abi.nmf_profile_double(A.p, A.i, A.x, m, n, 300, max_iter=1)
