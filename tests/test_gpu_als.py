"""GPU parity tests of the host harness: the Python ALS loop over the device-level C-ABI (rcppml_amd/als.py,
the loop bench.py times), the R-surface mirror (rcppml_amd/nmf.py), and the BASELINE.json full-size workload
(configs[1]: 20000 x 100000, 1 %, k = 64) through size-independent properties plus exact per-column
spot checks against the oracle (columns of a half-update are independent given G and the fixed factor)."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import load_fixture, lowrank_csc

pytestmark = pytest.mark.gpu


def _oracle_csc(A):
    return O.Csc(A.shape, A.p, A.i, A.x)


@pytest.mark.parametrize("dtype,tol_loss,tol_fac", [("f64", 1e-9, 1e-8), ("f32", 2e-4, 5e-3)])
@pytest.mark.parametrize("solver", [0, 1])
def test_python_loop_matches_oracle(dtype, tol_loss, tol_fac, solver):
    from rcppml_amd import als, data
    A, _, _ = data.simulate_nmf_sparse(500, 900, 8, 0.05, seed=31)
    k = 16
    nd = np.float32 if dtype == "f32" else np.float64
    W0, H0 = data.init_factors(5, k, A.rows, A.cols, nd)
    cfg = als.AlsConfig(k=k, max_iter=10, tol=0.0, L1_H=1e-8, L2_W=1e-6, solver_mode=solver)
    st = als.ShardedALS(als.HipOps(0, dtype), als.Comm(None), A, A.transpose(), W0, H0, cfg)
    res = st.fit()
    ref = O.nmf_fit(_oracle_csc(A), W0, H0, nd, max_iter=10, tol=0.0, L1=(0.0, 1e-8), L2=(1e-6, 0.0), solver_mode=solver,
                    sort_model=False)
    hist = np.array(res["loss_history"])
    assert np.abs(hist - ref.loss_history).max() / ref.loss_history.max() < tol_loss
    W_T, d, H = st.factors()
    assert np.abs(W_T - ref.W_T).max() < tol_fac and np.abs(H - ref.H).max() < tol_fac
    assert np.abs(d - ref.d).max() / ref.d.max() < tol_fac


def test_nmf_surface_matches_reference_semantics():
    """nmf(seed=42): W_init = matrix(runif(m*k)) after set.seed (R/nmf_thin.R:790-797), H = SplitMix64(seed)
    (fit_cpu.hpp:200-207), auto solver, sorted factors; evaluate() mean vs misc$loss sum; predict()."""
    from rcppml_amd import data, nmf
    Ao = load_fixture("hawaiibirds")
    A = data.CSC(( Ao.rows, Ao.cols), Ao.p, Ao.i, Ao.x)
    k = 10
    m, n = A.shape
    model = nmf.nmf(A, k, seed=42, tol=1e-5, maxit=30, precision="fp64", solver="cholesky")
    W0 = data.r_runif(42, m * k).reshape(k, m).T.copy()
    H0 = data.splitmix64_uniform(42, 0, k * n, np.float64).reshape(n, k)
    ref = O.nmf_fit(Ao, W0, H0, np.float64, max_iter=30, tol=1e-5, solver_mode=1)
    assert model.misc["iter"] == ref.iter
    assert abs(model.misc["loss"] - ref.loss) / ref.loss < 1e-8
    assert np.abs(model.w - ref.W_T).max() < 1e-7 and np.abs(model.h.T - ref.H).max() < 1e-7
    assert np.all(np.diff(model.d) <= 0)
    mse = nmf.evaluate(model, A)
    assert abs(mse * m * n - model.misc["loss"]) / model.misc["loss"] < 1e-8          # mean vs sum
    assert abs(mse - O.evaluate_mse(ref.W_T, ref.d, ref.H, Ao)) / mse < 1e-7
    h = nmf.predict(model, A)
    h_ref = O.c_nnls(ref.W_T, Ao)
    assert np.abs(h.T - h_ref).max() / np.abs(h_ref).max() < 1e-6
    w2 = nmf.nnls(h=h, A=A)                                                             # solve for w given h
    assert w2.shape == (m, k) and w2.min() >= 0
    # auto solver with a GPU visible: CD for k <= 32 (R/nmf_thin.R:369)
    assert nmf.nmf(A, k, seed=1, maxit=2).misc["solver"] == "cd"
    assert nmf.gpu_available()


def test_nmf_surface_robust_and_graph():
    """nmf(robust = TRUE) -> Huber delta 1.345 (R/nmf_thin.R:343-352); nmf(graph_W, graph_H, graph_lambda = c(w, h))
    (R/nmf_thin.R:67-68, 500-506) -> the plugin's graph_* slots; both against the oracle fit from the same init."""
    import scipy.sparse as sp
    from rcppml_amd import data, nmf
    Ao = lowrank_csc(80, 110, 4, 0.3, seed=5)
    A = data.CSC((Ao.rows, Ao.cols), Ao.p, Ao.i, Ao.x)
    k, m, n = 5, Ao.rows, Ao.cols
    W0 = data.r_runif(7, m * k).reshape(k, m).T.copy()
    H0 = data.splitmix64_uniform(7, 0, k * n, np.float64).reshape(n, k)
    mod = nmf.nmf(A, k, seed=7, maxit=6, tol=0.0, precision="fp64", robust=True)
    ref = O.nmf_fit(Ao, W0, H0, np.float64, max_iter=6, tol=0.0, robust_delta=1.345, dispersion_mode=2)
    assert mod.misc["solver"] == "cd" and abs(mod.misc["loss"] - ref.loss) / abs(ref.loss) < 1e-6
    assert np.abs(mod.w - ref.W_T).max() < 1e-6

    def chain(dim):
        Adj = sp.diags([np.ones(dim - 1), np.ones(dim - 1)], [-1, 1], format="csc")
        return sp.csc_matrix(sp.diags(np.asarray(Adj.sum(axis=0)).ravel()) - Adj)
    LW, LH = chain(m), chain(n)
    mod = nmf.nmf(A, k, seed=7, maxit=6, tol=0.0, precision="fp64", solver="cd", graph_W=LW, graph_H=LH, graph_lambda=(0.05, 0.1))
    oc = lambda L: O.Csc(L.shape, L.indptr, L.indices, L.data)
    ref = O.nmf_fit(Ao, W0, H0, np.float64, max_iter=6, tol=0.0, graph_W=(oc(LW), 0.05), graph_H=(oc(LH), 0.1))
    assert abs(mod.misc["loss"] - ref.loss) / abs(ref.loss) < 1e-6
    assert np.abs(mod.w - ref.W_T).max() < 1e-6 and np.abs(mod.h.T - ref.H).max() < 1e-6
    with pytest.raises(ValueError):
        nmf.nmf(A, k, seed=7, maxit=2, graph_W=LH, graph_lambda=(0.1, 0.0))
    with pytest.raises(NotImplementedError):
        nmf.nmf(A, k, seed=7, maxit=2, graph_W=LW, graph_lambda=(0.1, 0.0), loss="nb")


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_full_size_c2_properties(dtype):
    """BASELINE configs[1] at full size on the GPU: properties that do not need a full CPU run."""
    import torch
    from rcppml_amd import als, data
    m, n, k = 20000, 100000, 64
    A, _, _ = data.simulate_nmf_sparse(m, n, k, 0.0115, seed=123, device=torch.device("cuda", 0))
    assert abs(A.nnz / (m * float(n)) - 0.01) < 0.001
    At = A.transpose()
    nd = np.float32 if dtype == "f32" else np.float64
    W0, H0 = data.init_factors(42, k, m, n, nd)
    ops = als.HipOps(0, dtype)
    cfg = als.AlsConfig(k=k, max_iter=4, tol=0.0)
    st = als.ShardedALS(ops, als.Comm(None), A, At, W0, H0, cfg)
    tolv = 2e-4 if dtype == "f32" else 1e-10
    losses = []
    for it in range(3):
        losses.append(float(st.step()[0].item()))
        rs = ops.row_norms(st.H, 0).cpu().numpy()
        assert np.allclose(rs, 1.0, atol=1e-4 if dtype == "f32" else 1e-10)           # rows of H sum to 1
    assert np.all(np.isfinite(losses)) and all(losses[i + 1] <= losses[i] * (1 + 1e-5) for i in range(len(losses) - 1))
    W_T, d, H = st.factors()
    assert W_T.min() >= 0 and H.min() >= 0 and np.all(d > 0)
    assert np.allclose(W_T.sum(axis=0), 1.0, atol=1e-4 if dtype == "f32" else 1e-10)
    # --- linearity checksum of the SpMM-like kernel: sum_j B(:,j) = F (A 1)  (exact in exact arithmetic)
    B = ops.rhs(st.A, st.W_T).double().sum(dim=0).cpu().numpy()
    rowsum = np.bincount(A.i, weights=A.x, minlength=m)
    assert np.abs(B - W_T.T @ rowsum).max() / np.abs(B).max() < (1e-4 if dtype == "f32" else 1e-11)
    Bw = ops.rhs(st.At, st.H).double().sum(dim=0).cpu().numpy()
    colsum = np.add.reduceat(A.x, A.p[:-1].astype(np.int64)) * (np.diff(A.p) > 0)
    assert np.abs(Bw - H.T @ colsum).max() / np.abs(Bw).max() < (1e-4 if dtype == "f32" else 1e-11)
    # --- Gram trick loss equals the explicit loss over a column sample extrapolation-free identity:
    #     loss = ||A||^2 - 2 <A, W d H> + <G_W, G_H>_d  -> recompute the three terms in fp64 on the host
    Wd = W_T * d
    cross = 0.0
    for c0 in range(0, n, 10000):
        c1 = min(n, c0 + 10000)
        s, e = A.p[c0], A.p[c1]
        colidx = np.repeat(np.arange(c0, c1), np.diff(A.p[c0:c1 + 1]))
        cross += float(np.sum(A.x[s:e] * np.einsum("ij,ij->i", Wd[A.i[s:e]], H[colidx])))
    recon = float(np.sum((Wd.T @ Wd) * (H.T @ H)))
    true_loss = float(np.sum(A.x ** 2)) - 2 * cross + recon
    assert abs(losses[-1] - true_loss) / true_loss < (5e-4 if dtype == "f32" else 1e-8)


def _pick_columns(A, cols):
    return O.Csc((A.rows, len(cols)), np.concatenate([[0], np.cumsum(np.diff(A.p)[cols])]).astype(np.int32),
                 np.concatenate([A.i[A.p[c]:A.p[c + 1]] for c in cols]),
                 np.concatenate([A.x[A.p[c]:A.p[c + 1]] for c in cols]))


@pytest.mark.parametrize("dtype,tol", [("f64", 1e-9), ("f32", 5e-3)])
def test_full_size_spot_columns_exact(dtype, tol):
    """Two ALS iterations at full size, driven op by op; after each half-update 256 sampled columns of the raw
    NNLS solution (before scaling) are recomputed by the oracle's fused RHS+CD from the device's own inputs
    (G, fixed factor, previous iterate).  Columns are independent given those, so this is an exact check of the
    full-size kernels that costs milliseconds on the CPU.  Tolerance: the warm-start residual b = B - G x_old
    cancels several digits (x_old is the normalised factor, the raw solution is ~1e5 smaller), so agreement is
    eps * |G x_old| / |G x| : ~1e-9 in fp64, ~1e-3..1e-2 in fp32 -- inherent to the reference algorithm."""
    import torch
    from rcppml_amd import als, data
    m, n, k = 20000, 100000, 64
    A, _, _ = data.simulate_nmf_sparse(m, n, k, 0.0115, seed=7, device=torch.device("cuda", 0))
    At = A.transpose()
    nd = np.float32 if dtype == "f32" else np.float64
    W0, H0 = data.init_factors(3, k, m, n, nd)
    ops = als.HipOps(0, dtype)
    cfg = als.AlsConfig(k=k)
    W, H = ops.to_device(W0), ops.to_device(H0)
    Ad, Atd = ops.upload_csc(A), ops.upload_csc(At)
    sums, d = ops.empty((k,)), ops.empty((k,))
    worst = 0.0
    for it in range(2):
        for side in ("H", "W"):
            F, X, csc, host = (W, H, Ad, A) if side == "H" else (H, W, Atd, At)
            G = ops.gram(F, 1e-15, 0.0)
            B = ops.rhs(csc, F)
            X_prev, F_host, G_host = X.cpu().numpy(), F.cpu().numpy(), G.cpu().numpy()
            ops.solve(G, B, X, cfg, side, it > 0)
            X_new = X.cpu().numpy()
            cols = np.sort(np.random.default_rng(10 * it + (side == "W")).choice(host.cols, size=256, replace=False))
            ref = O.fused_cd(_pick_columns(host, cols), F_host, G_host, X_prev[cols], maxit=100, tol=1e-8, warm=(it > 0))
            err = np.abs(X_new[cols] - ref).max() / np.abs(ref).max()
            worst = max(worst, err)
            assert err < tol, (it, side, err)
            assert X_new.min() >= 0
            ops.row_norms(X, 0, out=sums)
            ops.apply_scaling(X, sums, 0, d)
    print("worst relative deviation of sampled columns (%s): %.3e" % (dtype, worst))


def test_full_size_c2_fit_matches_oracle_fp64():
    """The north star's parity bar at FULL size: BASELINE configs[1] (20 000 x 100 000, 1 %, k = 64, CD) through the
    73-pointer fp64 entry vs the CPU oracle's fp64 fit on identical inputs and iteration count: relative loss deviation
    <= 1e-6 (observed 3e-14), d / W / H to 1e-8.  The oracle runs OpenMP over the host's cores (~0.7 s per iteration on
    the 128-thread GPU box)."""
    import torch
    from oracle.oracle import Csc
    from rcppml_amd import _abi, data
    m, n, k, iters = 20000, 100000, 64, 5
    A, _, _ = data.simulate_nmf_sparse(m, n, k, 0.0115, seed=123, device=torch.device("cuda", 0))
    W0, H0 = data.init_factors(42, k, m, n, np.float64)
    W, H = W0.copy(), H0.copy()
    res = _abi.nmf_unified(A.p.astype(np.int32), A.i.astype(np.int32), A.x.astype(np.float64), m, n, k, W, H, entry="double",
                           max_iter=iters, tol=0.0, solver_mode=0)
    assert res["status"] == 0 and res["iter"] == iters
    try:
        O.build(native=True)
        native = True
    except Exception:
        native = False
    ref = O.nmf_fit(Csc((m, n), A.p, A.i, A.x), W0, H0, np.float64, max_iter=iters, tol=0.0, solver_mode=0, threads=0, native=native)
    assert abs(res["loss"] - ref.loss) <= 1e-6 * abs(ref.loss)
    assert np.abs(res["d"] - ref.d).max() <= 1e-8 * np.abs(ref.d).max()
    assert np.abs(W - ref.W_T).max() < 1e-8 and np.abs(H - ref.H).max() < 1e-8


def test_full_size_c2_fp32_entry_loss_within_1e6_of_fp64_oracle():
    """The north star's loss bar for the HEADLINE arithmetic: BASELINE configs[1] at full size through the 73-pointer fp32 entry
    (`rcppml_gpu_nmf_unified_float`, what R's nmf() calls), 25 ALS iterations, against the CPU oracle's fp64 fit from the same
    (fp32-representable) starting factors and the same iteration count: relative deviation of the final loss <= 1e-6
    (bench.py reports the same quantity as `loss_rel_dev_vs_cpu_ref`: 4e-8 .. 5e-8).  fp32 iterates drift from fp64 ones
    factor by factor (a CD early exit taken one sweep apart moves an entry by ~1e-6), so the factors are compared loosely;
    the loss -- what convergence and evaluate() see -- is the bar.  ~25 x 0.7 s of oracle time on the GPU box's host cores."""
    import torch
    from oracle.oracle import Csc
    from rcppml_amd import _abi, data
    m, n, k, iters = 20000, 100000, 64, 25
    A, _, _ = data.simulate_nmf_sparse(m, n, k, 0.0115, seed=123, device=torch.device("cuda", 0))
    W0, H0 = data.init_factors(42, k, m, n, np.float64)
    W0, H0 = W0.astype(np.float32).astype(np.float64), H0.astype(np.float32).astype(np.float64)
    W, H = W0.copy(), H0.copy()
    res = _abi.nmf_unified(A.p.astype(np.int32), A.i.astype(np.int32), A.x.astype(np.float64), m, n, k, W, H, entry="float",
                           max_iter=iters, tol=0.0, solver_mode=0)
    assert res["status"] == 0 and res["iter"] == iters
    try:
        O.build(native=True)
        native = True
    except Exception:
        native = False
    ref = O.nmf_fit(Csc((m, n), A.p, A.i, A.x.astype(np.float32).astype(np.float64)), W0, H0, np.float64, max_iter=iters, tol=0.0,
                    solver_mode=0, threads=0, native=native)
    dev = abs(res["loss"] - ref.loss) / abs(ref.loss)
    print("fp32 entry vs fp64 oracle after %d iterations: loss %.9g vs %.9g, relative deviation %.3e" % (iters, res["loss"], ref.loss, dev))
    assert dev <= 1e-6
    assert np.abs(res["d"] - ref.d).max() <= 1e-3 * np.abs(ref.d).max()
    assert np.abs(W.sum(axis=0) - 1).max() < 1e-4 and np.abs(H.sum(axis=0) - 1).max() < 1e-4        # L1 scaling: columns of W, rows of H


def test_full_size_c4_shard_properties():
    """BASELINE configs[3], one GPU's share at FULL size: 30 000 x 162 500 (1.3 M columns over 8 GPUs), 3 %-dense
    (nnz ~ 1.46e8), k = 128, fp32, CD.  Two ALS iterations through the sharded loop, then the size-independent properties:
    rows of H and columns of W sum to 1, non-negativity, loss finite and non-increasing, the linearity checksum of the
    SpMM-like kernel on both sides (sum_j B(:,j) = F (A 1)), and 128 sampled columns of each half-update against the
    oracle's fused RHS + CD from the device's own inputs (same tolerance reasoning as the C2 spot test)."""
    import torch
    from rcppml_amd import als, data
    m, n, k = 30000, 162500, 128
    A, _, _ = data.simulate_nmf_sparse(m, n, k, 0.031, seed=11, device=torch.device("cuda", 0))
    assert abs(A.nnz / (m * float(n)) - 0.03) < 0.003
    At = A.transpose()
    W0, H0 = data.init_factors(42, k, m, n, np.float32)
    ops = als.HipOps(0, "f32")
    cfg = als.AlsConfig(k=k, max_iter=3, tol=0.0)
    st = als.ShardedALS(ops, als.Comm(None), A, At, W0, H0, cfg)
    assert st.A.get("plans", {}).get(k) is not None and st.At.get("plans", {}).get(k) is not None   # the row-tiled kernel runs
    losses = []
    for it in range(2):
        losses.append(float(st.step()[0].item()))
        rs = ops.row_norms(st.H, 0).cpu().numpy()
        assert np.allclose(rs, 1.0, atol=2e-4)
    assert np.all(np.isfinite(losses)) and losses[1] <= losses[0] * (1 + 1e-5)
    W_T, d, H = st.factors()
    assert W_T.min() >= 0 and H.min() >= 0 and np.all(d > 0)
    assert np.allclose(W_T.sum(axis=0), 1.0, atol=2e-4)
    # linearity checksums (fp32 kernels, fp64 reference): both the planned and the gather form
    rowsum = np.bincount(A.i, weights=A.x, minlength=m)
    colsum = np.add.reduceat(A.x, A.p[:-1].astype(np.int64)) * (np.diff(A.p) > 0)
    for csc, F, ref in ((st.A, st.W_T, W_T.T @ rowsum), (st.At, st.H, H.T @ colsum)):
        B = ops.rhs(csc, F)
        s = B.double().sum(dim=0).cpu().numpy()
        assert np.abs(s - ref).max() / np.abs(s).max() < 2e-4
        B2 = ops.empty(tuple(B.shape))
        ops.ctx.rhs(ops.dt, csc["p"], csc["i"], csc["x"], csc["cols"], F, k, B2)
        assert float((B - B2).abs().max() / B2.abs().max()) < 5e-5          # planned == gather kernel up to summation order
    # sampled columns of both half-updates vs the oracle (third iteration, driven op by op)
    sums, dd = ops.empty((k,)), ops.empty((k,))
    for side in ("H", "W"):
        F, X, csc, host = (st.W_T, st.H, st.A, A) if side == "H" else (st.H, st.W_T, st.At, At)
        G = ops.gram(F, 1e-15, 0.0)
        B = ops.rhs(csc, F)
        X_prev, F_host, G_host = X.cpu().numpy(), F.cpu().numpy(), G.cpu().numpy()
        ops.solve(G, B, X, cfg, side, True)
        X_new = X.cpu().numpy()
        cols = np.sort(np.random.default_rng(5 + (side == "W")).choice(host.cols, size=128, replace=False))
        ref = O.fused_cd(_pick_columns(host, cols), F_host, G_host, X_prev[cols], maxit=100, tol=1e-8, warm=True)
        err = np.abs(X_new[cols] - ref).max() / np.abs(ref).max()
        assert err < 2e-2, (side, err)
        assert X_new.min() >= 0
        ops.row_norms(X, 0, out=sums)
        ops.apply_scaling(X, sums, 0, dd)


@pytest.mark.parametrize("norm", ["L1", "L2", "none"])
@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_fused_tail_loop_equals_separate_kernel_loop_and_plugin(dtype, norm):
    """k = 64 (the shape whose fp32 tail scales inside the Gram's partial-tile kernel), 17 000 x 3 000 so that the W side is
    ranked by sweep counts inside the tail's launches: the harness loop with the fused tail, the same loop with the separate kernels
    (HipOps.fused_tail = False) and the plugin's loop (rcppml_gpu_nmf_ex, which issues the fused calls under a hipGraph from its third
    iteration on) end on the same loss history, d and factors BIT FOR BIT under every norm; and within tolerance of the oracle."""
    from rcppml_amd import als, data, _abi
    A, _, _ = data.simulate_nmf_sparse(17000, 3000, 8, 0.01, seed=77)
    k = 64
    nd = np.float32 if dtype == "f32" else np.float64
    nt = {"L1": 0, "L2": 1, "none": 2}[norm]
    W0, H0 = data.init_factors(9, k, A.rows, A.cols, nd)
    out = []
    for fused in (True, False):
        ops = als.HipOps(0, dtype)
        ops.fused_tail = fused
        cfg = als.AlsConfig(k=k, max_iter=6, tol=0.0, norm_type=nt)
        st = als.ShardedALS(ops, als.Comm(None), A, A.transpose(), W0, H0, cfg)
        res = st.fit()
        W_T, d, H = st.factors()
        out.append((np.array(res["loss_history"]), W_T.copy(), d.copy(), H.copy()))
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)
    W, H = W0.astype(np.float64), H0.astype(np.float64)
    r = _abi.nmf_unified(A.p, A.i, A.x, A.rows, A.cols, k, W, H, entry="ex", precision=_abi.F32 if dtype == "f32" else _abi.F64, max_iter=6, tol=0.0,
                         solver_mode=0, norm_type=nt, sort_model=0, want_history=True)
    assert r["status"] == 0, r.get("error")
    assert np.array_equal(np.asarray(r["loss_history"], np.float64), out[0][0].astype(np.float64))
    assert np.array_equal(W.astype(nd), out[0][1]) and np.array_equal(H.astype(nd), out[0][3]) and np.array_equal(np.asarray(r["d"]).astype(nd), out[0][2])
    ref = O.nmf_fit(_oracle_csc(A), W0, H0, np.float64, max_iter=6, tol=0.0, norm_type=nt, solver_mode=0, sort_model=False)
    # (k = 64 on rank-8 data: dead and duplicated factors make the trajectory sensitive -- fp64 within the north star's 1e-6 of the
    # fp64 oracle (1e-13 under L1 / L2, 8e-8 under "none", the conditioning case documented in test_gpu_reference_suite.py), fp32 within 1e-2)
    rel = abs(out[0][0][-1] - ref.loss) / ref.loss
    assert rel < (1e-6 if dtype == "f64" else 1e-2), rel
