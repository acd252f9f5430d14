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
    for kw in (dict(loss_type=3), dict(loss_type=5, solver_mode=1), dict(loss_type=5, projective=1), dict(robust_delta=1.0, ortho_H=0.1),
               dict(loss_type=4, dispersion_mode=7), dict(symmetric=1), dict(solver_mode=2), dict(ortho_H=-0.1)):
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


DENSE_LOSS_CASES = [
    ("nb_per_row", dict(loss_type=5, dispersion_mode=2), "counts"),
    ("nb_per_col", dict(loss_type=5, dispersion_mode=3), "counts"),
    ("nb_global", dict(loss_type=5, dispersion_mode=1), "counts"),
    ("nb_none_l1", dict(loss_type=5, dispersion_mode=0, L1_H=0.02, L1_W=0.01, L2_H=0.03), "counts"),
    ("gp_per_row", dict(loss_type=4, dispersion_mode=2), "counts"),
    ("gp_per_col", dict(loss_type=4, dispersion_mode=3), "counts"),
    ("kl", dict(loss_type=4, dispersion_mode=0), "counts"),
    ("gamma", dict(loss_type=6, dispersion_mode=2), "positive"),
    ("inverse_gaussian_per_col", dict(loss_type=7, dispersion_mode=3), "positive"),
    ("tweedie", dict(loss_type=8, dispersion_mode=2, tweedie_power=1.3), "counts"),
    ("robust_mse", dict(loss_type=0, robust_delta=1.345), "counts"),
    ("robust_nb", dict(loss_type=5, dispersion_mode=2, robust_delta=1.0), "counts"),
]


def _loss_problem(kind, m=46, n=37, r=3, seed=5):
    rng = np.random.default_rng(seed)
    mu = rng.gamma(2.0, 1.0, (m, r)) @ rng.gamma(2.0, 0.5, (r, n))
    if kind == "counts":
        M = rng.negative_binomial(4.0, 4.0 / (4.0 + mu)).astype(np.float64)     # overdispersed counts, ~ 10 % exact zeros: they carry weight in the dense solves
    else:
        M = mu * rng.gamma(8.0, 1.0 / 8.0, (m, n)) + 0.05       # strictly positive (the Gamma / inverse Gaussian deviances need y > 0)
    return M


@pytest.mark.parametrize("name,kw,kind", DENSE_LOSS_CASES, ids=[c[0] for c in DENSE_LOSS_CASES])
def test_dense_fit_with_distribution_losses(name, kw, kind):
    """Dense input under every distribution loss the sparse entries take -- the reference's nnls_batch_irls_dense half-updates
    (nmf/fit_cpu.hpp:607-614, :855-863; primitives/cpu/nnls_batch_irls.hpp:376-450: EVERY entry weighted, zeros included, each batch
    from H = 0), the dense branches of the dispersion updates (:953-968, :1041-1053, :1137-1148, :1226-1238, :1576-1650) and
    explicit_loss_dense (explicit_loss.hpp:86-107) -- through the 50-pointer fp64 entry against the oracle's dense_input fit:
    loss 1e-6 relative, factors 1e-6 (L1-normalised), dispersion vector 1e-5 relative; theta has m entries, n under per_col."""
    from rcppml_amd import _abi
    M = _loss_problem(kind)
    m, n = M.shape
    k = 4
    A = O.dense_as_csc(M)
    W0, H0 = O.init_factors(9, k, m, n, np.float64)
    okw = dict(loss_type=kw.get("loss_type", 0), dispersion_mode=kw.get("dispersion_mode", 2), tweedie_power=kw.get("tweedie_power", 1.5),
               robust_delta=kw.get("robust_delta", 0.0), L1=(kw.get("L1_W", 0.0), kw.get("L1_H", 0.0)), L2=(kw.get("L2_W", 0.0), kw.get("L2_H", 0.0)))
    ref = O.nmf_fit(A, W0, H0, np.float64, max_iter=4, tol=0.0, cd_maxit=20, dense_input=True, **okw)
    W, H = W0.copy(), H0.copy()
    res = _abi.nmf_dense(M, k, W, H, entry="double", max_iter=4, tol=0.0, cd_maxit=20, **kw)
    assert res["status"] == 0, res["error"]
    assert res["iter"] == ref.iter
    assert abs(res["loss"] - ref.loss) <= 1e-6 * abs(ref.loss), (res["loss"], ref.loss)
    assert np.abs(res["d"] - ref.d).max() <= 1e-6 * np.abs(ref.d).max()
    assert np.abs(W - ref.W_T).max() <= 1e-6 and np.abs(H - ref.H).max() <= 1e-6
    want_len = n if kw.get("dispersion_mode", 2) == 3 else m
    assert res["theta"].shape == (want_len,) and ref.theta.shape == (want_len,)
    assert np.abs(res["theta"] - ref.theta).max() <= 1e-5 * max(np.abs(ref.theta).max(), 1e-300)
    # not the sparse entry on the same numbers: there the zeros of A have weight 1 and drop out of the loss
    if kind == "counts" and kw.get("loss_type", 0) != 0:
        sp = O.nmf_fit(O.Csc.from_dense(M), W0, H0, np.float64, max_iter=4, tol=0.0, cd_maxit=20, **okw)
        assert abs(sp.loss - ref.loss) > 1e-4 * abs(ref.loss)


def test_dense_fit_with_nb_loss_fp32_entry_and_surface():
    """The fp32 dense entry under NB (loss 2e-3 against the fp64 oracle; fp32 forms G_w as G_base + sum (w - 1) f f^T, the
    cancellation costs ~1e-6 of the Gram), and nmf(<ndarray>, loss = "nb") reaching the dense entry with theta in misc."""
    from rcppml_amd import _abi, nmf as N
    M = _loss_problem("counts", seed=6)
    m, n = M.shape
    k = 3
    W0, H0 = O.init_factors(2, k, m, n, np.float64)
    W0, H0 = W0.astype(np.float32).astype(np.float64), H0.astype(np.float32).astype(np.float64)
    ref = O.nmf_fit(O.dense_as_csc(M), W0, H0, np.float64, max_iter=4, tol=0.0, cd_maxit=20, loss_type=5, dense_input=True)
    W, H = W0.copy(), H0.copy()
    res = _abi.nmf_dense(M, k, W, H, entry="float", max_iter=4, tol=0.0, cd_maxit=20, loss_type=5)
    assert res["status"] == 0, res["error"]
    assert abs(res["loss"] - ref.loss) <= 2e-3 * abs(ref.loss)
    assert np.abs(W - ref.W_T).max() <= 5e-3 and np.abs(H - ref.H).max() <= 5e-3
    mod = N.nmf(M, k, loss="nb", seed=4, maxit=3, tol=0.0, precision="fp64")
    assert mod.misc["input"] == "dense" and mod.misc["theta"].shape == (m,) and np.isfinite(mod.misc["loss"])
    assert mod.misc["entry"] == "rcppml_gpu_nmf_dense_unified_double"
    # arguments the dense entry has no slot for must not switch a dense matrix to the sparse-input model (zeros unweighted) silently
    for extra in (dict(sort_model=False), dict(cd_tol=1e-6)):
        with pytest.raises(NotImplementedError, match="dense input"):
            N.nmf(M, k, loss="nb", seed=4, maxit=2, precision="fp64", **extra)
        with pytest.raises(NotImplementedError, match="dense input"):
            N.nmf(M, k, robust=True, seed=4, maxit=2, precision="fp64", **extra)
    # ... while plain MSE keeps both routes (same model either way; misc says which entry ran)
    a = N.nmf(M, k, seed=4, maxit=2, precision="fp64", sort_model=False)
    assert a.misc["entry"] == "rcppml_gpu_nmf_ex"


def test_reference_dense_irls_properties():
    """The properties the reference's own tests/testthat/test_dense_irls.R asserts for dense input under non-MSE losses, on the GPU
    path: robust = TRUE / "mae" and GP fits end with a finite loss and non-negative factors (:7-45), dense and sparse robust fits of
    the same data end within an order of magnitude (:47-62), robust with L1 + L2 (:84-95), dense GP under cross-validation (:97-107),
    and the Gamma / inverse Gaussian / Tweedie fits improve on their second iteration's loss (:113-160; the dense boundary returns
    no history, so the second iteration's loss comes from a two-iteration fit of the same seed -- the fits are deterministic)."""
    from rcppml_amd import nmf as N
    import scipy.sparse as sp
    rng = np.random.default_rng(42)
    A = np.abs(rng.standard_normal((50, 40))) + 0.1
    for kw in (dict(robust=True), dict(robust="mae"), dict(loss="gp"), dict(robust=True, L1=(0.01, 0.01), L2=(0.01, 0.01))):
        mod = N.nmf(A + (0.4 if kw.get("loss") == "gp" else 0.0), 3, maxit=50, tol=1e-5, seed=42, precision="fp64", **kw)
        assert mod.misc["input"] == "dense" and np.isfinite(mod.misc["loss"]) and mod.w.min() >= 0 and mod.h.min() >= 0, kw
    S = np.abs(rng.standard_normal((30, 3))) @ np.abs(rng.standard_normal((3, 25))) + 0.1 * np.abs(rng.standard_normal((30, 25)))
    md = N.nmf(S, 3, robust=True, maxit=30, seed=42, precision="fp64")
    ms = N.nmf(sp.csc_matrix(S), 3, robust=True, maxit=30, seed=42, precision="fp64")
    assert abs(md.misc["loss"] - ms.misc["loss"]) / max(md.misc["loss"], ms.misc["loss"]) < 1.0
    cv = N.nmf(A + 0.4, 3, test_fraction=0.1, loss="gp", maxit=20, seed=42, precision="fp64")
    assert np.isfinite(cv.misc["loss"]) and np.isfinite(cv.misc["test_loss"])
    G = np.maximum(np.abs(2.0 + 0.5 * rng.standard_normal((50, 40))), 1e-8)
    for loss, extra in (("gamma", {}), ("inverse_gaussian", {}), ("tweedie", dict(tweedie_power=1.5))):
        full = N.nmf(G, 3, loss=loss, dispersion="per_row", maxit=30, tol=1e-6, seed=42, precision="fp64", **extra)
        two = N.nmf(G, 3, loss=loss, dispersion="per_row", maxit=2, tol=0.0, seed=42, precision="fp64", **extra)
        assert np.isfinite(full.misc["loss"]) and full.w.min() >= 0, loss
        assert full.misc["loss"] < two.misc["loss"], (loss, full.misc["loss"], two.misc["loss"])


def test_reference_na_handling_properties():
    """tests/testthat/test_masking.R:240-300 on the GPU path: NA values in a dense matrix are detected with the reference's warning,
    the fit succeeds with finite loss and non-negative factors, explicit mask = "NA" works, and the region without NAs is fitted well
    (R/nmf_thin.R:686-696: the NAs become zeros, mask <- "NA", whose mask matrix is empty, R/nmf_validation.R:253-257)."""
    from rcppml_amd import nmf as N
    rng = np.random.default_rng(42)
    A = rng.uniform(size=(100, 50))
    A[:5, :5] = np.nan
    with pytest.warns(UserWarning, match="Detected 25 NA values"):
        mod = N.nmf(A, 3, maxit=20, seed=1, precision="fp64")
    assert np.isfinite(mod.misc["loss"]) and mod.w.min() >= 0 and mod.h.min() >= 0
    B = rng.uniform(size=(80, 40)); B[:3, :3] = np.nan
    with pytest.warns(UserWarning):
        m2 = N.nmf(B, 2, mask="NA", maxit=15, seed=1, precision="fp64")
    assert np.isfinite(m2.misc["loss"])
    Cm = rng.uniform(size=(60, 40))
    orig = Cm[9:15, 9:15].copy()
    Cm[:5, :5] = np.nan
    with pytest.warns(UserWarning):
        m3 = N.nmf(Cm, 3, maxit=30, seed=1, precision="fp64")
    recon = (m3.w * m3.d) @ m3.h
    assert np.mean((orig - recon[9:15, 9:15]) ** 2) < 0.5
    # the same zeros, passed as zeros, give the same fit: the "mask" of the reference is empty
    Z = np.where(np.isnan(Cm), 0.0, Cm)
    m4 = N.nmf(Z, 3, maxit=30, seed=1, precision="fp64")
    assert np.array_equal(m3.w, m4.w) and m3.misc["loss"] == m4.misc["loss"]
