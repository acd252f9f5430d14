"""CPU tests of the round-5 oracle pieces -- the combinations the product used to hand back -- against brute-force numpy
restatements of the reference formulas (dense loops over every (i, j); the per-element loss terms are the oracle's own,
which tests/test_oracle_ref.py pins bit for bit to the reference's math/loss.hpp):

  * an explicit mask together with a distribution loss: the reference tests `use_mask` BEFORE `requires_irls()`
    (nmf/fit_cpu.hpp:560-564, :799-803), so the half-updates are the masked MSE solves, and the loss is masked_loss with
    compute_loss(a, pred, config.loss) at its default theta = 0 (nmf/masked_nnls.hpp:250-282, :277; fit_cpu.hpp:1685-1690);
  * dispersion = "per_col": one NB size / GP theta / phi per COLUMN of A (fit_cpu.hpp:300-301, :319-320, :341-342, :578-583,
    :820-830, :1009-1083, :1103-1162, :1570-1611; explicit_loss.hpp:59-71)."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O
from tests.util import random_csc


def _olib():
    L = O.lib()
    L.oracle_loss_nb_f64.restype = C.c_double
    L.oracle_loss_gp_f64.restype = C.c_double
    L.oracle_loss_dev_f64.restype = C.c_double
    return L


def _term(L, loss_type, y, pred, theta, power):
    if loss_type == 5:
        return L.oracle_loss_nb_f64(C.c_double(y), C.c_double(pred), C.c_double(theta))
    if loss_type == 4:
        return L.oracle_loss_gp_f64(C.c_double(y), C.c_double(pred), C.c_double(theta))
    return L.oracle_loss_dev_f64(C.c_int(loss_type), C.c_double(y), C.c_double(pred), C.c_double(power))


def dense_of(A):
    D = np.zeros((A.rows, A.cols))
    for j in range(A.cols):
        D[A.i[A.p[j]:A.p[j + 1]], j] = A.x[A.p[j]:A.p[j + 1]]
    return D


def _problem(m=31, n=23, seed=3):
    A = random_csc(m, n, 0.35, seed, values="poisson")
    M = random_csc(m, n, 0.15, seed + 100)
    return A, M


@pytest.mark.parametrize("loss_type,disp,power", [(5, 2, 1.5), (5, 0, 1.5), (4, 2, 1.5), (6, 0, 1.5), (7, 2, 1.5), (8, 2, 1.3)])
def test_mask_with_distribution_loss_is_masked_mse_solves_and_theta0_loss(loss_type, disp, power):
    """One ALS iteration from a fixed start: (a) the factors equal those of the MSE fit with the same mask -- the half-updates do
    not look at the loss; (b) the loss is the sum over the unmasked NONZEROS of the loss term at theta = 0 (not at the fitted
    dispersion); (c) the dispersion vector is still updated from ALL nonzeros, as without a mask."""
    A, M = _problem()
    k = 4
    W0, H0 = O.init_factors(7, k, A.rows, A.cols, np.float64)
    kw = dict(max_iter=1, tol=0.0, mask=M, L1=(0.01, 0.02), sort_model=False)
    mse = O.nmf_fit(A, W0, H0, np.float64, **kw)
    fit = O.nmf_fit(A, W0, H0, np.float64, loss_type=loss_type, dispersion_mode=disp, tweedie_power=power, **kw)
    assert np.array_equal(fit.W_T, mse.W_T) and np.array_equal(fit.H, mse.H) and np.array_equal(fit.d, mse.d)
    L = _olib()
    Mk = dense_of(M) != 0
    pred = (fit.W_T * fit.d) @ fit.H.T
    tot = 0.0
    for j in range(A.cols):
        for t in range(A.p[j], A.p[j + 1]):
            i = A.i[t]
            if not Mk[i, j]:
                tot += _term(L, loss_type, A.x[t], pred[i, j], 0.0, power)
    assert abs(fit.loss - tot) <= 1e-10 * abs(tot)
    # the same fit without the mask: a different loss value (the mask and the theta = 0 both matter) ...
    free = O.nmf_fit(A, W0, H0, np.float64, max_iter=1, tol=0.0, L1=(0.01, 0.02), sort_model=False, loss_type=loss_type,
                     dispersion_mode=disp, tweedie_power=power)
    assert abs(free.loss - fit.loss) > 1e-6 * abs(fit.loss)
    # ... and the dispersion of the masked fit is the unmasked estimator applied to the masked fit's factors
    if disp != 0:
        start = {5: 10.0, 4: 0.1}.get(loss_type, 1.0)
        if loss_type == 5:
            ref = O.nb_size_update(A, fit.W_T, fit.H, fit.d, np.full(A.rows, start), dispersion_mode=disp)
        else:
            ref = O.dispersion_update(loss_type, A, fit.W_T, fit.H, fit.d, np.full(A.rows, start), dispersion_mode=disp, power=power,
                                      hi=5.0 if loss_type == 4 else 1e4)
        assert np.allclose(fit.theta, ref, rtol=1e-12, atol=0)


def test_mask_with_robust_mse_ignores_the_robust_modifier():
    """robust_delta > 0 makes requires_irls() true, but the mask branch is taken first and masked_loss calls compute_loss (not
    compute_robust_loss): the fit is the plain masked MSE fit."""
    A, M = _problem(seed=5)
    W0, H0 = O.init_factors(2, 3, A.rows, A.cols, np.float64)
    a = O.nmf_fit(A, W0, H0, np.float64, max_iter=3, tol=0.0, mask=M)
    b = O.nmf_fit(A, W0, H0, np.float64, max_iter=3, tol=0.0, mask=M, robust_delta=1.345)
    assert np.array_equal(a.W_T, b.W_T) and np.array_equal(a.H, b.H) and a.loss == b.loss


def test_nb_size_per_col_is_the_column_moment_estimator():
    """fit_cpu.hpp:1103-1162 by a dense loop: r_j = S_i mu^2 / S_i ((y - mu)^2 - mu), nonzero predictions floored at 1e-10."""
    A, _ = _problem(m=27, n=19, seed=11)
    k = 3
    rng = np.random.default_rng(0)
    W_T, H, d = rng.uniform(0.1, 1.0, (A.rows, k)), rng.uniform(0.1, 1.0, (A.cols, k)), rng.uniform(0.5, 2.0, k)
    got = O.nb_size_update(A, W_T, H, d, np.full(A.cols, 10.0), dispersion_mode=3)
    D = dense_of(A)
    mu = (W_T * d) @ H.T
    nz = D != 0
    mu_c = np.where(nz, np.maximum(mu, 1e-10), mu)
    s_mu2 = (mu_c ** 2).sum(axis=0)
    s_exc = ((D - mu_c) ** 2 - mu_c).sum(axis=0)
    want = np.where((s_exc > 1e-10) & (s_mu2 > 1e-10), np.clip(s_mu2 / np.where(s_exc > 1e-10, s_exc, 1.0), 0.01, 1e6), 1e6)
    assert got.shape == (A.cols,) and np.allclose(got, want, rtol=1e-10, atol=0)


@pytest.mark.parametrize("loss_type,power", [(6, 2.0), (7, 3.0), (8, 1.4)])
def test_phi_per_col_is_the_column_pearson_estimator(loss_type, power):
    A, _ = _problem(m=21, n=17, seed=13)
    k = 3
    rng = np.random.default_rng(1)
    W_T, H, d = rng.uniform(0.1, 1.0, (A.rows, k)), rng.uniform(0.1, 1.0, (A.cols, k)), rng.uniform(0.5, 2.0, k)
    got = O.dispersion_update(loss_type, A, W_T, H, d, np.ones(A.cols), dispersion_mode=3, power=power)
    D = dense_of(A)
    mu = np.maximum((W_T * d) @ H.T, 1e-10)
    pos = D > 0
    pear = np.where(pos, (D - mu) ** 2 / np.maximum(mu ** power, 1e-20), 0.0)
    cnt = pos.sum(axis=0)
    want = np.where(cnt > 0, np.clip(pear.sum(axis=0) / np.maximum(cnt, 1), 1e-6, 1e4), 1.0)
    assert np.allclose(got, want, rtol=1e-10, atol=0)


def test_gp_theta_per_col_is_the_row_update_of_the_transposed_problem():
    """The PER_COL MM update (fit_cpu.hpp:1009-1083) is the PER_ROW update (:914-1008) with the roles of rows and columns swapped;
    s_ij = (w_i * d) . h_j either way, so the transposed problem with d folded into W reproduces it to rounding."""
    A, _ = _problem(m=19, n=25, seed=17)
    k = 3
    rng = np.random.default_rng(2)
    W_T, H, d = rng.uniform(0.1, 1.0, (A.rows, k)), rng.uniform(0.1, 1.0, (A.cols, k)), rng.uniform(0.5, 2.0, k)
    got = O.dispersion_update(4, A, W_T, H, d, np.full(A.cols, 0.1), dispersion_mode=3, hi=5.0)
    D = dense_of(A)
    At = O.dense_as_csc(D.T)
    keep = At.x != 0                                           # dense_as_csc keeps explicit zeros: drop them
    p = np.zeros(At.cols + 1, np.int32)
    for j in range(At.cols):
        p[j + 1] = p[j] + int(keep[At.p[j]:At.p[j + 1]].sum())
    At = O.Csc((At.rows, At.cols), p, At.i[keep], At.x[keep])
    want = O.dispersion_update(4, At, H, W_T * d, np.ones(k), np.full(A.cols, 0.1), dispersion_mode=2, hi=5.0)
    assert np.allclose(got, want, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("loss_type", [5, 4])
def test_per_col_fit_uses_the_column_dispersion_in_solves_and_loss(loss_type):
    """Whole fit with dispersion = per_col: theta has n entries; the loss is the explicit loss with theta_j of the nonzero's column
    (explicit_loss.hpp:59-71); with dispersion held constant (init = min = max) per_col, per_row and global are the same fit."""
    A, _ = _problem(m=29, n=21, seed=19)
    k = 3
    W0, H0 = O.init_factors(4, k, A.rows, A.cols, np.float64)
    fit = O.nmf_fit(A, W0, H0, np.float64, max_iter=2, tol=0.0, loss_type=loss_type, dispersion_mode=3, sort_model=False)
    assert fit.theta.shape == (A.cols,)
    L = _olib()
    pred = (fit.W_T * fit.d) @ fit.H.T
    tot = sum(_term(L, loss_type, A.x[t], pred[A.i[t], j], fit.theta[j], 1.5) for j in range(A.cols) for t in range(A.p[j], A.p[j + 1]))
    assert abs(fit.loss - tot) <= 1e-10 * abs(tot)
    if loss_type == 5:
        const = dict(nb_size=(7.0, 7.0, 7.0))
        a = O.nmf_fit(A, W0, H0, np.float64, max_iter=2, tol=0.0, loss_type=5, dispersion_mode=3, **const)
        b = O.nmf_fit(A, W0, H0, np.float64, max_iter=2, tol=0.0, loss_type=5, dispersion_mode=2, **const)
        assert np.allclose(a.W_T, b.W_T, rtol=0, atol=1e-12) and np.allclose(a.H, b.H, rtol=0, atol=1e-12)
        assert abs(a.loss - b.loss) <= 1e-12 * abs(b.loss)


# ---------------------------------------------------------------------------------------------------------------------------
# Dense input with a distribution loss: the reference's DENSE branches (nnls_batch_irls.hpp:376-450, :525-555; fit_cpu.hpp:953-968,
# :1041-1053, :1137-1148, :1226-1238), which the oracle takes with dense_input=True on a CSC that stores every entry
# ---------------------------------------------------------------------------------------------------------------------------
def _dense_problem(m=19, n=13, k=3, seed=23):
    rng = np.random.default_rng(seed)
    D = rng.poisson(1.4, (m, n)).astype(np.float64)
    D[rng.uniform(size=(m, n)) < 0.3] = 0.0
    F = rng.uniform(0.1, 1.0, (m, k))
    return D, O.dense_as_csc(D), F, rng


def test_dense_irls_half_update_weights_every_row_and_builds_the_gram_from_nothing():
    """irls_nnls_col_dense: x starts at 0; per pass recon = F x, w_i = weight(a_i - recon_i, recon_i) for ALL rows (zeros of A
    included), G_w = F^T diag(w) F (+ L2), b_w = F^T (w * a), CD warm-started by b_w - G_w x; stop on the relative change."""
    D, A, F, rng = _dense_problem()
    m, n = D.shape
    k = F.shape[1]
    theta = rng.uniform(2.0, 30.0, m)
    L = O.lib()
    L.oracle_irls_weight_nb_f64.restype = C.c_double
    L1, L2, cd_maxit, irls_it, irls_tol = 0.02, 0.05, 9, 4, 1e-4
    got = O.irls(5, A, F, np.full((k, k), 7.0), k, L1=L1, L2=L2, cd_maxit=cd_maxit, irls_max_iter=irls_it, irls_tol=irls_tol,
                 theta_row=theta, dense_input=True)                     # G_base is not read: any matrix gives the same result
    want = np.zeros((n, k))
    for j in range(n):
        x = np.zeros(k)
        for _ in range(irls_it):
            recon = F @ x
            w = np.array([L.oracle_irls_weight_nb_f64(C.c_double(recon[i]), C.c_double(theta[i])) for i in range(m)])
            Gw = (F * w[:, None]).T @ F + L2 * np.eye(k)
            bw = F.T @ (w * D[:, j])
            xo = x.copy()
            x, _, _ = O.cd_col(Gw, bw - Gw @ xo, xo, L1=L1, maxit=cd_maxit, tol=0.0)
            if (np.abs(x - xo) / (np.abs(xo) + 1e-12)).max() < irls_tol:
                break
        want[j] = x
    assert np.allclose(got, want, rtol=1e-10, atol=1e-13)
    # the sparse form on the same all-entries CSC is the same solve up to the eps of G_base and rounding
    sp = O.irls(5, A, F, F.T @ F + 1e-15 * np.eye(k), k, L1=L1, L2=L2, cd_maxit=cd_maxit, irls_max_iter=irls_it, irls_tol=irls_tol,
                theta_row=theta)
    assert np.allclose(got, sp, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("mode", [2, 3, 1])
def test_dense_nb_size_update_floors_every_prediction(mode):
    """:1137-1148 (PER_COL) / :1226-1238 (PER_ROW, GLOBAL = median): sums over ALL entries of mu^2 and (y - mu)^2 - mu with
    mu = max(prediction, 1e-10) -- no Gram-trick totals as in the sparse branch."""
    D, A, W_T, rng = _dense_problem(seed=29)
    m, n = D.shape
    k = W_T.shape[1]
    H, d = rng.uniform(0.1, 1.0, (n, k)), rng.uniform(0.5, 2.0, k)
    H[:, 0] = 0.0; W_T[3] = 0.0                                          # a row of exact-zero predictions: the floor is live
    length = n if mode == 3 else m
    got = O.nb_size_update(A, W_T, H, d, np.full(length, 10.0), dispersion_mode=mode, dense_input=True)
    mu = np.maximum((W_T * d) @ H.T, 1e-10)
    ax = 0 if mode == 3 else 1
    s_mu2, s_exc = (mu ** 2).sum(axis=ax), ((D - mu) ** 2 - mu).sum(axis=ax)
    want = np.where((s_exc > 1e-10) & (s_mu2 > 1e-10), np.clip(s_mu2 / np.where(s_exc > 1e-10, s_exc, 1.0), 0.01, 1e6), 1e6)
    if mode == 1:
        want = np.full(m, np.partition(want, m // 2)[m // 2])
    assert np.allclose(got, want, rtol=1e-10, atol=0)


@pytest.mark.parametrize("mode", [2, 3])
def test_dense_gp_theta_update_sums_the_floored_predictions(mode):
    """:953-968 / :1041-1053: sum_s is the per-entry sum of max(prediction, 1e-10) (the sparse branch takes it from the factors' sums,
    unfloored); the MM passes read the entries with y >= 1 only.  Five passes of the closed-form quadratic, restated in numpy."""
    D, A, W_T, rng = _dense_problem(seed=31)
    m, n = D.shape
    k = W_T.shape[1]
    H, d = rng.uniform(0.1, 1.0, (n, k)), rng.uniform(0.5, 2.0, k)
    W_T[5] = 0.0
    length = n if mode == 3 else m
    got = O.dispersion_update(4, A, W_T, H, d, np.full(length, 0.1), dispersion_mode=mode, hi=5.0, dense_input=True)
    s = np.maximum((W_T * d) @ H.T, 1e-10)
    ax = 0 if mode == 3 else 1
    sum_y, sum_s, n_nz = D.sum(axis=ax), s.sum(axis=ax), (D >= 1).sum(axis=ax)
    th = np.full(length, 0.1)
    for _ in range(5):
        thb = th[None, :] if mode == 3 else th[:, None]
        eta1 = s / np.maximum(s + thb * D, 1e-10)
        big = D >= 1
        alpha = np.where(big, (D - 1) * eta1, 0.0).sum(axis=ax) + n_nz
        gamma = np.where(big, (D - 1) * (1 - eta1), 0.0).sum(axis=ax)
        beta = (sum_y - sum_s) - gamma + alpha
        disc = beta ** 2 + 4 * alpha * gamma
        ok = (alpha > 1e-15) & (disc > 0)
        new = np.where(ok, (-beta + np.sqrt(np.where(ok, disc, 1.0))) / (2 * np.where(ok, alpha, 1.0)), th)
        th = np.where(ok & (new >= 0), np.minimum(new, 5.0), th)
    assert np.allclose(got, th, rtol=1e-9, atol=1e-12)


def test_dense_input_fit_with_nb_loss_is_the_all_entries_fit():
    """Whole fit, dense input + NB: theta has m entries, the loss is the explicit loss over ALL m n entries (explicit_loss.hpp:86-107)
    and the fit agrees with the sparse-branch fit on the same all-entries CSC to rounding (the branches differ by floors that are not
    live here and by the eps of G_base)."""
    D, A, _, _ = _dense_problem(m=21, n=16, seed=37)
    k = 3
    W0, H0 = O.init_factors(6, k, A.rows, A.cols, np.float64)
    fit = O.nmf_fit(A, W0, H0, np.float64, max_iter=3, tol=0.0, loss_type=5, dense_input=True, sort_model=False)
    L = _olib()
    pred = (fit.W_T * fit.d) @ fit.H.T
    tot = sum(_term(L, 5, D[i, j], pred[i, j], fit.theta[i], 1.5) for i in range(A.rows) for j in range(A.cols))
    assert fit.theta.shape == (A.rows,) and abs(fit.loss - tot) <= 1e-10 * abs(tot)
    sp = O.nmf_fit(A, W0, H0, np.float64, max_iter=3, tol=0.0, loss_type=5, unfused=True, sort_model=False)
    assert abs(sp.loss - fit.loss) <= 1e-9 * abs(fit.loss) and np.allclose(sp.H, fit.H, rtol=0, atol=1e-9)
