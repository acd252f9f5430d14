"""CPU tests of the cross-validation pieces of the oracle (oracle/nmf_oracle.cpp) against brute-force numpy restatements of
the formulas in the reference's nmf/cv_detail.hpp:66-292 and nmf/fit_cv.hpp:1377-1443 -- dense loops over every (i, j), the
mask from the pinned hash (tests/golden: hash_seed*, holdout_seed*), weights from the pinned weight functions."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import random_csc

U64 = (1 << 64) - 1


def holdout_matrix(m, n, frac, seed):
    """rng/rng.hpp:129-170 is_holdout over a whole m x n grid (seed 0 -> 12345, inv_prob = floor(1/frac))."""
    s = (seed & 0xFFFFFFFF) or 12345
    inv = int(1.0 / frac)
    thr = U64 // inv
    return np.array([[O.cv_hash(s, i, j) < thr for j in range(n)] for i in range(m)])


def dense_of(A):
    D = np.zeros((A.rows, A.cols))
    for j in range(A.cols):
        D[A.i[A.p[j]:A.p[j + 1]], j] = A.x[A.p[j]:A.p[j + 1]]
    return D


def sparse_of(D):
    """CSC of the nonzeros of a dense array (O.dense_as_csc keeps explicit zeros, which mask_zeros would count)."""
    p, ii, xx = [0], [], []
    for j in range(D.shape[1]):
        r = np.nonzero(D[:, j])[0].astype(np.int32)
        ii.append(r); xx.append(D[r, j]); p.append(p[-1] + len(r))
    return O.Csc(D.shape, np.asarray(p, np.int32), np.concatenate(ii), np.concatenate(xx))


def weight(loss_type, pred, power):
    if loss_type == 0:
        return 1.0
    if loss_type == 4:
        return O.irls_weight_gp(0.0, pred, 0.0, 1.0)
    mu = max(pred, 1e-10)                                       # math/loss.hpp:270-278 (pinned by test_golden_power_family)
    return {5: None, 6: 1.0 / mu ** 2, 7: 1.0 / mu ** 3, 8: 1.0 / mu ** power}[loss_type]


@pytest.mark.parametrize("transposed", [False, True])
@pytest.mark.parametrize("mask_zeros", [False, True])
@pytest.mark.parametrize("loss_type", [0, 4, 6, 8])
def test_cv_irls_half_update_is_the_weighted_train_solve(loss_type, mask_zeros, transposed):
    m, n, k, frac, seed = 23, 17, 4, 0.2, 77
    rng = np.random.default_rng(5)
    A = random_csc(m, n, 0.4, 3, values="poisson")
    D = dense_of(A)
    held = holdout_matrix(m, n, frac, seed)
    if transposed:                                              # W side: columns of A^T, factor H
        At = sparse_of(D.T)
        data, Dd, hd = At, D.T, held.T
    else:
        data, Dd, hd = A, D, held
    F = rng.uniform(0.2, 1.0, size=(Dd.shape[0], k))
    X0 = rng.uniform(0.2, 1.0, size=(Dd.shape[1], k))
    G_add = 0.01 * np.eye(k)
    got = O.cv_irls_half_update(data, F, X0, k, frac, seed, loss_type, G_add=G_add, mask_zeros=mask_zeros, transposed=transposed,
                                solver_mode=1, irls_max_iter=1, power=1.5)
    want = np.zeros_like(X0)
    for j in range(Dd.shape[1]):
        G = G_add + 1e-15 * np.eye(k)
        b = np.zeros(k)
        for r in range(Dd.shape[0]):
            if hd[r, j] or (mask_zeros and Dd[r, j] == 0):
                continue
            w = weight(loss_type, float(F[r] @ X0[j]), 1.5)
            G = G + w * np.outer(F[r], F[r])
            b = b + w * Dd[r, j] * F[r]
        want[j] = np.maximum(np.linalg.solve(G, b), 0)
    assert np.allclose(got, want, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("transposed", [False, True])
def test_cv_irls_with_unit_weights_is_the_mse_cv_update(transposed):
    """loss_type 0, robust off: G_w = sum_train f f^T must agree with the MSE path's G - sum_test f f^T."""
    m, n, k, frac, seed = 31, 19, 5, 0.1, 9
    rng = np.random.default_rng(2)
    A = random_csc(m, n, 0.3, 11)
    data = sparse_of(dense_of(A).T) if transposed else A
    F = rng.uniform(size=(data.rows, k))
    X0 = rng.uniform(size=(data.cols, k))
    G = F.T @ F + 1e-15 * np.eye(k)
    a = O.cv_half_update(data, F, G, X0, k, frac, seed, transposed=transposed, cd_maxit=30)
    b = O.cv_irls_half_update(data, F, X0, k, frac, seed, 0, transposed=transposed, cd_maxit=30, irls_max_iter=1)
    assert np.allclose(a, b, rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("mask_zeros", [False, True])
@pytest.mark.parametrize("loss_type", [4, 6])
def test_cv_explicit_loss_splits_train_and_test(loss_type, mask_zeros):
    m, n, k, frac, seed = 19, 13, 3, 0.25, 4
    rng = np.random.default_rng(8)
    A = random_csc(m, n, 0.5, 21, values="poisson")
    D = dense_of(A)
    held = holdout_matrix(m, n, frac, seed)
    W_T, H, d = rng.uniform(0.2, 1, (m, k)), rng.uniform(0.2, 1, (n, k)), rng.uniform(0.5, 2, k)
    theta = rng.uniform(0, 0.3, m) if loss_type == 4 else np.zeros(m)
    tr, ntr, te, nte = O.cv_explicit_loss(A, W_T, d, H, frac, seed, loss_type, theta=theta, mask_zeros=mask_zeros)
    considered = (D != 0) if mask_zeros else np.ones_like(held)
    assert ntr == int((considered & ~held).sum()) and nte == int((considered & held).sum())
    # with a holdout fraction of 0 everything is training: the two sums of the masked call add up to that total
    tr_all, n_all, te0, nte0 = O.cv_explicit_loss(A, W_T, d, H, 0.0, seed, loss_type, theta=theta, mask_zeros=mask_zeros)
    assert nte0 == 0 and te0 == 0 and n_all == ntr + nte
    assert np.isclose(tr + te, tr_all, rtol=1e-10)


def test_cv_gp_theta_without_holdout_is_the_plain_update():
    m, n, k = 17, 29, 3
    rng = np.random.default_rng(1)
    A = random_csc(m, n, 0.4, 6, values="poisson")
    W_T, H, d = rng.uniform(0.2, 1, (m, k)), rng.uniform(0.2, 1, (n, k)), rng.uniform(0.5, 2, k)
    th0 = np.full(m, 0.1)
    all_entries = O.cv_gp_theta_update(A, W_T, d, H, th0, 0.0, 3)
    masked = O.cv_gp_theta_update(A, W_T, d, H, th0, 0.2, 3)
    assert np.all(np.isfinite(all_entries)) and np.all(all_entries >= 0) and np.all(all_entries <= 5.0)
    assert not np.array_equal(all_entries, masked)             # the holdout changes sum_y / sum_s of most rows
    # rows without a held-out pair keep exactly the unmasked estimate
    held = holdout_matrix(m, n, 0.2, 3)
    untouched = ~held.any(axis=1)
    assert np.array_equal(all_entries[untouched], masked[untouched])


@pytest.mark.parametrize("mask_zeros", [False, True])
def test_cv_fit_with_user_mask_excludes_masked_entries_everywhere(mask_zeros):
    """nmf_fit_cv with a user mask (fit_cv.hpp:327-331, :491-501, :779-790; cv_detail.hpp:433-505; losses :1377-1443), one MSE
    iteration restated over dense arrays: per column the excluded rows are the test rows (held-out nonzeros with mask_zeros, every
    held-out row without) UNION the user-masked rows; b sums the nonzeros that are neither held out nor masked; the Gram loses
    f f^T of every excluded row (a masked ZERO entry too); CD starts from the current column with b used as it is; train and test
    losses are explicit means over the entries that are not user-masked (nonzeros only with mask_zeros)."""
    m, n, k, frac, seed, sweeps = 21, 16, 3, 0.25, 5, 6
    A = random_csc(m, n, 0.45, 8, values="poisson")
    M = random_csc(m, n, 0.2, 9)
    D, Mk = dense_of(A), dense_of(M) != 0
    assert (Mk & (D == 0)).any() and (Mk & (D != 0)).any()          # masked zeros and masked nonzeros both occur
    held = holdout_matrix(m, n, frac, seed)
    W0, H0 = O.init_factors(3, k, m, n, np.float64)
    fit = O.nmf_fit_cv(A, W0, H0, np.float64, max_iter=1, tol=0.0, cd_maxit=sweeps, holdout_fraction=frac, cv_seed=seed,
                       mask_zeros=mask_zeros, mask=M, L1=(0.01, 0.02))

    def half(F, Dd, hd, mk, X, l1):
        G = F.T @ F + 2e-15 * np.eye(k)
        out = np.empty_like(X)
        for j in range(Dd.shape[1]):
            nz = Dd[:, j] != 0
            test = hd[:, j] & nz if mask_zeros else hd[:, j]
            excl = test | mk[:, j]
            train_nz = nz & ~hd[:, j] & ~mk[:, j]
            b = F[train_nz].T @ Dd[train_nz, j]
            Gl = G - F[excl].T @ F[excl]
            out[j], _, _ = O.cd_col(Gl, b, X[j].copy(), L1=l1, maxit=sweeps, tol=0.0)
        s = np.abs(out).sum(axis=0) + 1e-15
        return out / s, s

    H, d = half(W0, D, held, Mk, H0, 0.02)
    W, d = half(H, D.T, held.T, Mk.T, W0, 0.01)
    pred = (W * d) @ H.T
    use = ~Mk & ((D != 0) if mask_zeros else np.ones_like(Mk))
    se = (D - pred) ** 2
    want_test, want_train = se[use & held].mean(), se[use & ~held].mean()
    assert np.allclose(fit.W_T, W, rtol=0, atol=1e-10) and np.allclose(fit.H, H * d, rtol=0, atol=1e-10)
    assert abs(fit.test_loss - want_test) <= 1e-10 * want_test and abs(fit.train_loss - want_train) <= 1e-10 * want_train
    # an empty mask is the unmasked fit's factors (its losses come from the Gram-trick branch: equal to rounding)
    E = O.Csc((m, n), np.zeros(n + 1, np.int32), np.zeros(0, np.int32), np.zeros(0))
    a = O.nmf_fit_cv(A, W0, H0, np.float64, max_iter=2, tol=0.0, cd_maxit=sweeps, holdout_fraction=frac, cv_seed=seed, mask_zeros=mask_zeros, mask=E)
    b = O.nmf_fit_cv(A, W0, H0, np.float64, max_iter=2, tol=0.0, cd_maxit=sweeps, holdout_fraction=frac, cv_seed=seed, mask_zeros=mask_zeros)
    assert np.array_equal(a.W_T, b.W_T) and np.array_equal(a.H, b.H) and abs(a.test_loss - b.test_loss) <= 1e-12 * b.test_loss
    if not mask_zeros:      # (with mask_zeros the Gram-trick branch's total includes the zero entries' predictions, :1500-1530; the explicit one does not)
        assert abs(a.train_loss - b.train_loss) <= 1e-9 * b.train_loss


@pytest.mark.parametrize("loss_type", [4, 6])
def test_cv_irls_fit_with_user_mask_drops_masked_entries_from_the_weighted_solves(loss_type):
    """IRLS CV with a user mask: the masked rows join is_test (cv_detail.hpp:116-117 via gram_rows), i.e. the training entries of a
    column are the not-held-out, not-masked ones; one half-update restated with the pinned weights."""
    m, n, k, frac, seed = 19, 14, 3, 0.2, 11
    A = random_csc(m, n, 0.5, 4, values="poisson")
    M = random_csc(m, n, 0.2, 6)
    D, Mk = dense_of(A), dense_of(M) != 0
    held = holdout_matrix(m, n, frac, seed)
    W0, H0 = O.init_factors(8, k, m, n, np.float64)
    fit = O.nmf_fit_cv(A, W0, H0, np.float64, max_iter=1, tol=0.0, cd_maxit=5, holdout_fraction=frac, cv_seed=seed, mask_zeros=False,
                       mask=M, loss_type=loss_type, irls_max_iter=1, dispersion_mode=0)
    # H after the first half-update = fit.H / d restated: one IRLS pass from H0
    Hb = np.empty_like(H0)
    for j in range(n):
        tr = ~held[:, j] & ~Mk[:, j]
        x = H0[j].copy()
        pred = W0[tr] @ x
        w = np.array([weight(loss_type, p, 1.5) for p in pred])
        Gw = (W0[tr] * w[:, None]).T @ W0[tr] + 1e-15 * np.eye(k)
        bw = W0[tr].T @ (w * D[tr, j])
        Hb[j], _, _ = O.cd_col(Gw, bw, x, maxit=5, tol=0.0)
    s = np.abs(Hb).sum(axis=0) + 1e-15
    # the returned H carries the FINAL d (after the W update); compare the normalised shape of the H half-update
    Hn = fit.H / (np.abs(fit.H).sum(axis=0))
    assert np.allclose(Hn, (Hb / s) / np.abs(Hb / s).sum(axis=0), rtol=0, atol=1e-9)
