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
