"""Host-side input code (rcppml_amd/data.py): the vectorised SplitMix64 stream equals the sequential one,
R's runif clone matches published values, CSC helpers, the simulateNMF restatement."""
import numpy as np

from oracle import oracle as O
from rcppml_amd import data


def test_vectorised_splitmix_equals_sequential():
    for dt in (np.float32, np.float64):
        W, H = data.init_factors(42, 7, 31, 53, dt)
        Wo, Ho = O.init_factors(42, 7, 31, 53, dt)
        assert np.array_equal(W, Wo) and np.array_equal(H, Ho)
        _, H2 = data.init_factors(42, 7, 31, 20, dt, col_offset=11, n_total=53)
        assert np.array_equal(H2, Ho[11:31])
    assert list(data.splitmix64_raw(1234567, 0, 3)) == [6457827717110365317, 3203168211198807973, 9817491932198370423]
    assert np.array_equal(data.splitmix64_raw(0, 0, 4), data.splitmix64_raw(12345, 0, 4))


def test_r_runif_published_values():
    # `set.seed(42); runif(5)` etc. as printed by any R session (R is not in the reference tree)
    assert np.allclose(data.r_runif(42, 5), [0.9148060, 0.9370754, 0.2861395, 0.8304476, 0.6417455], atol=5e-8)
    assert np.allclose(data.r_runif(1, 3), [0.2655087, 0.3721239, 0.5728534], atol=5e-8)
    assert np.allclose(data.r_runif(123, 3), [0.2875775, 0.7883051, 0.4089769], atol=5e-8)


def test_csc_transpose_and_slice():
    A, _, _ = data.simulate_nmf_sparse(120, 200, 4, 0.05, seed=3)
    Ao = O.Csc(A.shape, A.p, A.i, A.x)
    At, Ato = A.transpose(), Ao.transpose()
    assert np.array_equal(At.p, Ato.p) and np.array_equal(At.i, Ato.i) and np.array_equal(At.x, Ato.x)
    S = A.col_slice(50, 120)
    assert np.array_equal(S.to_scipy().toarray(), A.to_scipy().toarray()[:, 50:120])
    B = data.CSC.from_scipy(A.to_scipy())
    assert np.array_equal(B.i, A.i) and np.array_equal(B.p, A.p)


def test_simulate_structure_and_shards():
    A, w, h = data.simulate_nmf_sparse(400, 600, 8, 0.05, seed=5)
    assert A.shape == (400, 600) and np.all(A.x > 0)
    assert 0.5 * 0.05 < A.nnz / (400 * 600) <= 0.05 * 1.05          # clamped-to-zero entries are dropped
    assert np.allclose(w.sum(axis=0), 1) and np.allclose(h.sum(axis=1), 1)
    for j in range(A.cols):
        r = A.i[A.p[j]:A.p[j + 1]]
        assert np.all(np.diff(r) > 0)                               # sorted, unique rows
    # column shards of the same wide matrix use the same factors
    _, w2, h2 = data.simulate_nmf_sparse(400, 300, 8, 0.05, seed=5, col_offset=300, ncol_total=600)
    assert np.array_equal(w, w2) and np.array_equal(h2, h[:, 300:])
    nb, _, _ = data.simulate_nb_counts(200, 300, 4, density=0.1, seed=2)
    assert np.all(nb.x >= 1) and np.all(nb.x == np.round(nb.x))
