"""The R surface's argument validation, restated on rcppml_amd.nmf() -- the host-side mirror of the reference's interface keeps its
error behaviour, message for message: tests/testthat/test_validation_errors.R (validate_all_penalties, validate_cv_params,
validate_simple_params, validate_mask, validate_graphs, the loss string; R/nmf_validation.R:84-296) and the solver guards of
tests/testthat/test_unsupported_combos.R (R/nmf_thin.R:363-388).  Every check fires before the backend is touched, so this runs
without a GPU."""
import numpy as np
import pytest
import scipy.sparse as sp

from rcppml_amd import nmf as N

A_dense = np.abs(np.random.default_rng(1).standard_normal((50, 20)))
A_sparse = sp.csc_matrix(A_dense * (A_dense > 0.8))


@pytest.mark.parametrize("kw,msg", [
    (dict(L1=1.0), r"range \[0,1\)"), (dict(L1=1.5), r"range \[0,1\)"), (dict(L1=-0.1), r"range \[0,1\)|non-negative"),   # :35-39
    (dict(L1=(0.1, 0.1, 0.1)), "length 1 or 2"),                                                                          # :41-43
    (dict(L2=-0.1), "non-negative|>= 0"), (dict(L2=(0.1, 0.1, 0.1)), "length 1 or 2"),                                    # :45-51
    (dict(L21=-0.1), "non-negative|>= 0"), (dict(angular=-0.1), "non-negative|>= 0"),                                     # :53-59
    (dict(graph_lambda=-0.5), "non-negative"), (dict(upper_bound=-1), "non-negative"),                                    # :61-67
    (dict(test_fraction=1.5), r"range \[0, 1\)"), (dict(test_fraction=-0.1), r"range \[0, 1\)"),                          # :73-78
    (dict(test_fraction="half"), "single numeric"), (dict(test_fraction=(0.1, 0.2)), "single numeric"),
    (dict(patience="five"), "single numeric"), (dict(patience=(1, 2)), "single numeric"),                                 # :80-83
    (dict(sort_model=2), "single logical"), (dict(sort_model="yes"), "single logical"),                                   # :89-92
    (dict(nonneg="yes"), "must be logical"), (dict(nonneg=(True, True, False)), "length 1 or 2"),                         # :94-97
    (dict(mask="foo"), "must be NULL"), (dict(mask="random"), "must be NULL"),                                            # :103-106
    (dict(loss="invalid"), "should be one of"), (dict(loss="MSE"), "should be one of"), (dict(loss="l2"), "should be one of"),   # :134-138
])
def test_validation_errors(kw, msg):
    with pytest.raises(ValueError, match=msg):
        N.nmf(A_dense, 3, **kw)


def test_graph_dimensions_are_validated():
    """test_validation_errors.R:112-130: graph_W must be p x p, graph_H n x n."""
    wrong = sp.identity(5, format="csc")
    with pytest.raises(ValueError, match="must be a .* matrix"):
        N.nmf(A_dense, 3, graph_W=wrong, graph_lambda=(1, 0))
    with pytest.raises(ValueError, match="must be a .* matrix"):
        N.nmf(A_dense, 3, graph_H=wrong, graph_lambda=(0, 1))


def test_solver_guards():
    """test_unsupported_combos.R:44-66: Cholesky with an IRLS distribution or with the robust modifier is rejected with the reference's
    messages; :68-73: the auto solver picks CD for IRLS distributions (select_solver); :4-42, :75-82: zero-inflation is not offered by
    this backend at all (out of scope) and says so."""
    for loss in ("gp", "nb", "gamma"):
        with pytest.raises(ValueError, match="solver='cholesky' is not supported with non-MSE"):
            N.nmf(A_sparse, 2, loss=loss, solver="cholesky")
    with pytest.raises(ValueError, match="solver='cholesky' is not supported with robust"):
        N.nmf(A_sparse, 2, loss="mse", solver="cholesky", robust=True)
    assert N.select_solver("auto", 2, (0.0, 0.0), "gp") == "cd"
    for loss in ("mse", "gamma", "inverse_gaussian", "tweedie", "gp"):
        with pytest.raises(NotImplementedError, match="zero-inflated"):
            N.nmf(A_sparse, 2, loss=loss, zi="row")


def test_initialisation_guards():
    """test_parameters.R:581-620: a custom W_init of the wrong rank and several initialisations under cross-validation are rejected with the
    reference's messages (R/nmf_thin.R:748-751, :829-832) -- before the backend is touched."""
    W3 = np.abs(np.random.default_rng(2).standard_normal((50, 3)))
    with pytest.raises(ValueError, match="Rank mismatch: k=4 specified but custom initialization has rank 3"):
        N.nmf(A_dense, 4, seed=W3)
    with pytest.raises(ValueError, match="Multiple initializations are not compatible with cross-validation"):
        N.nmf(A_dense, 3, seed=[1, 2, 3], test_fraction=0.1)
    with pytest.raises(ValueError, match="Multiple initializations are not compatible with cross-validation"):
        N.nmf(A_dense, [2, 3], seed=[W3, W3])
