"""Host-side mirror of the reference's R surface for the ALS-NNLS path: nmf(), nnls(), predict(), evaluate().

Same argument names, meaning, defaults and error behaviour as the R functions (R/nmf_thin.R:219-229,
R/solve.R:84, R/predict_nmf.R:48, R/nmf_methods.R:356), restricted to what the MI355X plugin implements
(MSE and NB loss, CD or Cholesky+clip, L1/L2/upper bounds, non-negativity flags, explicit mask, L1/L2/no normalisation).
All compute goes through RcppML_gpu.so (rcppml_amd._abi); anything else raises -- there is no CPU path here.
"""
import numpy as np

from . import _abi
from .data import CSC, r_runif, splitmix64_uniform

_LOSSES = ("mse", "gp", "nb", "gamma", "inverse_gaussian", "tweedie")


class NMFModel:
    """S4 class `nmf` of the reference (R/nmf_methods.R:18-30): w (m x k), d (k), h (k x n), misc."""

    def __init__(self, w, d, h, misc):
        self.w, self.d, self.h, self.misc = w, d, h, misc

    def __repr__(self):
        return "<nmf model: %d x %d, k=%d, iter=%s, loss=%.6g>" % (self.w.shape[0], self.h.shape[1], self.d.shape[0],
                                                                   self.misc.get("iter"), self.misc.get("loss", float("nan")))


def _as_csc(data):
    if isinstance(data, CSC):
        return data
    if hasattr(data, "tocsc"):
        return CSC.from_scipy(data)
    a = np.asarray(data, dtype=np.float64)
    if a.ndim != 2:
        raise ValueError("data must be a matrix")
    import scipy.sparse as sp
    return CSC.from_scipy(sp.csc_matrix(a))


def _pair(v, name):
    v = np.atleast_1d(np.asarray(v, dtype=np.float64))
    if v.shape[0] == 1:
        v = np.repeat(v, 2)
    if v.shape[0] != 2:
        raise ValueError("'%s' must be length 1 or 2 for c(w, h)" % name)          # R/nmf_validation.R:90-95 validate_penalty
    return float(v[0]), float(v[1])


def gpu_available():
    """R/gpu_backend.R:68-125."""
    try:
        return len(_abi.detect()) > 0
    except _abi.BackendError:
        return False


def select_solver(solver, k, L1, loss="mse", use_gpu=True):
    """Auto solver rule of R/nmf_thin.R:363-388."""
    if solver != "auto":
        if solver not in ("cd", "cholesky"):
            raise ValueError("solver must be 'auto', 'cd' or 'cholesky'")
        return solver
    if loss != "mse":
        return "cd"
    if use_gpu:
        return "cd" if k <= 32 else "cholesky"
    return "cholesky" if (k < 32 and all(v == 0 for v in L1)) else "cd"


class CVTable(list):
    """The reference's `nmfCrossValidate` data.frame (R/nmf_thin.R:1074-1090): a list of rows {rep, k, train_mse, test_mse, best_iter,
    total_iter, mean_theta}; `col(name)` returns one column."""

    def col(self, name):
        return [r[name] for r in self]


def _cv_across_ranks(data, ranks, seed, cv_seed, test_fraction, kw):
    """R/nmf_thin.R:803-812, :1034-1090: test_fraction defaults to 0.1, the replicates' CV seeds default to the fit's seed, and every
    (replicate, rank) fit starts from the reference's own initialize_factors stream SplitMix64((cv_seed + rank) mod INT_MAX) -- one
    stream fills W_T and continues into H (nmf/nmf_init.hpp:166-182) -- with the replicate's holdout pattern."""
    from .data import init_factors
    if isinstance(seed, (list, tuple)) or (seed is not None and np.ndim(seed) == 2):
        raise ValueError("Multiple initializations are not compatible with cross-validation. Use a single seed or matrix.")
    if not test_fraction:
        test_fraction = 0.1
    A = _as_csc(data)
    m, n = A.shape
    seed_int = int(np.random.SeedSequence().generate_state(1)[0] % (2 ** 31 - 1)) + 1 if seed is None else int(np.atleast_1d(seed)[0])
    cv_seeds = [seed_int] if cv_seed is None else [int(v) for v in np.atleast_1d(cv_seed)]
    rows = CVTable()
    for rep, cs in enumerate(cv_seeds, start=1):
        for rank in ranks:
            init_seed = int((cs + rank) % (2 ** 31 - 1))
            # (the stream is drawn in the compute Scalar: initialize_factors<float> when the fit runs in fp32, as the single fit does)
            W0, H0 = init_factors(init_seed, rank, m, n, np.float32 if kw.get("precision", "fp32") == "fp32" else np.float64)
            W0, H0 = W0.astype(np.float64), H0.astype(np.float64)
            mod = nmf(data, rank, seed=W0, h_init=H0.T, test_fraction=test_fraction, cv_seed=cs, **kw)
            th = mod.misc.get("theta")
            rows.append(dict(rep=rep, k=rank, train_mse=mod.misc["loss"], test_mse=mod.misc.get("best_test_loss", mod.misc["test_loss"]),
                             best_iter=int(mod.misc.get("best_iter", 0)) + 1, total_iter=mod.misc["iter"],
                             mean_theta=float(np.mean(th)) if th is not None else float("nan")))
    return rows


def nmf(data, k, tol=1e-4, maxit=100, L1=(0.0, 0.0), L2=(0.0, 0.0), seed=None, mask=None, loss="mse",
        nonneg=(True, True), test_fraction=0, verbose=False, projective=False, symmetric=False, zi="none",
        robust=False, *, solver="auto", upper_bound=(0.0, 0.0), cd_maxit=100, cd_tol=1e-8, norm="L1", sort_model=True,
        patience=5, h_init=None, precision="fp32", resource="gpu", dispersion="per_row", irls_max_iter=5, irls_tol=1e-4,
        nb_size_init=10.0, nb_size_max=1e6, nb_size_min=0.01, tweedie_power=1.5, L21=(0.0, 0.0), angular=(0.0, 0.0),
        graph_W=None, graph_H=None, graph_lambda=(0.0, 0.0), target_H=None, target_lambda=0.0, theta_init=0.1, theta_max=5.0,
        theta_min=0.0, cv_seed=None):
    """Non-negative matrix factorisation A ~ w diag(d) h by alternating NNLS on the MI355X.

    `L1`, `L2`, `upper_bound`, `nonneg` are c(w, h) pairs (src/RcppFunctions_nmf.cpp:59-62).
    `seed`: None / int -> W_init = matrix(runif(m*k), m, k) after set.seed(seed) (R/nmf_thin.R:790-797) and
    H from SplitMix64(seed) (nmf/fit_cpu.hpp:200-207); or an m x k (or k x m) matrix used as W_init.
    `precision`: "fp32" is what the reference computes in (F1); "fp64" is the parity mode.
    `robust`: False | True (Huber delta 1.345) | "mae" (1e-4) | positive delta (R/nmf_thin.R:343-352).
    `graph_W` (m x m) / `graph_H` (n x n): sparse graph Laplacians, `graph_lambda` = c(w, h) (R/nmf_thin.R:67-68, 500-506).
    `theta_init`, `theta_max`, `theta_min`: the GP dispersion bounds R takes through `...` (R/nmf_thin.R:246-248).
    `test_fraction` > 0 with loss in {gp, nb, gamma, inverse_gaussian, tweedie} or `robust`: the CV fit with per-column weighted
    Grams over the training entries (nmf/fit_cv.hpp:446-456, :670-689).
    `target_H` (k x n) with `target_lambda` (a scalar is the H side, as R/nmf_thin.R:646-648; > 0 enrichment, < 0 PROJ_ADV).
    `k` a vector: cross-validation across ranks (R/nmf_thin.R:803-812, :1034-1090) -- returns a CVTable (the reference's
    nmfCrossValidate data.frame: one row per replicate and rank); `cv_seed`: the holdout pattern's seed(s), default the fit's seed.
    `seed` a vector of integers or a list of W matrices: one fit per initialisation, the one with the lowest loss is returned
    (R/nmf_thin.R:744-786, :828-917; misc$all_init_losses / best_init_idx).
    """
    # ---- several ranks: cross-validation table (R/nmf_thin.R:803-812, 1034-1090)
    if np.ndim(k) == 1 and len(k) > 1:
        kw = dict(locals())
        for drop in ("data", "k", "seed", "cv_seed", "test_fraction", "h_init"):
            kw.pop(drop)
        return _cv_across_ranks(data, [int(v) for v in k], seed, cv_seed, test_fraction, kw)
    if np.ndim(k) == 1:
        k = int(k[0])
    # ---- several initialisations: best of (R/nmf_thin.R:744-786, 828-917)
    multi = None
    if isinstance(seed, (list, tuple)) and len(seed) > 0 and all(np.ndim(v) == 2 for v in seed):
        multi = list(seed)
    elif seed is not None and np.ndim(seed) == 1 and len(seed) > 1:
        multi = [int(v) for v in seed]
    elif seed is not None and np.ndim(seed) == 1:
        seed = int(seed[0])
    if multi is not None and len(multi) > 1:
        if test_fraction and test_fraction > 0:
            raise ValueError("Multiple initializations are not compatible with cross-validation. Use a single seed or matrix.")
        kw = dict(locals())
        for drop in ("data", "k", "seed", "multi"):
            kw.pop(drop)
        fits = [nmf(data, k, seed=sd, **kw) for sd in multi]
        losses = [f.misc["loss"] for f in fits]
        best = int(np.argmin(losses))
        fits[best].misc["all_init_losses"] = np.asarray(losses)
        fits[best].misc["best_init_idx"] = best
        return fits[best]
    if multi is not None:
        seed = multi[0]
    if loss not in _LOSSES:
        raise ValueError("'arg' should be one of %s" % ", ".join(repr(x) for x in _LOSSES))
    if zi != "none":
        raise NotImplementedError("zero-inflated losses are not implemented by the MI355X backend")
    if isinstance(robust, (bool, np.bool_)):
        robust_delta = 1.345 if robust else 0.0
    elif isinstance(robust, str) and robust.lower() == "mae":
        robust_delta = 1e-4
    elif isinstance(robust, (int, float)):
        robust_delta = float(robust)
    else:
        raise ValueError("'robust' must be FALSE, TRUE, 'mae', or a positive numeric Huber delta.")
    if robust_delta < 0:
        raise ValueError("'robust' must be FALSE, TRUE, 'mae', or a positive numeric Huber delta.")
    if robust_delta > 0 and solver == "cholesky":
        raise ValueError("solver='cholesky' is not supported with robust IRLS (robust_delta > 0). Use solver='cd' for robust estimation.")
    if solver == "cholesky" and loss != "mse" and robust_delta == 0:                 # R/nmf_thin.R:378-382
        raise ValueError("solver='cholesky' is not supported with non-MSE distributions (got '%s'). Use solver='cd' for IRLS-based distributions." % loss)
    # R/nmf_validation.R:148-160 validate_cv_params, :280-296 validate_simple_params, :87-118 validate_mask -- message for message
    if isinstance(test_fraction, (str, bytes)) or np.ndim(test_fraction) != 0:
        raise ValueError("'test_fraction' must be a single numeric value")
    if not (0 <= float(test_fraction) < 1):
        raise ValueError("'test_fraction' must be in the range [0, 1)")
    if isinstance(patience, (str, bytes)) or np.ndim(patience) != 0:
        raise ValueError("'patience' must be a single numeric value")
    if not isinstance(sort_model, (bool, np.bool_)):
        raise ValueError("'sort_model' must be a single logical value")
    _nn = np.atleast_1d(np.asarray(nonneg))
    if _nn.dtype != np.bool_:
        raise ValueError("'nonneg' must be logical")
    if _nn.shape[0] not in (1, 2):
        raise ValueError("'nonneg' must be length 1 or 2 with no NA values")
    if isinstance(mask, str) and mask not in ("zeros", "NA"):
        raise ValueError("'mask' must be NULL, 'zeros', 'NA', a matrix, or list(\"zeros\", <matrix>)")
    if dispersion not in ("none", "global", "per_row", "per_col"):
        raise ValueError("dispersion must be 'none', 'global', 'per_row' or 'per_col'")
    if dispersion == "per_col" and test_fraction and test_fraction > 0:
        raise NotImplementedError("dispersion = 'per_col' under cross-validation is not implemented by the MI355X backend")
    if symmetric and (robust_delta > 0 or loss != "mse" or projective or (test_fraction and test_fraction > 0)
                      or (mask is not None and not isinstance(mask, str))):
        raise NotImplementedError("symmetric NMF is implemented for the plain MSE path")
    if projective and (robust_delta > 0 or loss != "mse" or (test_fraction and test_fraction > 0) or (mask is not None and not isinstance(mask, str))):
        raise NotImplementedError("projective NMF is implemented for the plain MSE path")
    cv = bool(test_fraction) and test_fraction > 0
    if cv:
        if not (0 < test_fraction < 1):
            raise ValueError("'test_fraction' must be in the range [0, 1)")
    if resource != "gpu":
        raise ValueError("rcppml_amd has no CPU path; resource must be 'gpu'")
    dense_in = isinstance(data, np.ndarray) and data.ndim == 2        # a base R matrix: the reference's dense path
    # NA values (R/nmf_validation.R:44-51, :69-76 warn; R/nmf_thin.R:686-696 set them to 0 and mask <- "NA"; validate_mask :253-257
    # then returns an EMPTY mask matrix with a `mask_na` flag nothing downstream reads): the fit sees zeros there, as in the reference
    if dense_in and np.isnan(data).any():
        import warnings
        n_na = int(np.isnan(data).sum())
        warnings.warn("Detected %d NA values (%.2f%% of data). Automatically creating mask for missing values." % (n_na, 100.0 * n_na / data.size))
        data = np.where(np.isnan(data), 0.0, data)
        if mask is None:
            mask = "NA"
    A = _as_csc(data)
    if not dense_in and np.isnan(A.x).any():          # any sparse container (scipy of every format, the repo's CSC): checked on the converted values
        import warnings
        n_na = int(np.isnan(A.x).sum())
        warnings.warn("Detected %d NA values (%.2f%% of data). Automatically creating mask for missing values." % (n_na, 100.0 * n_na / (A.shape[0] * A.shape[1])))
        from .data import CSC as _CSC
        A = _CSC(A.shape, A.p, A.i, np.where(np.isnan(A.x), 0.0, A.x))
        if mask is None:
            mask = "NA"
    m, n = A.shape
    if symmetric and m != n:
        raise ValueError('symmetric = TRUE requires a square matrix')
    k = int(k)
    if k < 1:
        raise ValueError("k must be a positive integer")
    L1w, L1h = _pair(L1, "L1")
    L2w, L2h = _pair(L2, "L2")
    L21w, L21h = _pair(L21, "L21")
    angw, angh = _pair(angular, "angular")
    ubw, ubh = _pair(upper_bound, "upper_bound")
    # R/nmf_validation.R:108-141 validate_all_penalties, message for message
    if max(L1w, L1h) >= 1 or min(L1w, L1h) < 0:
        raise ValueError("L1 penalties must be strictly in the range [0,1)")
    if min(L2w, L2h) < 0:
        raise ValueError("L2 penalties must be strictly >= 0")
    if min(L21w, L21h) < 0:
        raise ValueError("L21 penalties must be strictly >= 0")
    if min(angw, angh) < 0:
        raise ValueError("angular penalties must be strictly >= 0")
    if min(ubw, ubh) < 0:
        raise ValueError("'upper_bound' values must be non-negative")
    if norm not in ("L1", "L2", "none", "None"):
        raise ValueError("'arg' should be one of 'L1', 'L2', 'none'")          # match.arg(norm), R/nmf_thin.R
    nn = np.atleast_1d(nonneg)
    nnw, nnh = (bool(nn[0]), bool(nn[-1]))
    norm_type = {"L1": 0, "L2": 1, "none": 2, "None": 2}[norm]
    solver = select_solver(solver, k, (L1w, L1h), loss if robust_delta == 0 else "robust", use_gpu=True)
    glw, glh = _pair(graph_lambda, "graph_lambda")
    if min(glw, glh) < 0:
        raise ValueError("'graph_lambda' values must be non-negative")
    graph_args = {}
    for name, g, dim, lam in (("graph_W", graph_W, m, glw), ("graph_H", graph_H, n, glh)):
        if g is None or lam <= 0:
            continue
        Lg = _as_csc(g)
        if Lg.shape != (dim, dim):                                                   # R/nmf_validation.R:171-205 validate_graphs
            raise ValueError("'%s' must be a %d x %d matrix (%s)" % (name, dim, dim, "p x p where p is number of features" if name == "graph_W"
                                                                       else "n x n where n is number of samples"))
        graph_args[name] = (Lg.p, Lg.i, Lg.x, float(lam))
    if graph_args and (loss != "mse" or robust_delta > 0 or (mask is not None and not isinstance(mask, str)) or k > 128):
        raise NotImplementedError("graph regularisation is implemented for the plain MSE path, k <= 128")
    # ---- initialisation
    if seed is None:
        seed_int = int(np.random.SeedSequence().generate_state(1)[0] % (2 ** 31 - 1)) + 1
        W0 = r_runif(seed_int, m * k).reshape(k, m).T.copy()          # matrix(runif(m*k), m, k): column-major fill
    elif np.ndim(seed) == 2:
        s = np.asarray(seed, dtype=np.float64)
        if s.shape == (m, k):
            W0 = s.copy()
        elif s.shape == (k, m):
            W0 = s.T.copy()
        else:
            actual_k = s.shape[1] if s.shape[0] == m else (s.shape[0] if s.shape[1] == m else None)
            if actual_k is not None and actual_k != k:                   # R/nmf_thin.R:760-763
                raise ValueError("Rank mismatch: k=%d specified but custom initialization has rank %d." % (k, actual_k))
            raise ValueError("Custom init matrix dimensions incompatible with data")
        seed_int = int(abs(int(np.sum(s * 1e6) % (2 ** 31 - 1))))
    else:
        seed_int = int(seed)
        W0 = r_runif(seed_int, m * k).reshape(k, m).T.copy()
    W_T = np.ascontiguousarray(W0, dtype=np.float64)                   # (m, k) C-order == column-major k x m
    if h_init is not None:
        H = np.ascontiguousarray(np.asarray(h_init, dtype=np.float64).T)   # h_init is k x n
        if H.shape != (n, k):
            raise ValueError("h_init must be k x n")
    else:
        sdt = np.float32 if precision == "fp32" else np.float64       # the reference fills H in its Scalar type
        H = splitmix64_uniform(seed_int & 0xFFFFFFFF, 0, k * n, sdt).astype(np.float64).reshape(n, k)
    mask_arg = None
    if mask is not None:
        if isinstance(mask, str):
            if mask == "zeros":
                pass            # fit-time no-op in the reference; only evaluate() honours it
            elif mask == "NA":
                pass            # validate_mask: an empty mask matrix (R/nmf_validation.R:256-257); the NAs are zeros by now
            else:
                raise NotImplementedError("mask='%s' is not implemented" % mask)
        else:
            M = _as_csc(mask)
            if M.shape != A.shape:
                raise ValueError("mask dimensions must match data")
            mask_arg = (M.p, M.i)
    if cv:
        # nmf/fit_cv.hpp: speckled holdout mask (seed = the fit's seed), per-column Gram correction, early stopping on the
        # test loss with `patience`; mask = "zeros" <=> mask_zeros (only nonzeros can be held out)
        if max(L21w, L21h, angw, angh, ubw, ubh) > 0:
            raise NotImplementedError("cross-validation with L21 / angular / upper bounds is not implemented by the MI355X backend")
        irls_cv = loss != "mse" or robust_delta > 0
        if irls_cv and graph_args:
            raise NotImplementedError("graph regularisation is implemented for the MSE cross-validation path")
        cv_kw = {}
        if irls_cv and dispersion not in ("none", "global", "per_row"):
            raise NotImplementedError("dispersion = %r is not implemented for cross-validation by the MI355X backend (none / global / per_row)" % (dispersion,))
        if irls_cv:          # nmf/fit_cv.hpp:446-456, :670-689: per-column weighted Grams over the training entries
            cv_kw = dict(loss_type={"mse": 0, "gp": 4, "nb": 5, "gamma": 6, "inverse_gaussian": 7, "tweedie": 8}[loss],
                         irls_max_iter=int(irls_max_iter), irls_tol=float(irls_tol),
                         dispersion_mode={"none": 0, "global": 1, "per_row": 2}[dispersion], gp_theta=(float(theta_init), float(theta_max)),
                         tweedie_power=float(tweedie_power), robust_delta=float(robust_delta))
        if mask_arg is not None:          # fit_cv.hpp:327-331: the user mask under CV (build-defined entry rcppml_gpu_nmf_cv_masked_ex)
            if graph_args:
                raise NotImplementedError("graph regularisation together with a mask matrix is not implemented for cross-validation")
            cv_kw["mask"] = mask_arg
        res = _abi.nmf_cv(A.p, A.i, A.x, m, n, k, W_T, H, entry="irls_ex" if irls_cv else "ex", **cv_kw, max_iter=int(maxit), tol=float(tol), L1_H=L1h, L1_W=L1w,
                          L2_H=L2h, L2_W=L2w, cd_maxit=int(cd_maxit), verbose=int(verbose), seed=seed_int & 0x7FFFFFFF,
                          holdout_fraction=float(test_fraction), cv_seed=(int(cv_seed) if cv_seed is not None else seed_int) & 0x7FFFFFFF,
                          mask_zeros=int(isinstance(mask, str) and mask == "zeros"), nonneg_W=int(nnw), nonneg_H=int(nnh),
                          norm_type=norm_type, solver_mode=0 if solver == "cd" else 1, sort_model=int(sort_model),
                          precision=_abi.F32 if precision == "fp32" else _abi.F64, cv_patience=int(patience), **graph_args)
        if res["status"] != 0:
            raise _abi.BackendError("GPU CV NMF failed: %s" % res.get("error"))
        misc = dict(iter=res["iter"], converged=res["converged"], loss=res["train_loss"], test_loss=res["test_loss"],
                    best_test_loss=res["best_test_loss"], best_iter=res["best_iter"], loss_history=res.get("train_history"),
                    test_loss_history=res.get("test_history"), solver=solver, solver_mode=0 if solver == "cd" else 1,
                    L1=(L1w, L1h), L2=(L2w, L2h), seed=seed_int, cv_seed=(int(cv_seed) if cv_seed is not None else seed_int) & 0x7FFFFFFF,
                    precision=precision, resource="gpu", loss_type=loss, test_fraction=float(test_fraction))
        if irls_cv and loss == "gp":
            misc["theta"] = res.get("theta")
        return NMFModel(w=W_T.copy(), d=res["d"], h=H.T.copy(), misc=misc)
    dense_entry_ok = mask_arg is None and not graph_args and sort_model and target_H is None and float(cd_tol) == 1e-8
    if dense_in and (loss != "mse" or robust_delta > 0) and not dense_entry_ok:
        # a dense matrix under a distribution loss / robust modifier is weighted ENTRY BY ENTRY, zeros included (nnls_batch_irls_dense,
        # fit_cpu.hpp:607-614); the sparse entry gives zeros weight 1 -- another model.  Unrelated arguments must not switch between the
        # two silently: what the dense entry has no slot for is refused here
        raise NotImplementedError("dense input with loss = %r%s: the dense entry (rcppml_gpu_nmf_dense_unified_*) carries no mask, graph, target, "
                                  "cd_tol or sort_model = FALSE; pass a sparse matrix for the sparse-input semantics (zeros unweighted)"
                                  % (loss, " + robust" if robust_delta > 0 else ""))
    if dense_in and dense_entry_ok:
        # dense input -> rcppml_gpu_nmf_dense_unified_* (GEMM right-hand sides, the reference's unfused update order; under a
        # distribution loss the dense IRLS solves, which weight EVERY entry -- the sparse entry gives zeros weight 1).
        # The dense ABI (bridge_nmf.hpp:101-126) has no slot for cd_tol (the plugin uses the reference default 1e-8) and
        # returns no loss history: any other cd_tol keeps the sparse entry, which honours it; misc says which entry ran.
        res = _abi.nmf_dense(np.asarray(data, np.float64), k, W_T, H, entry="float" if precision == "fp32" else "double",
                             max_iter=int(maxit), tol=float(tol), L1_H=L1h, L1_W=L1w, L2_H=L2h, L2_W=L2w, L21_H=L21h, L21_W=L21w,
                             ortho_H=angh, ortho_W=angw, ub_H=ubh, ub_W=ubw, cd_maxit=int(cd_maxit), verbose=int(verbose),
                             seed=seed_int & 0x7FFFFFFF, patience=int(patience), nonneg_W=int(nnw), nonneg_H=int(nnh),
                             norm_type=norm_type, projective=int(bool(projective)), symmetric=int(bool(symmetric)),
                             solver_mode=0 if solver == "cd" else 1,
                             loss_type={"mse": 0, "gp": 4, "nb": 5, "gamma": 6, "inverse_gaussian": 7, "tweedie": 8}[loss],
                             robust_delta=robust_delta, irls_max_iter=int(irls_max_iter), irls_tol=float(irls_tol),
                             dispersion_mode={"none": 0, "global": 1, "per_row": 2, "per_col": 3}[dispersion],
                             gp_theta_init=float(theta_init), gp_theta_max=float(theta_max), nb_size_init=nb_size_init,
                             nb_size_max=nb_size_max, nb_size_min=nb_size_min, tweedie_power=float(tweedie_power))
        if res["status"] != 0:
            raise _abi.BackendError("GPU dense NMF failed: %s" % res.get("error"))
        misc = dict(tol=res["tol"], iter=res["iter"], loss=res["loss"], converged=res["converged"], solver=solver,
                    solver_mode=0 if solver == "cd" else 1, L1=(L1w, L1h), L2=(L2w, L2h), seed=seed_int, precision=precision,
                    resource="gpu", loss_type=loss, input="dense", loss_history=None,
                    entry="rcppml_gpu_nmf_dense_unified_" + ("float" if precision == "fp32" else "double"))
        if loss != "mse":
            misc["theta"] = res["theta"]
        return NMFModel(w=W_T.copy(), d=res["d"], h=H.T.copy(), misc=misc)
    target_args = {}
    if target_H is not None:
        tl = (0.0, float(target_lambda)) if np.ndim(target_lambda) == 0 else tuple(float(v) for v in target_lambda)   # R/nmf_thin.R:646-648
        TH = np.asarray(target_H, np.float64)
        if TH.shape == (k, n):
            TH = TH.T
        if TH.shape != (n, k):
            raise ValueError("target_H must be k x n")
        if tl[1] != 0:
            target_args["target_H"] = (np.ascontiguousarray(TH), tl[1])
    res = _abi.nmf_unified(A.p, A.i, A.x, m, n, k, W_T, H, entry="ex", max_iter=int(maxit), tol=float(tol), L1_H=L1h,
                           L1_W=L1w, L2_H=L2h, L2_W=L2w, L21_H=L21h, L21_W=L21w, ortho_H=angh, ortho_W=angw, ub_H=ubh, ub_W=ubw, cd_maxit=int(cd_maxit), verbose=int(verbose),
                           seed=seed_int & 0x7FFFFFFF, patience=int(patience), nonneg_W=int(nnw), nonneg_H=int(nnh),
                           norm_type=norm_type, solver_mode=0 if solver == "cd" else 1, mask=mask_arg, cd_tol=float(cd_tol),
                           loss_type={"mse": 0, "gp": 4, "nb": 5, "gamma": 6, "inverse_gaussian": 7, "tweedie": 8}[loss], projective=int(bool(projective)), symmetric=int(bool(symmetric)),
                           tweedie_power=float(tweedie_power), robust_delta=robust_delta, irls_max_iter=int(irls_max_iter), irls_tol=float(irls_tol),
                           gp_dispersion_mode={"none": 0, "global": 1, "per_row": 2, "per_col": 3}[dispersion],
                           nb_size=(nb_size_init, nb_size_max, nb_size_min), gp_theta=(float(theta_init), float(theta_max), float(theta_min)),
                           sort_model=int(sort_model), precision=_abi.F32 if precision == "fp32" else _abi.F64,
                           want_history=True, **graph_args, **target_args)
    if res["status"] != 0:
        raise _abi.BackendError("GPU NMF failed: %s" % res.get("error"))
    misc = dict(tol=res["tol"], iter=res["iter"], loss=res["loss"], loss_history=res.get("loss_history"),
                converged=res["converged"], solver=solver, solver_mode=0 if solver == "cd" else 1, L1=(L1w, L1h),
                L2=(L2w, L2h), seed=seed_int, precision=precision, resource="gpu", loss_type=loss,
                entry="rcppml_gpu_nmf_target" if target_args else "rcppml_gpu_nmf_ex")
    if loss != "mse":
        misc["theta"] = res["theta"]                                   # R: misc$theta (RcppFunctions_nmf.cpp:156-158)
    return NMFModel(w=W_T.copy(), d=res["d"], h=H.T.copy(), misc=misc)


def nnls(w=None, h=None, A=None, L1=0.0, L2=0.0, cd_maxit=100, cd_tol=1e-8, upper_bound=0.0, nonneg=True, warm_start=None):
    """R/solve.R:84-357 (MSE path, fp64): given w (m x k) solve for h (k x n), or given h (k x n) solve for w (m x k)."""
    if A is None or (w is None) == (h is None):
        raise ValueError("provide A and exactly one of w, h")
    Ac = _as_csc(A)
    if w is not None:
        w = np.asarray(w, dtype=np.float64)
        if w.shape[0] != Ac.rows:
            raise ValueError("dimensions of 'w' and 'A' are incompatible")
        k = w.shape[1]
        out = np.zeros((Ac.cols, k)) if warm_start is None else np.ascontiguousarray(np.asarray(warm_start, np.float64).T)
        _abi.nnls_double(Ac.p, Ac.i, Ac.x, Ac.rows, Ac.cols, k, np.ascontiguousarray(w), out, cd_maxit=cd_maxit, cd_tol=cd_tol,
                         L1=L1, L2=L2, ub=upper_bound, nonneg=int(nonneg), warm=int(warm_start is not None))
        return out.T.copy()
    h = np.asarray(h, dtype=np.float64)                                 # k x n  -> solve on A^T (R/solve.R:326-355)
    if h.shape[1] != Ac.cols:
        raise ValueError("dimensions of 'h' and 'A' are incompatible")
    At = Ac.transpose()
    k = h.shape[0]
    out = np.zeros((At.cols, k)) if warm_start is None else np.ascontiguousarray(np.asarray(warm_start, np.float64))
    _abi.nnls_double(At.p, At.i, At.x, At.rows, At.cols, k, np.ascontiguousarray(h.T), out, cd_maxit=cd_maxit, cd_tol=cd_tol,
                     L1=L1, L2=L2, ub=upper_bound, nonneg=int(nonneg), warm=int(warm_start is not None))
    return out


def predict(model, data, L1=None, L2=None, upper_bound=0.0):
    """R/predict_nmf.R:48-97 -> Rcpp_predict (src/RcppFunctions_utils.cpp:23-52): project new samples onto model.w
    with cd_maxit = 100, cd_tol = 1e-8, nonneg = TRUE.  Returns h (k x n).  L1 / L2 default to the h-side penalties the model was
    fitted with (misc$L1[2], misc$L2[2]; R/predict_nmf.R:52-53) and are validated with the reference's messages (:56-59)."""
    if L1 is None:
        L1 = model.misc.get("L1", (0.0, 0.0))[1] if isinstance(model.misc, dict) else 0.0
    if L2 is None:
        L2 = model.misc.get("L2", (0.0, 0.0))[1] if isinstance(model.misc, dict) else 0.0
    if np.ndim(L1) != 0:
        raise ValueError("'L1' must be a single value giving the penalty on 'h'")
    if L1 >= 1 or L1 < 0:
        raise ValueError("L1 penalty must be strictly in the range [0,1)")
    if np.ndim(L2) != 0:
        raise ValueError("'L2' must be a single value giving the penalty on 'h'")
    if L2 < 0:
        raise ValueError("L2 penalty must be strictly >= 0")
    return nnls(w=model.w, A=data, L1=float(L1), L2=float(L2), cd_maxit=100, cd_tol=1e-8, upper_bound=upper_bound, nonneg=True)


def evaluate(model, data, mask=None, missing_only=False):
    """R/nmf_methods.R:356-469 -> Rcpp_evaluate_loss / Rcpp_evaluate_loss_missing (src/RcppFunctions_utils.cpp:95-213): MEAN squared
    error of w diag(d) h over all entries, over the nonzeros of `data` if mask == 'zeros', or -- mask = <matrix>, missing_only = TRUE --
    over the entries the mask marks (values > 0), zeros of `data` included.  As in the reference, a mask MATRIX without missing_only
    changes nothing: Rcpp_evaluate_loss receives it and never reads it (utils.cpp:152-163 -> compute_loss_general, :95-148).
    (model.misc['loss'] is the SUM)"""
    if missing_only and mask is None:
        raise ValueError("a mask matrix must be specified to set 'missing_only = TRUE'")          # R/nmf_methods.R:366
    A = _as_csc(data)
    k = model.d.shape[0]
    if mask is not None and not isinstance(mask, str):
        M = _as_csc(mask)
        if M.shape != A.shape:
            raise ValueError("mask dimensions must match data")
        if not missing_only:
            mask = None                                                # the reference's quirk: the mask matrix is ignored here
        else:
            # the marked entries as a CSC that STORES them (zeros of `data` included): the device pass over stored entries is then
            # exactly Rcpp_evaluate_loss_missing's loop (utils.cpp:185-206)
            keep = M.x > 0
            cols = np.repeat(np.arange(M.cols, dtype=np.int64), np.diff(M.p))[keep]
            rows = M.i[keep].astype(np.int64)
            if rows.shape[0] == 0:
                return 0.0
            vals = np.asarray(A.to_scipy().tocsr()[rows, cols]).ravel().astype(np.float64)
            counts = np.bincount(cols, minlength=M.cols)
            p = np.zeros(M.cols + 1, np.int64)
            np.cumsum(counts, out=p[1:])
            E = CSC(A.shape, p.astype(np.int32), rows.astype(np.int32), vals)
            return _abi.evaluate_mse_double(E.p, E.i, E.x, E.rows, E.cols, k, np.ascontiguousarray(model.w), model.d,
                                            np.ascontiguousarray(model.h.T), mask_zeros=True)
    if mask not in (None, "zeros"):
        raise ValueError("'mask' must be NULL, 'zeros', or a matrix")
    return _abi.evaluate_mse_double(A.p, A.i, A.x, A.rows, A.cols, k, np.ascontiguousarray(model.w),
                                    model.d, np.ascontiguousarray(model.h.T), mask_zeros=(mask == "zeros"))


def mse(w, d=None, h=None, data=None, mask=None, missing_only=False):
    """R/nmf_methods.R:488-492: evaluate() of the model (w, d, h) given as separate arrays; d defaults to ones."""
    if h is None or data is None:
        raise ValueError("'h' and 'data' are required")
    h = np.asarray(h, np.float64)
    d = np.ones(h.shape[0]) if d is None else np.asarray(d, np.float64)
    return evaluate(NMFModel(w=np.asarray(w, np.float64), d=d, h=h, misc={}), data, mask=mask, missing_only=missing_only)
