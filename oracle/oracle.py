"""ctypes binding of oracle/liboracle.so -- the CPU restatement of RcppML's ALS-NNLS NMF path.

TEST INFRASTRUCTURE ONLY.  Importable from tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke(); the product package (rcppml_amd/) must never import this module.
See oracle/nmf_oracle.hpp for the parity status ("parity unpinned" except the reference's
known answers) and the reference file:line each routine restates.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RCPPML_ORACLE_DIR: load the shared objects from another directory (oracle/_san: the ASan + UBSan builds of
# `make -C oracle sanitize`, run by tools/sanitize/run.sh)
_LIBDIR = os.environ.get("RCPPML_ORACLE_DIR") or _HERE
_LIBS = {}

i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(native=False, quiet=True):
    """Compile the oracle (g++, a few seconds).  native=True builds the -O3 -march=native timing build."""
    target = "native" if native else "all"
    subprocess.run(["make", "-C", _HERE, target], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def lib(native=False):
    name = "liboracle_native.so" if native else "liboracle.so"
    if name not in _LIBS:
        path = os.path.join(_LIBDIR, name)
        if not os.path.exists(path):
            if _LIBDIR != _HERE:
                raise FileNotFoundError(path)
            build(native)
        _LIBS[name] = C.CDLL(path)
    return _LIBS[name]


def _suf(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32", C.c_float
    if dtype == np.float64:
        return "f64", C.c_double
    raise TypeError(dtype)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


class Csc:
    """CSC matrix view: shape (rows, cols), int32 p/i, values x (float64 storage, cast on demand)."""

    def __init__(self, shape, p, i, x):
        self.rows, self.cols = int(shape[0]), int(shape[1])
        self.p = np.ascontiguousarray(p, dtype=np.int32)
        self.i = np.ascontiguousarray(i, dtype=np.int32)
        self.x = np.ascontiguousarray(x, dtype=np.float64)
        assert self.p.shape[0] == self.cols + 1 and self.p[-1] == self.i.shape[0] == self.x.shape[0]

    @property
    def nnz(self):
        return int(self.x.shape[0])

    def values(self, dtype):
        return self.x if np.dtype(dtype) == np.float64 else self.x.astype(dtype)

    def transpose(self):
        L = lib()
        tp = np.empty(self.rows + 1, np.int32)
        ti = np.empty(max(self.nnz, 1), np.int32)
        tx = np.empty(max(self.nnz, 1), np.float64)
        L.oracle_transpose_csc_f64(C.c_int(self.rows), C.c_int(self.cols), _p(self.p), _p(self.i), _p(self.x),
                                   _p(tp), _p(ti), _p(tx))
        return Csc((self.cols, self.rows), tp, ti[:self.nnz], tx[:self.nnz])

    def toarray(self):
        out = np.zeros((self.rows, self.cols))
        for j in range(self.cols):
            s, e = self.p[j], self.p[j + 1]
            out[self.i[s:e], j] = self.x[s:e]
        return out

    @staticmethod
    def from_dense(a):
        a = np.asarray(a, dtype=np.float64)
        rows, cols = a.shape
        p = [0]
        ii, xx = [], []
        for j in range(cols):
            nz = np.nonzero(a[:, j])[0]
            ii.append(nz.astype(np.int32))
            xx.append(a[nz, j])
            p.append(p[-1] + len(nz))
        return Csc((rows, cols), np.asarray(p, np.int32),
                   np.concatenate(ii) if ii else np.zeros(0, np.int32),
                   np.concatenate(xx) if xx else np.zeros(0))


# ----------------------------------------------------------------------------- RNG / init
def splitmix_stream(seed, count):
    L = lib()
    L.oracle_splitmix_init_state.restype = C.c_uint64
    L.oracle_splitmix_next.restype = C.c_uint64
    st = C.c_uint64(L.oracle_splitmix_init_state(C.c_uint64(seed)))
    return [int(L.oracle_splitmix_next(C.byref(st))) for _ in range(count)]


def splitmix_state(seed):
    L = lib()
    L.oracle_splitmix_init_state.restype = C.c_uint64
    return int(L.oracle_splitmix_init_state(C.c_uint64(seed)))


def init_factors(seed, k, m, n, dtype=np.float64):
    """reference nmf/nmf_init.hpp:166-182: one SplitMix64 stream fills W_T (k x m) then H (k x n)."""
    suf, _ = _suf(dtype)
    W_T = np.empty((m, k), dtype)   # memory = column-major k x m
    H = np.empty((n, k), dtype)
    getattr(lib(), "oracle_init_factors_" + suf)(C.c_uint32(seed), C.c_int(k), C.c_int(m), C.c_int(n), _p(W_T), _p(H))
    return W_T, H


def fill_uniform(seed, rows, cols, dtype=np.float64):
    suf, _ = _suf(dtype)
    out = np.empty((cols, rows), dtype)
    getattr(lib(), "oracle_fill_uniform_" + suf)(C.c_uint64(seed), _p(out), C.c_int(rows), C.c_int(cols))
    return out


# ----------------------------------------------------------------------------- primitives
# Dense matrices are passed as numpy arrays of shape (cols, k): C-contiguous memory == column-major k x cols.
def gram(F, dtype=None):
    dtype = dtype or F.dtype
    suf, _ = _suf(dtype)
    F = _f(F, dtype)
    r, k = F.shape
    G = np.empty((k, k), dtype)
    getattr(lib(), "oracle_gram_" + suf)(_p(F), C.c_int(k), C.c_int(r), _p(G))
    return G


def rhs(A, F, dtype=None, threads=1):
    dtype = dtype or F.dtype
    suf, _ = _suf(dtype)
    F = _f(F, dtype)
    k = F.shape[1]
    B = np.empty((A.cols, k), dtype)
    x = A.values(dtype)
    getattr(lib(), "oracle_rhs_" + suf)(C.c_int(A.rows), C.c_int(A.cols), _p(A.p), _p(A.i), _p(x), _p(F), C.c_int(k),
                                        _p(B), C.c_int(threads))
    return B


def cd_col(G, b, x, L1=0.0, L2=0.0, nonneg=True, maxit=100, ub=0.0, tol=0.0):
    dtype = G.dtype
    suf, ct = _suf(dtype)
    G = _f(G, dtype)
    b = _f(b, dtype).copy()
    x = _f(x, dtype).copy()
    fn = getattr(lib(), "oracle_cd_col_" + suf)
    fn.restype = C.c_int
    it = fn(_p(G), _p(b), _p(x), C.c_int(G.shape[0]), ct(L1), ct(L2), C.c_int(int(nonneg)), C.c_int(maxit), ct(ub), ct(tol))
    return x, b, it


def nnls_batch(G, B, X=None, maxit=100, tol=1e-8, L1=0.0, L2=0.0, nonneg=True, threads=1, ub=0.0, warm=False):
    dtype = G.dtype
    suf, ct = _suf(dtype)
    G = _f(G, dtype)
    B = _f(B, dtype).copy()
    n, k = B.shape
    X = np.zeros((n, k), dtype) if X is None else _f(X, dtype).copy()
    getattr(lib(), "oracle_nnls_batch_" + suf)(_p(G), _p(B), _p(X), C.c_int(k), C.c_int(n), C.c_int(maxit), ct(tol),
                                                ct(L1), ct(L2), C.c_int(int(nonneg)), C.c_int(threads), ct(ub),
                                                C.c_int(int(warm)))
    return X


def fused_cd(A, F, G, X, maxit=100, tol=1e-8, L1=0.0, nonneg=True, threads=1, warm=False, ub=0.0, native=False):
    dtype = G.dtype
    suf, ct = _suf(dtype)
    F, G = _f(F, dtype), _f(G, dtype)
    X = _f(X, dtype).copy()
    k = F.shape[1]
    x = A.values(dtype)
    getattr(lib(native), "oracle_fused_cd_" + suf)(C.c_int(A.rows), C.c_int(A.cols), _p(A.p), _p(A.i), _p(x), _p(F), _p(G),
                                                    _p(X), C.c_int(k), C.c_int(maxit), ct(tol), ct(L1),
                                                    C.c_int(int(nonneg)), C.c_int(threads), C.c_int(int(warm)), ct(ub))
    return X


def fused_chol(A, F, G, L1=0.0, nonneg=True, threads=1, ub=0.0):
    dtype = G.dtype
    suf, ct = _suf(dtype)
    F, G = _f(F, dtype), _f(G, dtype)
    k = F.shape[1]
    X = np.zeros((A.cols, k), dtype)
    x = A.values(dtype)
    getattr(lib(), "oracle_fused_chol_" + suf)(C.c_int(A.rows), C.c_int(A.cols), _p(A.p), _p(A.i), _p(x), _p(F), _p(G), _p(X),
                                               C.c_int(k), ct(L1), C.c_int(int(nonneg)), C.c_int(threads), ct(ub))
    return X


def chol_clip_batch(G, B, nonneg=True, threads=1):
    dtype = G.dtype
    suf, _ = _suf(dtype)
    G, B = _f(G, dtype), _f(B, dtype)
    n, k = B.shape
    X = np.zeros((n, k), dtype)
    getattr(lib(), "oracle_chol_clip_batch_" + suf)(_p(G), _p(B), _p(X), C.c_int(k), C.c_int(n), C.c_int(int(nonneg)),
                                                     C.c_int(threads))
    return X


def llt(G):
    dtype = G.dtype
    suf, _ = _suf(dtype)
    G = _f(G, dtype)
    k = G.shape[0]
    L = np.zeros((k, k), dtype)
    getattr(lib(), "oracle_llt_" + suf)(_p(G), C.c_int(k), _p(L))
    return L.T.copy()   # stored column-major -> return as row-major lower-triangular


def extract_scaling(X, norm_type=0):
    dtype = X.dtype
    suf, _ = _suf(dtype)
    X = _f(X, dtype).copy()
    c, k = X.shape
    d = np.empty(k, dtype)
    getattr(lib(), "oracle_extract_scaling_" + suf)(_p(X), C.c_int(k), C.c_int(c), _p(d), C.c_int(norm_type))
    return X, d


def trace_AtA(A, dtype=np.float64):
    suf, ct = _suf(dtype)
    fn = getattr(lib(), "oracle_trace_AtA_" + suf)
    fn.restype = ct
    x = A.values(dtype)
    return float(fn(C.c_int(A.rows), C.c_int(A.cols), _p(A.p), _p(A.i), _p(x)))


def loss_cross(At, W_T, H, d, threads=1):
    dtype = W_T.dtype
    suf, ct = _suf(dtype)
    fn = getattr(lib(), "oracle_loss_cross_" + suf)
    fn.restype = ct
    x = At.values(dtype)
    W_T, H, d = _f(W_T, dtype), _f(H, dtype), _f(d, dtype)
    return float(fn(C.c_int(At.rows), C.c_int(At.cols), _p(At.p), _p(At.i), _p(x), _p(W_T), _p(H), _p(d),
                    C.c_int(W_T.shape[1]), C.c_int(threads)))


# ----------------------------------------------------------------------------- full fit
class FitResult:
    pass


def _graph_args(g, dtype, ct):
    if g is None:
        return [None, None, None, ct(0)]
    L, lam = g
    x = L.values(dtype)
    _graph_args.keep = (getattr(_graph_args, 'keep', []) + [x])[-4:]   # keep the converted values alive for the call (2 per fit)
    return [_p(L.p), _p(L.i), _p(x), ct(lam)]


def dense_as_csc(M):
    """A dense m x n array as a CSC that stores EVERY entry (zeros included): with it nmf_fit(unfused=True) restates the
    reference's dense-input fit (dense products = sums over all rows, in row order)."""
    M = np.asarray(M, np.float64)
    m, n = M.shape
    return Csc((m, n), np.arange(0, m * n + 1, m, dtype=np.int32), np.tile(np.arange(m, dtype=np.int32), n), M.T.reshape(-1))


_KEEP_TARGETS = []


def _target_args(t, dtype, ct):
    if t is None:
        return None, ct(0)
    M = _f(t[0], dtype)
    _KEEP_TARGETS.append(M)
    del _KEEP_TARGETS[:-8]
    return _p(M), ct(t[1])


def proj_adv(G, TG, abs_lambda):
    """PROJ_ADV Gram modification (variant_helpers.hpp:112-146), fp64: returns the modified copy of G."""
    G = _f(G, np.float64).copy()
    TG = _f(TG, np.float64)
    lib().oracle_proj_adv_f64(_p(G), _p(TG), C.c_int(G.shape[0]), C.c_double(abs_lambda))
    return G


def nmf_fit(A, W_T, H, dtype=np.float64, max_iter=100, tol=1e-4, L1=(0.0, 0.0), L2=(0.0, 0.0), ub=(0.0, 0.0),
            cd_maxit=100, cd_tol=1e-8, patience=5, nonneg=(True, True), norm_type=0, solver_mode=0, loss_type=0,
            irls_max_iter=5, irls_tol=1e-4, dispersion_mode=2, nb_size=(10.0, 1e6, 0.01), sort_model=True, threads=1,
            mask=None, native=False, tweedie_power=1.5, L21=(0.0, 0.0), angular=(0.0, 0.0), robust_delta=0.0, projective=False, graph_H=None, graph_W=None,
            gp_theta=(0.1, 5.0), gamma_phi=(1.0, 1e4, 1e-6), symmetric=False, unfused=False, target_H=None, target_W=None, dense_input=False):
    """CPU restatement of nmf_fit<CPU> (reference nmf/fit_cpu.hpp).  L1/L2/ub/nonneg are (W, H) pairs as in R
    (src/RcppFunctions_nmf.cpp:59-62).  W_T: (m, k) array = column-major k x m; H: (n, k).
    graph_H / graph_W: (Csc Laplacian, lambda) over the columns of H / W_T (features/graph_reg.hpp).
    target_H / target_W: (matrix (n, k) / (m, k), lambda): target regularisation (nmf/variant_helpers.hpp:107-146)."""
    suf, ct = _suf(dtype)
    W_T = _f(W_T, dtype).copy()
    H = _f(H, dtype).copy()
    m, k = W_T.shape
    n = H.shape[0]
    assert (A.rows, A.cols) == (m, n)
    d = np.ones(k, dtype)
    x = A.values(dtype)
    hist = np.full(max(max_iter, 1), np.nan, dtype)
    theta = np.zeros(n if dispersion_mode == 3 else m, dtype)          # DispersionMode::PER_COL: one value per column
    it, conv = C.c_int(0), C.c_int(0)
    loss, ftol = ct(0), ct(0)
    if mask is not None:
        mx = mask.values(dtype)
        mp, mi, mxp = _p(mask.p), _p(mask.i), _p(mx)
    else:
        mp = mi = mxp = None
    getattr(lib(native), "oracle_nmf_fit_" + suf)(
        C.c_int(m), C.c_int(n), _p(A.p), _p(A.i), _p(x), C.c_int(k), _p(W_T), _p(H), _p(d), C.c_int(max_iter), ct(tol),
        ct(L1[1]), ct(L1[0]), ct(L2[1]), ct(L2[0]), ct(ub[1]), ct(ub[0]), C.c_int(cd_maxit), ct(cd_tol), C.c_int(patience),
        C.c_int(int(nonneg[0])), C.c_int(int(nonneg[1])), C.c_int(norm_type), C.c_int(solver_mode), C.c_int(loss_type),
        C.c_int(irls_max_iter), ct(irls_tol), C.c_int(dispersion_mode), ct(nb_size[0]), ct(nb_size[1]), ct(nb_size[2]),
        C.c_int(int(sort_model)), C.c_int(threads), mp, mi, mxp, C.byref(it), C.byref(conv), C.byref(loss), C.byref(ftol),
        _p(hist), _p(theta), ct(tweedie_power), ct(L21[1]), ct(L21[0]), ct(angular[1]), ct(angular[0]), ct(robust_delta), C.c_int(int(projective)),
        *_graph_args(graph_H, dtype, ct), *_graph_args(graph_W, dtype, ct),
        ct(gp_theta[0]), ct(gp_theta[1]), ct(gamma_phi[0]), ct(gamma_phi[1]), ct(gamma_phi[2]), C.c_int(int(symmetric)), C.c_int(2 if dense_input else int(unfused)),
        *_target_args(target_H, dtype, ct), *_target_args(target_W, dtype, ct))
    r = FitResult()
    r.W_T, r.H, r.d = W_T, H, d
    r.iter, r.converged, r.loss, r.tol = it.value, bool(conv.value), float(loss.value), float(ftol.value)
    r.loss_history = hist[:it.value].copy()
    r.theta = theta
    return r


# ----------------------------------------------------------------------------- fp64 R-surface helpers
def c_nnls(w_T, A, h0=None, cd_maxit=100, cd_tol=1e-8, L1=0.0, L2=0.0, ub=0.0, nonneg=True, threads=1):
    """reference src/RcppFunctions_utils.cpp:313-366.  w_T: (m, k); returns h (n, k)."""
    w_T = _f(w_T, np.float64)
    m, k = w_T.shape
    warm = h0 is not None
    h = _f(h0, np.float64).copy() if warm else np.zeros((A.cols, k))
    lib().oracle_c_nnls(_p(w_T), C.c_int(k), C.c_int(m), C.c_int(A.cols), _p(A.p), _p(A.i), _p(A.x), _p(h),
                        C.c_int(cd_maxit), C.c_double(cd_tol), C.c_double(L1), C.c_double(L2), C.c_double(ub),
                        C.c_int(int(nonneg)), C.c_int(threads), C.c_int(int(warm)))
    return h


def evaluate_mse(W, d, H, A, mask_zeros=False):
    """reference src/RcppFunctions_utils.cpp:95-163 (mean).  W: (m, k) ROW-major m x k as R holds it -> pass W_T (m,k)."""
    fn = lib().oracle_evaluate_mse
    fn.restype = C.c_double
    W_T = _f(W, np.float64)
    m, k = W_T.shape
    Wcm = np.ascontiguousarray(W_T.T)          # (k, m) C-contiguous == column-major m x k
    H = _f(H, np.float64)
    d = _f(d, np.float64)
    return float(fn(_p(Wcm), _p(d), _p(H), C.c_int(k), C.c_int(m), C.c_int(A.cols), _p(A.p), _p(A.i), _p(A.x),
                    C.c_int(int(mask_zeros))))


def num_threads():
    return int(lib().oracle_num_threads())


# ----------------------------------------------------------------------------- NB-IRLS primitives
def irls(loss_type, A, F, G, k, L1=0.0, L2=0.0, nonneg=True, cd_maxit=100, irls_max_iter=5, irls_tol=1e-4, threads=1,
         theta_row=None, theta_col=None, dtype=np.float64, power=1.5, robust=0.0, dense_input=False):
    """Generic IRLS half-update (dense_input: A stores every entry and the column solve is the reference's dense one): loss_type 5 = NB, 4 = GP (KL weights, fit_cpu.hpp:568-574), 6 = Gamma, 7 = inverse
    Gaussian, 8 = Tweedie(power)."""
    suf, ct = _suf(dtype)
    F, G = _f(F, dtype), _f(G, dtype)
    X = np.zeros((A.cols, k), dtype)
    x = A.values(dtype)
    tr = _f(theta_row, dtype) if theta_row is not None else None
    tc = _f(theta_col, dtype) if theta_col is not None else None
    getattr(lib(), "oracle_irls_" + suf)(C.c_int(loss_type + (16 if dense_input else 0)), C.c_int(A.rows), C.c_int(A.cols), _p(A.p), _p(A.i), _p(x), _p(F),
                                         _p(G), _p(X), C.c_int(k), ct(L1), ct(L2), C.c_int(int(nonneg)), C.c_int(cd_maxit),
                                         C.c_int(irls_max_iter), ct(irls_tol), C.c_int(threads),
                                         _p(tr) if tr is not None else None, _p(tc) if tc is not None else None, ct(power), ct(robust))
    return X


def irls_loss(loss_type, A, W_T, d, H, theta_row, dtype=np.float64, power=1.5, robust=0.0):
    suf, ct = _suf(dtype)
    fn = getattr(lib(), "oracle_irls_loss_" + suf)
    fn.restype = ct
    W_T, H, d, th = _f(W_T, dtype), _f(H, dtype), _f(d, dtype), _f(theta_row, dtype)
    x = A.values(dtype)
    return float(fn(C.c_int(loss_type), C.c_int(A.rows), C.c_int(A.cols), _p(A.p), _p(A.i), _p(x), _p(W_T), _p(d), _p(H),
                    C.c_int(W_T.shape[1]), _p(th), ct(power), ct(robust)))


def irls_nb(A, F, G, k, L1=0.0, L2=0.0, nonneg=True, cd_maxit=100, irls_max_iter=5, irls_tol=1e-4, threads=1,
            theta_row=None, theta_col=None, dtype=np.float64):
    suf, ct = _suf(dtype)
    F, G = _f(F, dtype), _f(G, dtype)
    X = np.zeros((A.cols, k), dtype)
    x = A.values(dtype)
    tr = _f(theta_row, dtype) if theta_row is not None else None
    tc = _f(theta_col, dtype) if theta_col is not None else None
    getattr(lib(), "oracle_irls_nb_" + suf)(C.c_int(A.rows), C.c_int(A.cols), _p(A.p), _p(A.i), _p(x), _p(F), _p(G), _p(X),
                                           C.c_int(k), ct(L1), ct(L2), C.c_int(int(nonneg)), C.c_int(cd_maxit),
                                           C.c_int(irls_max_iter), ct(irls_tol), C.c_int(threads),
                                           _p(tr) if tr is not None else None, _p(tc) if tc is not None else None)
    return X


def nb_size_update(A, W_T, H, d, nb_size, dispersion_mode=2, r_min=0.01, r_max=1e6, dtype=np.float64, dense_input=False):
    suf, ct = _suf(dtype)
    W_T, H, d = _f(W_T, dtype), _f(H, dtype), _f(d, dtype)
    out = _f(nb_size, dtype).copy()
    x = A.values(dtype)
    getattr(lib(), "oracle_nb_size_update_" + suf)(C.c_int(A.rows), C.c_int(A.cols), _p(A.p), _p(A.i), _p(x), _p(W_T), _p(H),
                                                  _p(d), C.c_int(W_T.shape[1]), C.c_int(dispersion_mode + (16 if dense_input else 0)), ct(r_min),
                                                  ct(r_max), _p(out))
    return out


def dispersion_update(loss_type, A, W_T, H, d, theta, dispersion_mode=2, power=1.5, lo=1e-6, hi=1e4, dtype=np.float64, dense_input=False):
    """GP theta (loss_type 4, fit_cpu.hpp:914-1008, hi = cap) or Gamma / IG / Tweedie phi (6 / 7 / 8, :1561-1670)."""
    suf, ct = _suf(dtype)
    th = _f(theta, dtype).copy()
    x = A.values(dtype)
    k = W_T.shape[1]
    getattr(lib(), "oracle_dispersion_update_" + suf)(C.c_int(loss_type), C.c_int(A.rows), C.c_int(A.cols), _p(A.p), _p(A.i), _p(x),
                                                     _p(_f(W_T, dtype)), _p(_f(H, dtype)), _p(_f(d, dtype)), C.c_int(k),
                                                     C.c_int(dispersion_mode + (16 if dense_input else 0)), ct(power), ct(lo), ct(hi), _p(th))
    return th


def nb_loss(A, W_T, d, H, theta_row, dtype=np.float64):
    suf, ct = _suf(dtype)
    fn = getattr(lib(), "oracle_nb_loss_" + suf)
    fn.restype = ct
    W_T, H, d, th = _f(W_T, dtype), _f(H, dtype), _f(d, dtype), _f(theta_row, dtype)
    x = A.values(dtype)
    return float(fn(C.c_int(A.rows), C.c_int(A.cols), _p(A.p), _p(A.i), _p(x), _p(W_T), _p(d), _p(H), C.c_int(W_T.shape[1]),
                    _p(th)))


# --------------------------------------------------------------------------------------------
# Cross-validation path (nmf/fit_cv.hpp): speckled holdout mask, per-column Gram correction
# --------------------------------------------------------------------------------------------
def cv_hash(seed, i, j):
    fn = lib().oracle_cv_hash
    fn.restype = C.c_uint64
    return int(fn(C.c_uint64(seed), C.c_uint32(i), C.c_uint32(j)))


def cv_half_update(A, F, G, X, k, frac, cv_seed, mask_zeros=False, transposed=False, L1=0.0, nonneg=True, cd_maxit=100,
                   solver_mode=0, threads=1, dtype=np.float64):
    """One CV half-update over the columns of A (the W side passes A^T with transposed=True)."""
    suf, ct = _suf(dtype)
    F, G = _f(F, dtype), _f(G, dtype)
    X = _f(X, dtype).copy()
    x = A.values(dtype)
    getattr(lib(), "oracle_cv_half_update_" + suf)(C.c_int(A.rows), C.c_int(A.cols), _p(A.p), _p(A.i), _p(x), _p(F), _p(G), _p(X),
                                                   C.c_int(k), C.c_double(frac), C.c_uint64(cv_seed), C.c_int(int(mask_zeros)),
                                                   C.c_int(int(transposed)), ct(L1), C.c_int(int(nonneg)), C.c_int(cd_maxit),
                                                   C.c_int(solver_mode), C.c_int(threads))
    return X


def cv_test_error(A, W_T, d, H, frac, cv_seed, mask_zeros=False, dtype=np.float64):
    suf, ct = _suf(dtype)
    W_T, H, d = _f(W_T, dtype), _f(H, dtype), _f(d, dtype)
    x = A.values(dtype)
    sq, cnt = ct(0), C.c_int64(0)
    getattr(lib(), "oracle_cv_test_error_" + suf)(C.c_int(A.rows), C.c_int(A.cols), _p(A.p), _p(A.i), _p(x), _p(W_T), _p(d), _p(H),
                                                  C.c_int(W_T.shape[1]), C.c_double(frac), C.c_uint64(cv_seed),
                                                  C.c_int(int(mask_zeros)), C.byref(sq), C.byref(cnt))
    return float(sq.value), int(cnt.value)


def cv_irls_half_update(A, F, X, k, frac, cv_seed, loss_type, G_add=None, mask_zeros=False, transposed=False, L1=0.0, nonneg=True,
                        cd_maxit=100, solver_mode=0, irls_max_iter=5, irls_tol=1e-4, power=1.5, robust=0.0, threads=1, dtype=np.float64):
    """One CV half-update with an IRLS loss (reference nmf/cv_detail.hpp:101-292) over the columns of A (W side: A^T, transposed)."""
    suf, ct = _suf(dtype)
    F = _f(F, dtype)
    X = _f(X, dtype).copy()
    Ga = _f(G_add, dtype) if G_add is not None else None
    x = A.values(dtype)
    getattr(lib(), "oracle_cv_irls_half_update_" + suf)(
        C.c_int(A.rows), C.c_int(A.cols), _p(A.p), _p(A.i), _p(x), _p(F), _p(Ga) if Ga is not None else None, _p(X), C.c_int(k),
        C.c_double(frac), C.c_uint64(cv_seed), C.c_int(int(mask_zeros)), C.c_int(int(transposed)), ct(L1), C.c_int(int(nonneg)),
        C.c_int(cd_maxit), C.c_int(solver_mode), C.c_int(loss_type), C.c_int(irls_max_iter), ct(irls_tol), ct(power), ct(robust),
        C.c_int(threads))
    return X


def cv_explicit_loss(A, W_T, d, H, frac, cv_seed, loss_type, theta=None, mask_zeros=False, power=1.5, dtype=np.float64):
    """(train sum, n_train, test sum, n_test) of the per-element losses (reference nmf/fit_cv.hpp:1377-1443)."""
    suf, ct = _suf(dtype)
    W_T, H, d = _f(W_T, dtype), _f(H, dtype), _f(d, dtype)
    th = _f(theta if theta is not None else np.zeros(A.rows), dtype)
    x = A.values(dtype)
    tr, te, ntr, nte = ct(0), ct(0), C.c_int64(0), C.c_int64(0)
    getattr(lib(), "oracle_cv_explicit_loss_" + suf)(
        C.c_int(A.rows), C.c_int(A.cols), _p(A.p), _p(A.i), _p(x), _p(W_T), _p(d), _p(H), C.c_int(W_T.shape[1]), C.c_double(frac),
        C.c_uint64(cv_seed), C.c_int(int(mask_zeros)), C.c_int(loss_type), ct(power), _p(th), C.byref(tr), C.byref(ntr), C.byref(te),
        C.byref(nte))
    return float(tr.value), int(ntr.value), float(te.value), int(nte.value)


def cv_gp_theta_update(A, W_T, d, H, theta, frac, cv_seed, mode=2, theta_max=5.0, dtype=np.float64):
    """GP theta (MM update, five inner passes) over the TRAINING entries (reference nmf/fit_cv.hpp:866-961); frac = 0: all entries."""
    suf, ct = _suf(dtype)
    W_T, H, d = _f(W_T, dtype), _f(H, dtype), _f(d, dtype)
    th = _f(theta, dtype).copy()
    x = A.values(dtype)
    getattr(lib(), "oracle_cv_gp_theta_update_" + suf)(
        C.c_int(A.rows), C.c_int(A.cols), _p(A.p), _p(A.i), _p(x), _p(W_T), _p(d), _p(H), C.c_int(W_T.shape[1]), C.c_double(frac),
        C.c_uint64(cv_seed), C.c_int(mode), ct(theta_max), _p(th))
    return th


def irls_weight_gp(observed, predicted, theta, blend=1.0, dtype=np.float64):
    suf, ct = _suf(dtype)
    fn = getattr(lib(), "oracle_irls_weight_gp_" + suf)
    fn.restype = ct
    return fn(ct(observed), ct(predicted), ct(theta), ct(blend))


class CvFitResult:
    pass


def nmf_fit_cv(A, W_T, H, dtype=np.float64, max_iter=100, tol=1e-4, L1=(0.0, 0.0), L2=(0.0, 0.0), cd_maxit=100,
               nonneg=(True, True), norm_type=0, solver_mode=0, holdout_fraction=0.1, cv_seed=0, mask_zeros=False,
               cv_patience=5, threads=1, native=False, graph_H=None, graph_W=None, loss_type=0, irls_max_iter=5, irls_tol=1e-4,
               dispersion_mode=2, gp_theta=(0.1, 5.0), tweedie_power=1.5, robust_delta=0.0, mask=None):
    """CPU restatement of nmf_fit_cv (reference nmf/fit_cv.hpp), sparse.  mask (Csc pattern, m x n): the user mask of
    fit_cv.hpp:327-331 -- its entries leave every half-update and both losses.  Returns W_T (normalised), H WITH d
    absorbed and d, as the reference packages them.  loss_type != 0 (4 GP, 5 NB, 6 Gamma, 7 inverse Gaussian, 8 Tweedie) or
    robust_delta > 0: the IRLS path (no graph arguments there); result.theta = GP theta at exit."""
    suf, ct = _suf(dtype)
    if mask is not None:
        assert graph_H is None and graph_W is None
        W_T = _f(W_T, dtype).copy()
        H = _f(H, dtype).copy()
        m, k = W_T.shape
        n = H.shape[0]
        d = np.ones(k, dtype)
        x = A.values(dtype)
        mx = np.ones(mask.i.shape[0], dtype)
        th = np.full(max(max_iter, 1), np.nan, dtype)
        eh = np.full(max(max_iter, 1), np.nan, dtype)
        theta = np.zeros(m, dtype)
        it, conv, bi = C.c_int(0), C.c_int(0), C.c_int(0)
        tr, te, bt = ct(0), ct(0), ct(0)
        getattr(lib(native), "oracle_nmf_fit_cv_masked_" + suf)(
            C.c_int(m), C.c_int(n), _p(A.p), _p(A.i), _p(x), C.c_int(k), _p(W_T), _p(H), _p(d), C.c_int(max_iter), ct(tol),
            ct(L1[1]), ct(L1[0]), ct(L2[1]), ct(L2[0]), C.c_int(cd_maxit), C.c_int(int(nonneg[0])), C.c_int(int(nonneg[1])),
            C.c_int(norm_type), C.c_int(solver_mode), C.c_double(holdout_fraction), C.c_uint64(cv_seed), C.c_int(int(mask_zeros)),
            C.c_int(cv_patience), C.c_int(threads), C.c_int(loss_type), C.c_int(irls_max_iter), ct(irls_tol), C.c_int(dispersion_mode),
            ct(gp_theta[0]), ct(gp_theta[1]), ct(tweedie_power), ct(robust_delta), _p(mask.p), _p(mask.i), _p(mx),
            C.byref(it), C.byref(conv), C.byref(tr), C.byref(te), C.byref(bt), C.byref(bi), _p(th), _p(eh), _p(theta))
        r = CvFitResult()
        r.W_T, r.H, r.d, r.theta = W_T, H, d, theta
        r.iter, r.converged = it.value, bool(conv.value)
        r.train_loss, r.test_loss, r.best_test_loss, r.best_iter = float(tr.value), float(te.value), float(bt.value), bi.value
        r.train_history, r.test_history = th[:it.value].copy(), eh[:it.value].copy()
        return r
    if loss_type != 0 or robust_delta > 0:
        assert graph_H is None and graph_W is None
        W_T = _f(W_T, dtype).copy()
        H = _f(H, dtype).copy()
        m, k = W_T.shape
        n = H.shape[0]
        d = np.ones(k, dtype)
        x = A.values(dtype)
        th = np.full(max(max_iter, 1), np.nan, dtype)
        eh = np.full(max(max_iter, 1), np.nan, dtype)
        theta = np.zeros(m, dtype)
        it, conv, bi = C.c_int(0), C.c_int(0), C.c_int(0)
        tr, te, bt = ct(0), ct(0), ct(0)
        getattr(lib(native), "oracle_nmf_fit_cv_irls_" + suf)(
            C.c_int(m), C.c_int(n), _p(A.p), _p(A.i), _p(x), C.c_int(k), _p(W_T), _p(H), _p(d), C.c_int(max_iter), ct(tol),
            ct(L1[1]), ct(L1[0]), ct(L2[1]), ct(L2[0]), C.c_int(cd_maxit), C.c_int(int(nonneg[0])), C.c_int(int(nonneg[1])),
            C.c_int(norm_type), C.c_int(solver_mode), C.c_double(holdout_fraction), C.c_uint64(cv_seed), C.c_int(int(mask_zeros)),
            C.c_int(cv_patience), C.c_int(threads), C.c_int(loss_type), C.c_int(irls_max_iter), ct(irls_tol), C.c_int(dispersion_mode),
            ct(gp_theta[0]), ct(gp_theta[1]), ct(tweedie_power), ct(robust_delta),
            C.byref(it), C.byref(conv), C.byref(tr), C.byref(te), C.byref(bt), C.byref(bi), _p(th), _p(eh), _p(theta))
        r = CvFitResult()
        r.W_T, r.H, r.d, r.theta = W_T, H, d, theta
        r.iter, r.converged = it.value, bool(conv.value)
        r.train_loss, r.test_loss, r.best_test_loss, r.best_iter = float(tr.value), float(te.value), float(bt.value), bi.value
        r.train_history, r.test_history = th[:it.value].copy(), eh[:it.value].copy()
        return r
    W_T = _f(W_T, dtype).copy()
    H = _f(H, dtype).copy()
    m, k = W_T.shape
    n = H.shape[0]
    d = np.ones(k, dtype)
    x = A.values(dtype)
    th = np.full(max(max_iter, 1), np.nan, dtype)
    eh = np.full(max(max_iter, 1), np.nan, dtype)
    it, conv, bi = C.c_int(0), C.c_int(0), C.c_int(0)
    tr, te, bt = ct(0), ct(0), ct(0)
    getattr(lib(native), "oracle_nmf_fit_cv_" + suf)(
        C.c_int(m), C.c_int(n), _p(A.p), _p(A.i), _p(x), C.c_int(k), _p(W_T), _p(H), _p(d), C.c_int(max_iter), ct(tol),
        ct(L1[1]), ct(L1[0]), ct(L2[1]), ct(L2[0]), C.c_int(cd_maxit), C.c_int(int(nonneg[0])), C.c_int(int(nonneg[1])),
        C.c_int(norm_type), C.c_int(solver_mode), C.c_double(holdout_fraction), C.c_uint64(cv_seed), C.c_int(int(mask_zeros)),
        C.c_int(cv_patience), C.c_int(threads), C.byref(it), C.byref(conv), C.byref(tr), C.byref(te), C.byref(bt), C.byref(bi),
        _p(th), _p(eh), *_graph_args(graph_H, dtype, ct), *_graph_args(graph_W, dtype, ct))
    r = CvFitResult()
    r.W_T, r.H, r.d = W_T, H, d
    r.iter, r.converged = it.value, bool(conv.value)
    r.train_loss, r.test_loss, r.best_test_loss, r.best_iter = float(tr.value), float(te.value), float(bt.value), bi.value
    r.train_history, r.test_history = th[:it.value].copy(), eh[:it.value].copy()
    return r


# ----------------------------------------------------------------------------- StreamPress v2 decoder (spz_oracle.cpp)
_spz = None


def spz_lib():
    global _spz
    if _spz is None:
        path = os.path.join(_LIBDIR, "libspz_oracle.so")
        if not os.path.exists(path):
            if _LIBDIR != _HERE:
                raise FileNotFoundError(path)
            build()
        _spz = C.CDLL(path)
    return _spz


def spz_info(buf):
    """(status, m, n, nnz, value_type) of a .spz v2 byte stream (uint8 array)."""
    buf = np.ascontiguousarray(buf, np.uint8)
    m, n, nnz, vt = C.c_uint32(), C.c_uint32(), C.c_uint64(), C.c_int()
    st = spz_lib().oracle_spz_info(buf.ctypes.data_as(C.c_void_p), C.c_uint64(buf.size), C.byref(m), C.byref(n), C.byref(nnz), C.byref(vt))
    return st, m.value, n.value, nnz.value, vt.value


def spz_decode(buf):
    """Decode a .spz v2 byte stream -> (p uint32 (n+1), i uint32 (nnz), x float64 (nnz)); raises on a bad file."""
    buf = np.ascontiguousarray(buf, np.uint8)
    st, m, n, nnz, vt = spz_info(buf)
    if st != 0:
        raise ValueError("spz_info status %d" % st)
    p, i, x = np.zeros(n + 1, np.uint32), np.zeros(nnz, np.uint32), np.zeros(nnz, np.float64)
    st = spz_lib().oracle_spz_decode(buf.ctypes.data_as(C.c_void_p), C.c_uint64(buf.size), p.ctypes.data_as(C.c_void_p),
                                     i.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p))
    if st != 0:
        raise ValueError("spz_decode status %d" % st)
    return p, i, x
