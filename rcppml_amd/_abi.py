"""ctypes binding of rcppml_amd/lib/RcppML_gpu.so (C ABI declared in include/rcppml_gpu.h).

This is the ONLY compute backend of the package: there is no CPU or PyTorch fallback.  If the
shared library is missing or cannot be loaded, importing the symbols raises immediately.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RCPPML_GPU_LIB_PATH") or os.path.join(_HERE, "lib", "RcppML_gpu.so")
_lib = None

F32, F64 = 0, 1
CD_AUTO, CD_LANE, CD_WAVE, CD_GROUP, CD_MFMA, CD_MFMA16, CD_LMF = 0, 1, 2, 5, 6, 7, 8
# rcppml_hip_ctx_set_option
OPT_CD_COUNT_NOOP, OPT_CD_LMF_LANE_GROUPS, OPT_CD_LMF_WAVES_PER_SIMD, OPT_CD_NO_LMF, OPT_IRLS_COLUMNS_PER_WAVE, OPT_SMALL_GIVE_UP = 1, 2, 3, 4, 5, 6

# Every symbol include/rcppml_gpu.h declares (tests check the library exports all of them).
EXPORTED_SYMBOLS = [
    "rcppml_gpu_detect", "rcppml_gpu_nmf_unified_float", "rcppml_gpu_nmf_unified_double", "rcppml_gpu_nmf_ex",
    "rcppml_gpu_nmf_cv_unified_float", "rcppml_gpu_nmf_cv_unified_double", "rcppml_gpu_nmf_cv_ex", "rcppml_gpu_nmf_cv_irls_ex", "rcppml_gpu_nmf_cv_masked_ex", "rcppml_hip_ctx_set_cv_mask", "rcppml_gpu_nmf_zerocopy_double",
    "rcppml_gpu_nnls_double", "rcppml_gpu_evaluate_mse_double", "rcppml_gpu_nmf_profile_double", "rcppml_gpu_last_error",
    "rcppml_hip_ctx_create", "rcppml_hip_ctx_destroy", "rcppml_hip_ctx_sync", "rcppml_hip_ctx_stats", "rcppml_hip_ctx_irls_stats", "rcppml_hip_ctx_irls_sweep_stats", "rcppml_hip_ctx_cd_step_stats", "rcppml_hip_ctx_set_option", "rcppml_hip_transpose_csc", "rcppml_hip_transpose_csc_sort", "rcppml_hip_transpose_csc_gather", "rcppml_hip_cast", "rcppml_hip_gram", "rcppml_hip_rhs",
    "rcppml_hip_solve_cd", "rcppml_hip_order_columns", "rcppml_hip_solve_chol", "rcppml_hip_row_norms", "rcppml_hip_apply_scaling",
    "rcppml_hip_sumsq", "rcppml_hip_loss_mse", "rcppml_hip_solve_masked", "rcppml_hip_loss_nonzeros", "rcppml_hip_loss_masked",
    "rcppml_hip_solve_irls_nb", "rcppml_hip_nb_size_update", "rcppml_hip_nb_size_update_loss", "rcppml_hip_nb_loss", "rcppml_hip_solve_irls", "rcppml_hip_rhs_plan_kind", "rcppml_hip_irls_loss", "rcppml_hip_apply_l21", "rcppml_hip_angular_posthoc", "rcppml_hip_solve_cv", "rcppml_hip_cv_test_error", "rcppml_hip_solve_cv_irls", "rcppml_hip_cv_irls_loss", "rcppml_hip_cv_gp_theta_update", "rcppml_hip_mul_rows", "rcppml_hip_apply_graph_reg", "rcppml_hip_dispersion_update", "rcppml_hip_vec_global", "rcppml_hip_spz_info", "rcppml_hip_spz_decode",
    "rcppml_sp_read_gpu", "rcppml_sp_free_gpu", "rcppml_hip_rhs_dense", "rcppml_gpu_nmf_dense_unified_float",
    "rcppml_gpu_nmf_dense_unified_double",
    "rcppml_hip_rhs_plan_create", "rcppml_hip_rhs_plan_create_indices", "rcppml_hip_rhs_plan_set_values", "rcppml_hip_rhs_plan_destroy", "rcppml_hip_rhs_plan_info", "rcppml_hip_rhs_planned",
    "rcppml_gpu_nmf_target", "rcppml_hip_axpy", "rcppml_hip_add_diag", "rcppml_hip_clip_upper",
    "rcppml_hip_scale_order", "rcppml_hip_gram_loss_mse", "rcppml_hip_tail_scale_gram", "rcppml_hip_tail_scale_gram_loss",
    "rcppml_hip_als_small_eligible", "rcppml_hip_als_small_fit",
]


class BackendError(RuntimeError):
    pass


class RhsPlan:
    """Owner of a rcppml_rhs_plan handle (device memory: free it with close() or let the GC do it)."""

    def __init__(self, handle):
        self._h = handle

    def info(self):
        out = (C.c_double * 11)()
        lib().rcppml_hip_rhs_plan_info(self._h, out)
        keys = ("partitions", "waves", "rounds", "slots", "workgroups_per_partition", "tiles", "slot_count", "spilled_nnz",
                "fill", "stream_bytes", "tiled_columns")
        d = {k: (float(out[i]) if k == "fill" else int(out[i])) for i, k in enumerate(keys)}
        d["kind"] = {1: "window", 0: "slab"}.get(int(lib().rcppml_hip_rhs_plan_kind(self._h)), "?")
        d["slot_rate"] = float(out[3])        # window plans: slots per column and phase (fractional); slab plans: S
        return d

    def close(self):
        if self._h is not None and self._h.value:
            lib().rcppml_hip_rhs_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def lib():
    """Load RcppML_gpu.so (RTLD_GLOBAL, as R's dyn.load(local=FALSE) does: reference R/gpu_backend.R:87)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BackendError(
                "HIP backend missing: %s not built (run `python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C rcppml_amd/csrc`).  There is no CPU fallback." % LIB_PATH)
        try:
            _lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        except OSError as e:  # e.g. libamdhip64 missing
            raise BackendError("cannot load %s: %s" % (LIB_PATH, e))
        _lib.rcppml_gpu_last_error.restype = C.c_char_p
        for name in ("rcppml_hip_ctx_create", "rcppml_hip_ctx_sync", "rcppml_hip_ctx_stats", "rcppml_hip_ctx_set_option", "rcppml_hip_transpose_csc", "rcppml_hip_cast", "rcppml_hip_gram", "rcppml_hip_rhs",
                     "rcppml_hip_solve_cd", "rcppml_hip_order_columns", "rcppml_hip_solve_chol", "rcppml_hip_row_norms",
                     "rcppml_hip_apply_scaling",
                     "rcppml_hip_sumsq", "rcppml_hip_loss_mse", "rcppml_hip_solve_masked", "rcppml_hip_loss_nonzeros", "rcppml_hip_loss_masked",
                     "rcppml_hip_solve_irls_nb", "rcppml_hip_nb_size_update", "rcppml_hip_nb_loss", "rcppml_hip_scale_order",
                     "rcppml_hip_gram_loss_mse", "rcppml_hip_tail_scale_gram", "rcppml_hip_tail_scale_gram_loss"):
            getattr(_lib, name).restype = C.c_int
        _lib.rcppml_hip_ctx_destroy.restype = None
        _lib.rcppml_hip_rhs_plan_destroy.restype = None
        _lib.rcppml_hip_rhs_plan_destroy.argtypes = [C.c_void_p]
        for name in ("rcppml_hip_rhs_plan_create", "rcppml_hip_rhs_plan_create_indices", "rcppml_hip_rhs_plan_set_values",
                     "rcppml_hip_rhs_plan_info", "rcppml_hip_rhs_planned"):
            getattr(_lib, name).restype = C.c_int
        _lib.rcppml_hip_rhs_plan_info.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    return _lib


def small_eligible(m, n, nnz, k):
    """True when the one-kernel fit (rcppml_hip_als_small_fit) takes a plain sparse MSE fit of this size."""
    f = lib().rcppml_hip_als_small_eligible
    f.restype = C.c_int
    return bool(f(C.c_int(m), C.c_int(n), C.c_int64(nnz), C.c_int(k)))


def last_error():
    return lib().rcppml_gpu_last_error().decode("utf-8", "replace")


def _chk(rc, what):
    if rc != 0:
        raise BackendError("%s failed: %s" % (what, last_error()))


# ----------------------------------------------------------------------------- plugin boundary
def detect(max_gpus=8):
    """rcppml_gpu_detect -> list of (total_mb, free_mb); [] if no device (reference R/gpu_backend.R:101-106)."""
    n, st, mx = C.c_int(0), C.c_int(0), C.c_int(max_gpus)
    tot = (C.c_double * max_gpus)()
    fre = (C.c_double * max_gpus)()
    lib().rcppml_gpu_detect(C.byref(n), tot, fre, C.byref(mx), C.byref(st))
    if st.value != 0:
        return []
    return [(tot[i], fre[i]) for i in range(n.value)]


def _ci(v):
    return C.byref(C.c_int(int(v)))


def _cd(v):
    return C.byref(C.c_double(float(v)))


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def nmf_unified(p, i, x, m, n, k, W_T, H, *, entry="float", max_iter=100, tol=1e-4, L1_H=0.0, L1_W=0.0, L2_H=0.0,
                L2_W=0.0, L21_H=0.0, L21_W=0.0, ortho_H=0.0, ortho_W=0.0, ub_H=0.0, ub_W=0.0, cd_maxit=100, verbose=0,
                seed=0, loss_every=1, patience=5, nonneg_W=1, nonneg_H=1, loss_type=0, huber_delta=1.0, irls_max_iter=5,
                irls_tol=1e-4, norm_type=0, projective=0, symmetric=0, solver_mode=0, gp_dispersion_mode=2,
                nb_size=(10.0, 1e6, 0.01), mask=None, cd_tol=1e-8, sort_model=1, precision=F64, want_history=False,
                graph_W_nnz=0, guide_H_count=0, tweedie_power=1.5, robust_delta=0.0, graph_W=None, graph_H=None,
                gp_theta=(0.1, 5.0, 0.0), gamma_phi=(1.0, 1e4, 1e-6), target_H=None, target_W=None, theta_capacity=None):
    """Call the 73-pointer plugin entry exactly as reference gpu/bridge_nmf.hpp:310-342 does.

    p, i: int32 CSC arrays; x: float64 values.  W_T (m, k) and H (n, k) float64 arrays (memory = column-major
    k x m / k x n) are updated IN PLACE.  entry: "float" | "double" (the reference symbols) or "ex" (build-defined,
    adds mask / cd_tol / sort / precision / loss history).  target_H / target_W = (matrix (n, k) / (m, k), lambda): target
    regularisation through the build-defined rcppml_gpu_nmf_target (entry "ex" arguments + targets).
    Returns dict(d, iter, converged, loss, tol, status, ...).
    """
    L = lib()
    p = np.ascontiguousarray(p, np.int32)
    i = np.ascontiguousarray(i, np.int32)
    x = np.ascontiguousarray(x, np.float64)
    assert W_T.dtype == np.float64 and H.dtype == np.float64 and W_T.flags.c_contiguous and H.flags.c_contiguous
    assert W_T.shape == (m, k) and H.shape == (n, k)
    d = np.ones(k, np.float64)
    dummy_i = np.zeros(2, np.int32)
    dummy_d = np.zeros(2, np.float64)
    # out_theta: m doubles in the reference bridge (gpu/bridge_nmf.hpp:284); the build-defined "ex" entries take max(m, n) so that
    # dispersion = "per_col" (gp_dispersion_mode 3) can return its n values
    # (theta_capacity: what the caller's buffer holds -- tests hand the "ex" entry the bridge's m doubles)
    theta = np.zeros(max(m, n, 1) if entry == "ex" else max(m, 1), np.float64)
    if theta_capacity is not None:
        theta = np.zeros(max(int(theta_capacity), 1), np.float64)
    # (the build-defined entries read *out_theta_len on input as the capacity of out_theta)
    out_iter, out_conv, out_status, out_theta_len = C.c_int(0), C.c_int(0), C.c_int(-99), C.c_int((theta.shape[0] if theta_capacity is None else int(theta_capacity)) if entry == "ex" else 0)
    out_loss, out_tol = C.c_double(0), C.c_double(0)
    args = [
        _np_ptr(p), _np_ptr(i), _np_ptr(x), _ci(m), _ci(n), _ci(x.shape[0]), _ci(k),
        _np_ptr(W_T), _np_ptr(H), _np_ptr(d), _ci(max_iter), _cd(tol),
        _cd(L1_H), _cd(L1_W), _cd(L2_H), _cd(L2_W), _cd(L21_H), _cd(L21_W), _cd(ortho_H), _cd(ortho_W),
        _cd(ub_H), _cd(ub_W), _ci(cd_maxit), _ci(verbose), _ci(seed), _ci(loss_every), _ci(patience),
        _ci(nonneg_W), _ci(nonneg_H), _ci(loss_type), _cd(huber_delta), _ci(irls_max_iter), _cd(irls_tol),
        _ci(norm_type), _ci(projective), _ci(symmetric), _ci(solver_mode),
        _np_ptr(dummy_i), _np_ptr(dummy_i), _np_ptr(dummy_d), _ci(0), _ci(graph_W_nnz), _cd(0.0),
        _np_ptr(dummy_i), _np_ptr(dummy_i), _np_ptr(dummy_d), _ci(0), _ci(0), _cd(0.0),
        _ci(gp_dispersion_mode), _cd(gp_theta[0]), _cd(gp_theta[1]), _cd(gp_theta[2]), _cd(nb_size[0]), _cd(nb_size[1]), _cd(nb_size[2]),
        _cd(gamma_phi[0]), _cd(gamma_phi[1]), _cd(gamma_phi[2]), _cd(robust_delta), _cd(tweedie_power),
        _np_ptr(theta), C.byref(out_theta_len),
        _np_ptr(dummy_i), _np_ptr(dummy_i), _np_ptr(dummy_d), _np_ptr(dummy_i), _ci(guide_H_count),
        C.byref(out_iter), C.byref(out_conv), C.byref(out_loss), C.byref(out_status), C.byref(out_tol),
    ]
    # graph_W / graph_H: (p, i, x, lambda) CSC Laplacians (m x m / n x n), reference bridge_nmf.hpp graph_* slots
    for slot, g, dim in ((37, graph_W, m), (43, graph_H, n)):
        if g is not None:
            gp, gi, gx, lam = g
            gp = np.ascontiguousarray(gp, np.int32); gi = np.ascontiguousarray(gi, np.int32); gx = np.ascontiguousarray(gx, np.float64)
            args[slot:slot + 6] = [_np_ptr(gp), _np_ptr(gi), _np_ptr(gx), _ci(gp.shape[0] - 1), _ci(gx.shape[0]), _cd(lam)]
            args.append((gp, gi, gx))          # keep alive; popped below
    keep = args[73:]
    del args[73:]
    assert len(args) == 73
    hist = None
    if entry == "float":
        L.rcppml_gpu_nmf_unified_float(*args)
    elif entry == "double":
        L.rcppml_gpu_nmf_unified_double(*args)
    elif entry == "ex":
        if mask is not None:
            mp = np.ascontiguousarray(mask[0], np.int32)
            mi = np.ascontiguousarray(mask[1], np.int32)
            mnnz = int(mi.shape[0])
        else:
            mp, mi, mnnz = dummy_i, dummy_i, 0
        hist = np.full(max(max_iter, 1), np.nan) if want_history else None
        if target_H is not None or target_W is not None:
            tH = np.ascontiguousarray(target_H[0], np.float64) if target_H is not None else None
            tW = np.ascontiguousarray(target_W[0], np.float64) if target_W is not None else None
            assert tH is None or tH.shape == (n, k)
            assert tW is None or tW.shape == (m, k)
            L.rcppml_gpu_nmf_target(*args, _np_ptr(mp), _np_ptr(mi), _ci(mnnz), _cd(cd_tol), _ci(sort_model), _ci(precision),
                                    _np_ptr(hist) if hist is not None else None,
                                    _np_ptr(tH) if tH is not None else None, _cd(target_H[1] if target_H is not None else 0.0),
                                    _np_ptr(tW) if tW is not None else None, _cd(target_W[1] if target_W is not None else 0.0))
        else:
            L.rcppml_gpu_nmf_ex(*args, _np_ptr(mp), _np_ptr(mi), _ci(mnnz), _cd(cd_tol), _ci(sort_model), _ci(precision),
                                _np_ptr(hist) if hist is not None else None)
    else:
        raise ValueError(entry)
    res = dict(d=d, iter=out_iter.value, converged=bool(out_conv.value), loss=out_loss.value, tol=out_tol.value,
               status=out_status.value, theta=theta[:out_theta_len.value].copy())
    if hist is not None:
        res["loss_history"] = hist[:out_iter.value].copy()
    if out_status.value != 0:
        res["error"] = last_error()
    return res


def nnls_double(p, i, x, m, n, k, w_T, h, *, cd_maxit=100, cd_tol=1e-8, L1=0.0, L2=0.0, ub=0.0, nonneg=1, warm=0):
    st = C.c_int(-99)
    p = np.ascontiguousarray(p, np.int32); i = np.ascontiguousarray(i, np.int32); x = np.ascontiguousarray(x, np.float64)
    assert w_T.shape == (m, k) and h.shape == (n, k) and w_T.dtype == np.float64 and h.dtype == np.float64
    lib().rcppml_gpu_nnls_double(_np_ptr(p), _np_ptr(i), _np_ptr(x), _ci(m), _ci(n), _ci(x.shape[0]), _ci(k),
                                 _np_ptr(np.ascontiguousarray(w_T)), _np_ptr(h), _ci(cd_maxit), _cd(cd_tol), _cd(L1),
                                 _cd(L2), _cd(ub), _ci(nonneg), _ci(warm), C.byref(st))
    if st.value != 0:
        raise BackendError("rcppml_gpu_nnls_double: " + last_error())
    return h


def evaluate_mse_double(p, i, x, m, n, k, W_T, d, H, mask_zeros=False):
    st, out = C.c_int(-99), C.c_double(0)
    p = np.ascontiguousarray(p, np.int32); i = np.ascontiguousarray(i, np.int32); x = np.ascontiguousarray(x, np.float64)
    lib().rcppml_gpu_evaluate_mse_double(_np_ptr(p), _np_ptr(i), _np_ptr(x), _ci(m), _ci(n), _ci(x.shape[0]), _ci(k),
                                         _np_ptr(np.ascontiguousarray(W_T, np.float64)),
                                         _np_ptr(np.ascontiguousarray(d, np.float64)),
                                         _np_ptr(np.ascontiguousarray(H, np.float64)), _ci(int(mask_zeros)),
                                         C.byref(out), C.byref(st))
    if st.value != 0:
        raise BackendError("rcppml_gpu_evaluate_mse_double: " + last_error())
    return out.value


PROFILE_PHASES = ("gram_H", "rhs_H", "nnls_H", "norm_H", "gram_W", "rhs_W_gather", "rhs_W_planned", "nnls_W", "norm_W", "loss", "total")


def nmf_profile_double(p, i, x, m, n, k, max_iter=10, tol=0.0, cd_maxit=10, seed=42):
    """rcppml_gpu_nmf_profile_double (reference src/gpu_bridge_utils.cu:48): per-phase HIP-event times of the batch-CD ALS
    iteration.  Returns dict(total_ms, per_iter_ms: {phase: ms}, iters)."""
    p = np.ascontiguousarray(p, np.int32); i = np.ascontiguousarray(i, np.int32); x = np.ascontiguousarray(x, np.float64)
    tot, per = np.zeros(11), np.zeros(11)
    it, st = C.c_int(0), C.c_int(-99)
    lib().rcppml_gpu_nmf_profile_double(_np_ptr(p), _np_ptr(i), _np_ptr(x), _ci(m), _ci(n), _ci(x.shape[0]), _ci(k), _ci(max_iter),
                                        _cd(tol), _ci(cd_maxit), _ci(seed), _np_ptr(tot), _np_ptr(per), C.byref(it), C.byref(st))
    if st.value != 0:
        raise BackendError("rcppml_gpu_nmf_profile_double: " + last_error())
    return dict(total_ms=dict(zip(PROFILE_PHASES, tot.tolist())), per_iter_ms=dict(zip(PROFILE_PHASES, per.tolist())), iters=it.value)


# ----------------------------------------------------------------------------- device-level ops
def _dptr(t):
    """Device pointer of a torch tensor (or raw int / None)."""
    if t is None:
        return None
    if isinstance(t, int):
        return C.c_void_p(t)
    return C.c_void_p(t.data_ptr())


def nmf_cv(p, i, x, m, n, k, W_T, H, *, entry="ex", max_iter=100, tol=1e-4, L1_H=0.0, L1_W=0.0, L2_H=0.0, L2_W=0.0, cd_maxit=100,
           verbose=0, seed=0, holdout_fraction=0.1, cv_seed=0, mask_zeros=0, nonneg_W=1, nonneg_H=1, norm_type=0, loss_type=0,
           solver_mode=0, projective=0, symmetric=0, graph_W_nnz=0, sort_model=1, precision=F64, cv_patience=5,
           graph_W=None, graph_H=None, irls_max_iter=5, irls_tol=1e-4, dispersion_mode=2, gp_theta=(0.1, 5.0), tweedie_power=1.5,
           robust_delta=0.0, mask=None):
    """Call the CV plugin entry as reference gpu/bridge_nmf.hpp:407-497 does (51 pointers; entry "float" | "double"), or
    the build-defined "ex" form (+ sort flag, precision, patience, loss histories) or "irls_ex" (+ dispersion mode, GP theta
    init / max, Tweedie power, robust_delta; returns theta), or -- mask = (mask_p, mask_i) given -- "masked_ex" (irls_ex + the user mask).  W_T (m, k) and H (n, k) float64 arrays are updated IN PLACE (H
    returns with d absorbed).  loss_type 4..8: the IRLS CV path."""
    L = lib()
    p = np.ascontiguousarray(p, np.int32)
    i = np.ascontiguousarray(i, np.int32)
    x = np.ascontiguousarray(x, np.float64)
    assert W_T.dtype == np.float64 and H.dtype == np.float64 and W_T.flags.c_contiguous and H.flags.c_contiguous
    assert W_T.shape == (m, k) and H.shape == (n, k)
    d = np.ones(k, np.float64)
    dummy_i = np.zeros(2, np.int32)
    dummy_d = np.zeros(2, np.float64)
    out_iter, out_conv, out_best_iter, out_status = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(-99)
    out_train, out_test, out_best = C.c_double(0), C.c_double(0), C.c_double(0)
    args = [
        _np_ptr(p), _np_ptr(i), _np_ptr(x), _ci(m), _ci(n), _ci(x.shape[0]), _ci(k),
        _np_ptr(W_T), _np_ptr(H), _np_ptr(d), _ci(max_iter), _cd(tol),
        _cd(L1_H), _cd(L1_W), _cd(L2_H), _cd(L2_W), _ci(cd_maxit), _ci(verbose), _ci(seed),
        _cd(holdout_fraction), _ci(cv_seed), _ci(mask_zeros), _ci(nonneg_W), _ci(nonneg_H), _ci(norm_type),
        _ci(loss_type), _cd(1.0), _ci(irls_max_iter), _cd(irls_tol),
        _np_ptr(dummy_i), _np_ptr(dummy_i), _np_ptr(dummy_d), _ci(0), _ci(graph_W_nnz), _cd(0.0),
        _np_ptr(dummy_i), _np_ptr(dummy_i), _np_ptr(dummy_d), _ci(0), _ci(0), _cd(0.0),
        _ci(projective), _ci(symmetric), _ci(solver_mode),
        C.byref(out_iter), C.byref(out_conv), C.byref(out_train), C.byref(out_test), C.byref(out_best), C.byref(out_best_iter),
        C.byref(out_status),
    ]
    keep = []
    for slot, g in ((29, graph_W), (35, graph_H)):          # (p, i, x, lambda) CSC Laplacians, as in nmf_unified
        if g is not None:
            gp, gi, gx, lam = g
            gp = np.ascontiguousarray(gp, np.int32); gi = np.ascontiguousarray(gi, np.int32); gx = np.ascontiguousarray(gx, np.float64)
            args[slot:slot + 6] = [_np_ptr(gp), _np_ptr(gi), _np_ptr(gx), _ci(gp.shape[0] - 1), _ci(gx.shape[0]), _cd(lam)]
            keep.append((gp, gi, gx))
    assert len(args) == 51
    th = eh = None
    theta = None
    if mask is not None:
        mp = np.ascontiguousarray(mask[0], np.int32); mi = np.ascontiguousarray(mask[1], np.int32)
        if mi.shape[0] == 0:
            mi = np.zeros(1, np.int32)
        th = np.full(max(max_iter, 1), np.nan)
        eh = np.full(max(max_iter, 1), np.nan)
        theta = np.zeros(max(m, 1), np.float64)
        fn = L.rcppml_gpu_nmf_cv_masked_ex
        fn.restype = None
        fn(*args, _ci(sort_model), _ci(precision), _ci(cv_patience), _np_ptr(th), _np_ptr(eh), _ci(dispersion_mode), _cd(gp_theta[0]),
           _cd(gp_theta[1]), _cd(tweedie_power), _cd(robust_delta), _np_ptr(theta), _np_ptr(mp), _np_ptr(mi), _ci(int(mp[-1])))
    elif entry == "irls_ex":
        th = np.full(max(max_iter, 1), np.nan)
        eh = np.full(max(max_iter, 1), np.nan)
        theta = np.zeros(max(m, 1), np.float64)
        fn = L.rcppml_gpu_nmf_cv_irls_ex
        fn.restype = None
        fn(*args, _ci(sort_model), _ci(precision), _ci(cv_patience), _np_ptr(th), _np_ptr(eh), _ci(dispersion_mode), _cd(gp_theta[0]),
           _cd(gp_theta[1]), _cd(tweedie_power), _cd(robust_delta), _np_ptr(theta))
    elif entry == "ex":
        th = np.full(max(max_iter, 1), np.nan)
        eh = np.full(max(max_iter, 1), np.nan)
        fn = L.rcppml_gpu_nmf_cv_ex
        fn.restype = None
        fn(*args, _ci(sort_model), _ci(precision), _ci(cv_patience), _np_ptr(th), _np_ptr(eh))
    else:
        fn = getattr(L, "rcppml_gpu_nmf_cv_unified_" + entry)
        fn.restype = None
        fn(*args)
    res = dict(status=out_status.value, iter=out_iter.value, converged=bool(out_conv.value), train_loss=out_train.value,
               test_loss=out_test.value, best_test_loss=out_best.value, best_iter=out_best_iter.value, d=d)
    if out_status.value != 0:
        res["error"] = last_error()
    if th is not None:
        res["train_history"], res["test_history"] = th[:out_iter.value].copy(), eh[:out_iter.value].copy()
    if theta is not None:
        res["theta"] = theta
    return res


def nmf_zerocopy(d_col_ptr, d_row_idx, d_values, m, n, nnz, k, W_T, H, *, max_iter=100, tol=1e-4, L1_H=0.0, L1_W=0.0, L2_H=0.0,
                 L2_W=0.0, L21_H=0.0, L21_W=0.0, ortho_H=0.0, ortho_W=0.0, ub_H=0.0, ub_W=0.0, cd_maxit=100, verbose=0, seed=0,
                 patience=5, nonneg_W=1, nonneg_H=1, loss_type=0, norm_type=0):
    """Call rcppml_gpu_nmf_zerocopy_double as R/sp_gpu.R does: d_col_ptr / d_row_idx (int32) and d_values (float64) are
    DEVICE tensors (torch) or raw device addresses; addresses travel as doubles."""
    L = lib()

    def addr(t):
        return float(t.data_ptr()) if hasattr(t, "data_ptr") else float(t)

    d = np.ones(k, np.float64)
    out_iter, out_conv, out_status = C.c_int(0), C.c_int(0), C.c_int(-99)
    out_loss, out_tol = C.c_double(0), C.c_double(0)
    fn = L.rcppml_gpu_nmf_zerocopy_double
    fn.restype = None
    fn(_cd(addr(d_col_ptr)), _cd(addr(d_row_idx)), _cd(addr(d_values)), _ci(m), _ci(n), _cd(float(nnz)), _ci(k),
       _np_ptr(W_T), _np_ptr(H), _np_ptr(d), _ci(max_iter), _cd(tol), _cd(L1_H), _cd(L1_W), _cd(L2_H), _cd(L2_W), _cd(L21_H),
       _cd(L21_W), _cd(ortho_H), _cd(ortho_W), _cd(ub_H), _cd(ub_W), _ci(cd_maxit), _ci(verbose), _ci(seed), _ci(1),
       _ci(patience), _ci(nonneg_W), _ci(nonneg_H), _ci(loss_type), _cd(1.0), _ci(5), _cd(1e-4), _ci(norm_type),
       C.byref(out_iter), C.byref(out_conv), C.byref(out_loss), C.byref(out_status), C.byref(out_tol))
    res = dict(status=out_status.value, iter=out_iter.value, converged=bool(out_conv.value), loss=out_loss.value, tol=out_tol.value, d=d)
    if out_status.value != 0:
        res["error"] = last_error()
    return res


class Context:
    """rcppml_hip_ctx bound to a device and a HIP stream (default: torch's current stream on that device)."""

    def __init__(self, device=0, stream=None):
        self._h = C.c_void_p()
        if stream is None:
            import torch
            stream = torch.cuda.current_stream(device).cuda_stream
        _chk(lib().rcppml_hip_ctx_create(C.byref(self._h), C.c_int(device), C.c_void_p(stream)), "ctx_create")
        self.device = device

    def close(self):
        if self._h:
            lib().rcppml_hip_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _chk(lib().rcppml_hip_ctx_sync(self._h), "ctx_sync")

    def set_option(self, option, value):
        _chk(lib().rcppml_hip_ctx_set_option(self._h, C.c_int(option), C.c_int(value)), "ctx_set_option")

    def stats(self, reset=False):
        """Work counters since creation / the last reset (synchronises): cd_column_sweeps, cd_columns; cd_slot_sweeps (persistent
        LMF kernel: column slots x wave sweeps, idle and correction sweeps included) and cd_noop_steps (OPT_CD_COUNT_NOOP)."""
        out = (C.c_ulonglong * 4)()
        _chk(lib().rcppml_hip_ctx_stats(self._h, C.c_int(1 if reset else 0), out), "ctx_stats")
        return dict(cd_column_sweeps=int(out[0]), cd_columns=int(out[1]), cd_slot_sweeps=int(out[2]), cd_noop_steps=int(out[3]))

    def cd_step_stats(self, reset=False):
        """Per-(column, coordinate) steps of the lane = column CD kernel (only counted while OPT_CD_COUNT_NOOP is set): steps whose
        update is exactly 0 (the reference skips them) and all steps of live columns."""
        out = (C.c_ulonglong * 2)()
        _chk(lib().rcppml_hip_ctx_cd_step_stats(self._h, C.c_int(1 if reset else 0), out), "ctx_cd_step_stats")
        return dict(cd_zero_steps=int(out[0]), cd_steps=int(out[1]))

    def irls_stats(self, reset=False):
        """IRLS work counters (only counted while OPT_CD_COUNT_NOOP is set): passes over columns, nonzero-passes."""
        out = (C.c_ulonglong * 2)()
        _chk(lib().rcppml_hip_ctx_irls_stats(self._h, C.c_int(1 if reset else 0), out), "ctx_irls_stats")
        sw = (C.c_ulonglong * 1)()
        _chk(lib().rcppml_hip_ctx_irls_sweep_stats(self._h, C.c_int(1 if reset else 0), sw), "ctx_irls_sweep_stats")
        return dict(irls_column_passes=int(out[0]), irls_nonzero_passes=int(out[1]), irls_cd_sweeps=int(sw[0]))

    def transpose_csc(self, dt, rows, cols, col_ptr, row_idx, values, t_col_ptr, t_row_idx, t_values):
        _chk(lib().rcppml_hip_transpose_csc(self._h, C.c_int(dt), C.c_int(rows), C.c_int(cols), _dptr(col_ptr), _dptr(row_idx),
                                            _dptr(values), _dptr(t_col_ptr), _dptr(t_row_idx), _dptr(t_values)), "transpose_csc")

    def cast(self, dt_src, src, dt_dst, dst, n):
        _chk(lib().rcppml_hip_cast(self._h, C.c_int(dt_src), _dptr(src), C.c_int(dt_dst), _dptr(dst), C.c_int64(n)), "cast")

    # ---- ops (dt: F32/F64; tensors are torch CUDA tensors laid out (cols, k) == column-major k x cols)
    def gram(self, dt, F, k, r, eps, l2, G):
        _chk(lib().rcppml_hip_gram(self._h, C.c_int(dt), _dptr(F), C.c_int(k), C.c_int64(r), C.c_double(eps),
                                   C.c_double(l2), _dptr(G)), "gram")

    def rhs(self, dt, col_ptr, row_idx, values, ncols, F, k, B):
        _chk(lib().rcppml_hip_rhs(self._h, C.c_int(dt), _dptr(col_ptr), _dptr(row_idx), _dptr(values), C.c_int64(ncols),
                                  _dptr(F), C.c_int(k), _dptr(B)), "rhs")

    def rhs_plan(self, dt, col_ptr, row_idx, values, ncols, nrows, k, partitions=0, slots=0):
        """Tile-partitioned slot copy of one CSC matrix for the LDS row-tiled kernel; None when the shape is not eligible
        (the caller then keeps using rhs())."""
        h = C.c_void_p()
        _chk(lib().rcppml_hip_rhs_plan_create(self._h, C.c_int(dt), _dptr(col_ptr), _dptr(row_idx), _dptr(values), C.c_int64(ncols),
                                              C.c_int64(nrows), C.c_int(k), C.c_int(partitions), C.c_int(slots), C.byref(h)), "rhs_plan_create")
        return RhsPlan(h) if h.value else None

    def rhs_plan_indices(self, dt, col_ptr, row_idx, ncols, nrows, k, partitions=0, slots=0):
        """The index half of a window plan (no values yet); None when the window planner declines.  Then rhs_plan_set_values."""
        h = C.c_void_p()
        _chk(lib().rcppml_hip_rhs_plan_create_indices(self._h, C.c_int(dt), _dptr(col_ptr), _dptr(row_idx), C.c_int64(ncols),
                                                      C.c_int64(nrows), C.c_int(k), C.c_int(partitions), C.c_int(slots), C.byref(h)),
             "rhs_plan_create_indices")
        return RhsPlan(h) if h.value else None

    def rhs_plan_set_values(self, plan, values):
        _chk(lib().rcppml_hip_rhs_plan_set_values(self._h, plan._h, _dptr(values)), "rhs_plan_set_values")

    def rhs_planned(self, plan, F, B):
        _chk(lib().rcppml_hip_rhs_planned(self._h, plan._h, _dptr(F), _dptr(B)), "rhs_planned")

    def solve_cd(self, dt, G, B, X, k, ncols, l1_pre=0.0, warm=0, zero_init=0, l1_cd=0.0, l2_cd=0.0, nonneg=1, maxit=100,
                 tol=1e-8, ub_cd=0.0, ub_post=0.0, variant=CD_AUTO, sweeps_out=None, col_order=None):
        _chk(lib().rcppml_hip_solve_cd(self._h, C.c_int(dt), _dptr(G), _dptr(B), _dptr(X), C.c_int(k), C.c_int64(ncols),
                                       C.c_double(l1_pre), C.c_int(warm), C.c_int(zero_init), C.c_double(l1_cd),
                                       C.c_double(l2_cd), C.c_int(nonneg), C.c_int(maxit), C.c_double(tol),
                                       C.c_double(ub_cd), C.c_double(ub_post), C.c_int(variant), _dptr(sweeps_out), _dptr(col_order)), "solve_cd")

    def order_columns(self, sweeps, ncols, order):
        _chk(lib().rcppml_hip_order_columns(self._h, _dptr(sweeps), C.c_int64(ncols), _dptr(order)), "order_columns")

    def solve_chol(self, dt, G, B, X, k, ncols, l1_pre=0.0, nonneg=1, ub_post=0.0):
        _chk(lib().rcppml_hip_solve_chol(self._h, C.c_int(dt), _dptr(G), _dptr(B), _dptr(X), C.c_int(k), C.c_int64(ncols),
                                         C.c_double(l1_pre), C.c_int(nonneg), C.c_double(ub_post)), "solve_chol")

    def row_norms(self, dt, X, k, ncols, norm_type, out):
        _chk(lib().rcppml_hip_row_norms(self._h, C.c_int(dt), _dptr(X), C.c_int(k), C.c_int64(ncols), C.c_int(norm_type),
                                        _dptr(out)), "row_norms")

    def apply_scaling(self, dt, X, k, ncols, norm_type, sums, d):
        _chk(lib().rcppml_hip_apply_scaling(self._h, C.c_int(dt), _dptr(X), C.c_int(k), C.c_int64(ncols),
                                            C.c_int(norm_type), _dptr(sums), _dptr(d)), "apply_scaling")

    def scale_order(self, dt, X, k, ncols, norm_type, sums, d, sweeps=None, order=None):
        """row_norms + apply_scaling (+ order_columns for the next solve) in three launches instead of five; bit-identical to the separate calls."""
        _chk(lib().rcppml_hip_scale_order(self._h, C.c_int(dt), _dptr(X), C.c_int(k), C.c_int64(ncols), C.c_int(norm_type),
                                          _dptr(sums), _dptr(d), _dptr(sweeps), _dptr(order)), "scale_order")

    def gram_loss_mse(self, dt, W_T, k, m, eps, trAtA, d, B_w, G_saved, G_wt, out):
        """gram(W_T, eps) -> G_wt, then loss_mse with it, in three launches instead of four; bit-identical to the separate calls."""
        _chk(lib().rcppml_hip_gram_loss_mse(self._h, C.c_int(dt), _dptr(W_T), C.c_int(k), C.c_int64(m), C.c_double(eps), _dptr(trAtA),
                                            _dptr(d), _dptr(B_w), _dptr(G_saved), _dptr(G_wt), _dptr(out)), "gram_loss_mse")

    def tail_scale_gram(self, dt, X, k, ncols, norm_type, sums, d, sweeps, order, eps, l2, G):
        """scale_order(X) then gram(X, eps, l2) -> G in one call (fp32 k = 64: the scaling inside the Gram's partial-tile kernel)."""
        _chk(lib().rcppml_hip_tail_scale_gram(self._h, C.c_int(dt), _dptr(X), C.c_int(k), C.c_int64(ncols), C.c_int(norm_type), _dptr(sums),
                                              _dptr(d), _dptr(sweeps), _dptr(order), C.c_double(eps), C.c_double(l2), _dptr(G)), "tail_scale_gram")

    def tail_scale_gram_loss(self, dt, W_T, k, m, norm_type, sums, d, sweeps, order, eps, trAtA, B_w, G_saved, G_wt, out):
        """scale_order(W_T) then gram_loss_mse in one call (fp32 k = 64: the scaling inside the Gram's partial-tile kernel)."""
        _chk(lib().rcppml_hip_tail_scale_gram_loss(self._h, C.c_int(dt), _dptr(W_T), C.c_int(k), C.c_int64(m), C.c_int(norm_type), _dptr(sums),
                                                   _dptr(d), _dptr(sweeps), _dptr(order), C.c_double(eps), _dptr(trAtA), _dptr(B_w),
                                                   _dptr(G_saved), _dptr(G_wt), _dptr(out)), "tail_scale_gram_loss")

    def als_small_fit(self, dt, csc, csc_t, m, n, k, W, H, d, trAtA, *, L1_H=0.0, L1_W=0.0, L2_H=0.0, L2_W=0.0, ub_H=0.0, ub_W=0.0, nonneg_H=1,
                      nonneg_W=1, norm_type=0, solver_mode=0, cd_maxit=100, cd_tol=1e-8, max_iter=100, tol=1e-4, patience=5, iter0=0, loss_history=None,
                      result8=None):
        """The whole plain sparse MSE fit as one persistent kernel (small problems only: small_eligible)."""
        _chk(lib().rcppml_hip_als_small_fit(self._h, C.c_int(dt), _dptr(csc["p"]), _dptr(csc["i"]), _dptr(csc["x"]), _dptr(csc_t["p"]),
                                            _dptr(csc_t["i"]), _dptr(csc_t["x"]), C.c_int(m), C.c_int(n), C.c_int64(csc["nnz"]), C.c_int(k),
                                            _dptr(W), _dptr(H), _dptr(d), _dptr(trAtA), C.c_double(L1_H), C.c_double(L1_W), C.c_double(L2_H),
                                            C.c_double(L2_W), C.c_double(ub_H), C.c_double(ub_W), C.c_int(nonneg_H), C.c_int(nonneg_W),
                                            C.c_int(norm_type), C.c_int(solver_mode), C.c_int(cd_maxit), C.c_double(cd_tol), C.c_int(max_iter),
                                            C.c_double(tol), C.c_int(patience), C.c_int(iter0), _dptr(loss_history), _dptr(result8)), "als_small_fit")

    def sumsq(self, dt, x, length, out):
        _chk(lib().rcppml_hip_sumsq(self._h, C.c_int(dt), _dptr(x), C.c_int64(length), _dptr(out)), "sumsq")

    def loss_mse(self, dt, trAtA, d, W_T, B_w, k, m, G_wt, G_saved, out):
        _chk(lib().rcppml_hip_loss_mse(self._h, C.c_int(dt), _dptr(trAtA), _dptr(d), _dptr(W_T), _dptr(B_w), C.c_int(k),
                                       C.c_int64(m), _dptr(G_wt), _dptr(G_saved), _dptr(out)), "loss_mse")

    def solve_masked(self, dt, col_ptr, row_idx, values, mask_p, mask_i, ncols, F, G_full, X, k, l1=0.0, l2=0.0, nonneg=1,
                     cd_maxit=100, cd_tol=1e-8, solver_mode=0, warm=0):
        _chk(lib().rcppml_hip_solve_masked(self._h, C.c_int(dt), _dptr(col_ptr), _dptr(row_idx), _dptr(values),
                                           _dptr(mask_p), _dptr(mask_i), C.c_int64(ncols), _dptr(F), _dptr(G_full),
                                           _dptr(X), C.c_int(k), C.c_double(l1), C.c_double(l2), C.c_int(nonneg),
                                           C.c_int(cd_maxit), C.c_double(cd_tol), C.c_int(solver_mode), C.c_int(warm)),
             "solve_masked")

    def loss_masked(self, dt, loss_type, col_ptr, row_idx, values, mask_p, mask_i, ncols, W_T, d, H, k, out, power=1.5):
        _chk(lib().rcppml_hip_loss_masked(self._h, C.c_int(dt), C.c_int(loss_type), C.c_double(power), _dptr(col_ptr), _dptr(row_idx),
                                          _dptr(values), _dptr(mask_p), _dptr(mask_i), C.c_int64(ncols), _dptr(W_T), _dptr(d),
                                          _dptr(H), C.c_int(k), _dptr(out)), "loss_masked")

    def loss_nonzeros(self, dt, col_ptr, row_idx, values, mask_p, mask_i, ncols, W_T, d, H, k, out):
        _chk(lib().rcppml_hip_loss_nonzeros(self._h, C.c_int(dt), _dptr(col_ptr), _dptr(row_idx), _dptr(values),
                                            _dptr(mask_p), _dptr(mask_i), C.c_int64(ncols), _dptr(W_T), _dptr(d),
                                            _dptr(H), C.c_int(k), _dptr(out)), "loss_nonzeros")

    def solve_irls_nb(self, dt, col_ptr, row_idx, values, ncols, F, G_base, X, k, l1=0.0, l2=0.0, nonneg=1, cd_maxit=100,
                      irls_max_iter=5, irls_tol=1e-4, theta_row=None, theta_col=None):
        _chk(lib().rcppml_hip_solve_irls_nb(self._h, C.c_int(dt), _dptr(col_ptr), _dptr(row_idx), _dptr(values),
                                            C.c_int64(ncols), _dptr(F), _dptr(G_base), _dptr(X), C.c_int(k), C.c_double(l1),
                                            C.c_double(l2), C.c_int(nonneg), C.c_int(cd_maxit), C.c_int(irls_max_iter),
                                            C.c_double(irls_tol), _dptr(theta_row), _dptr(theta_col)), "solve_irls_nb")

    def solve_irls(self, dt, loss_type, col_ptr, row_idx, values, ncols, F, G_base, X, k, l1=0.0, l2=0.0, nonneg=1, cd_maxit=100,
                   irls_max_iter=5, irls_tol=1e-4, theta_row=None, theta_col=None, loss_param=0.0, robust_delta=0.0):
        _chk(lib().rcppml_hip_solve_irls(self._h, C.c_int(dt), C.c_int(loss_type), _dptr(col_ptr), _dptr(row_idx), _dptr(values),
                                         C.c_int64(ncols), _dptr(F), _dptr(G_base), _dptr(X), C.c_int(k), C.c_double(l1),
                                         C.c_double(l2), C.c_int(nonneg), C.c_int(cd_maxit), C.c_int(irls_max_iter),
                                         C.c_double(irls_tol), _dptr(theta_row), _dptr(theta_col), C.c_double(loss_param),
                                         C.c_double(robust_delta)), "solve_irls")

    def irls_loss(self, dt, loss_type, col_ptr, row_idx, values, ncols, W_T, d, H, theta_row, k, out, loss_param=0.0, robust_delta=0.0):
        _chk(lib().rcppml_hip_irls_loss(self._h, C.c_int(dt), C.c_int(loss_type), _dptr(col_ptr), _dptr(row_idx), _dptr(values),
                                        C.c_int64(ncols), _dptr(W_T), _dptr(d), _dptr(H), _dptr(theta_row), C.c_int(k),
                                        C.c_double(loss_param), C.c_double(robust_delta), _dptr(out)), "irls_loss")

    def solve_cv(self, dt, col_ptr, row_idx, values, ncols, nrows, F, G, X, k, frac, cv_seed, mask_zeros=0, transposed=0, l1=0.0,
                 nonneg=1, cd_maxit=100, solver_mode=0):
        _chk(lib().rcppml_hip_solve_cv(self._h, C.c_int(dt), _dptr(col_ptr), _dptr(row_idx), _dptr(values), C.c_int64(ncols),
                                       C.c_int(nrows), _dptr(F), _dptr(G), _dptr(X), C.c_int(k), C.c_double(frac),
                                       C.c_ulonglong(cv_seed), C.c_int(mask_zeros), C.c_int(transposed), C.c_double(l1),
                                       C.c_int(nonneg), C.c_int(cd_maxit), C.c_int(solver_mode)), "solve_cv")

    def cv_test_error(self, dt, col_ptr, row_idx, values, ncols, nrows, W_T, d, H, k, frac, cv_seed, mask_zeros, out2):
        _chk(lib().rcppml_hip_cv_test_error(self._h, C.c_int(dt), _dptr(col_ptr), _dptr(row_idx), _dptr(values), C.c_int64(ncols),
                                            C.c_int(nrows), _dptr(W_T), _dptr(d), _dptr(H), C.c_int(k), C.c_double(frac),
                                            C.c_ulonglong(cv_seed), C.c_int(mask_zeros), _dptr(out2)), "cv_test_error")

    def solve_cv_irls(self, dt, loss_type, col_ptr, row_idx, values, ncols, nrows, F, G_add, X, k, frac, cv_seed, mask_zeros=0, transposed=0,
                      l1=0.0, nonneg=1, cd_maxit=100, solver_mode=0, irls_max_iter=5, irls_tol=1e-4, loss_param=1.5, robust_delta=0.0):
        _chk(lib().rcppml_hip_solve_cv_irls(self._h, C.c_int(dt), C.c_int(loss_type), _dptr(col_ptr), _dptr(row_idx), _dptr(values),
                                            C.c_int64(ncols), C.c_int(nrows), _dptr(F), _dptr(G_add) if G_add is not None else None,
                                            _dptr(X), C.c_int(k), C.c_double(frac), C.c_ulonglong(cv_seed), C.c_int(mask_zeros),
                                            C.c_int(transposed), C.c_double(l1), C.c_int(nonneg), C.c_int(cd_maxit), C.c_int(solver_mode),
                                            C.c_int(irls_max_iter), C.c_double(irls_tol), C.c_double(loss_param), C.c_double(robust_delta)),
             "solve_cv_irls")

    def cv_irls_loss(self, dt, loss_type, col_ptr, row_idx, values, ncols, nrows, W_T, d, H, theta_row, k, frac, cv_seed, mask_zeros,
                     loss_param, out4):
        _chk(lib().rcppml_hip_cv_irls_loss(self._h, C.c_int(dt), C.c_int(loss_type), _dptr(col_ptr), _dptr(row_idx), _dptr(values),
                                           C.c_int64(ncols), C.c_int(nrows), _dptr(W_T), _dptr(d), _dptr(H),
                                           _dptr(theta_row) if theta_row is not None else None, C.c_int(k), C.c_double(frac),
                                           C.c_ulonglong(cv_seed), C.c_int(mask_zeros), C.c_double(loss_param), _dptr(out4)), "cv_irls_loss")

    def cv_gp_theta_update(self, dt, mode, t_col_ptr, t_row_idx, t_values, m, nnz, W_T, d, H, n, k, frac, cv_seed, theta_max, theta):
        _chk(lib().rcppml_hip_cv_gp_theta_update(self._h, C.c_int(dt), C.c_int(mode), _dptr(t_col_ptr), _dptr(t_row_idx), _dptr(t_values),
                                                 C.c_int64(m), C.c_int64(nnz), _dptr(W_T), _dptr(d), _dptr(H), C.c_int64(n), C.c_int(k),
                                                 C.c_double(frac), C.c_ulonglong(cv_seed), C.c_double(theta_max), _dptr(theta)),
             "cv_gp_theta_update")

    def apply_graph_reg(self, dt, G, lap_p, lap_i, lap_x, X, k, ncols, lam):
        _chk(lib().rcppml_hip_apply_graph_reg(self._h, C.c_int(dt), _dptr(G), _dptr(lap_p), _dptr(lap_i), _dptr(lap_x), _dptr(X),
                                              C.c_int(k), C.c_int64(ncols), C.c_double(lam)), "apply_graph_reg")

    def apply_l21(self, dt, G, X, k, ncols, lam):
        _chk(lib().rcppml_hip_apply_l21(self._h, C.c_int(dt), _dptr(G), _dptr(X), C.c_int(k), C.c_int64(ncols), C.c_double(lam)), "apply_l21")

    def angular_posthoc(self, dt, X, k, ncols, lam):
        _chk(lib().rcppml_hip_angular_posthoc(self._h, C.c_int(dt), _dptr(X), C.c_int(k), C.c_int64(ncols), C.c_double(lam)), "angular_posthoc")

    def dispersion_update(self, dt, loss_type, mode, t_col_ptr, t_row_idx, t_values, m, nnz, W_T, d, H, n, k, power, lo, hi, theta):
        _chk(lib().rcppml_hip_dispersion_update(self._h, C.c_int(dt), C.c_int(loss_type), C.c_int(mode), _dptr(t_col_ptr),
                                                _dptr(t_row_idx), _dptr(t_values), C.c_int64(m), C.c_int64(nnz), _dptr(W_T), _dptr(d),
                                                _dptr(H), C.c_int64(n), C.c_int(k), C.c_double(power), C.c_double(lo),
                                                C.c_double(hi), _dptr(theta)), "dispersion_update")

    def vec_global(self, dt, stat, x, m):
        _chk(lib().rcppml_hip_vec_global(self._h, C.c_int(dt), C.c_int(stat), _dptr(x), C.c_int64(m)), "vec_global")

    def rhs_dense(self, dt, A, m, n, transposed, F, k, B):
        _chk(lib().rcppml_hip_rhs_dense(self._h, C.c_int(dt), _dptr(A), C.c_int64(m), C.c_int64(n), C.c_int(transposed), _dptr(F),
                                        C.c_int(k), _dptr(B)), "rhs_dense")

    def spz_decode(self, file_bytes, d_col_ptr, d_row_idx, d_values):
        """file_bytes: uint8 numpy array (host); outputs: device int32 (n+1), int32 (nnz), float64 (nnz)."""
        buf = np.ascontiguousarray(file_bytes, np.uint8)
        st = lib().rcppml_hip_spz_decode(self._h, buf.ctypes.data_as(C.c_void_p), C.c_uint64(buf.size), _dptr(d_col_ptr),
                                         _dptr(d_row_idx), _dptr(d_values))
        if st != 0:
            raise BackendError("spz_decode failed (status %d): %s" % (st, last_error()))

    def nb_size_update(self, dt, t_col_ptr, t_row_idx, t_values, m, W_T, d, H, n, k, r_min, r_max, nb_size):
        _chk(lib().rcppml_hip_nb_size_update(self._h, C.c_int(dt), _dptr(t_col_ptr), _dptr(t_row_idx), _dptr(t_values),
                                             C.c_int64(m), _dptr(W_T), _dptr(d), _dptr(H), C.c_int64(n), C.c_int(k),
                                             C.c_double(r_min), C.c_double(r_max), _dptr(nb_size)), "nb_size_update")

    def nb_size_update_loss(self, dt, t_col_ptr, t_row_idx, t_values, m, nnz, W_T, d, H, n, k, r_min, r_max, nb_size, out):
        """nb_size_update followed by nb_loss with the updated sizes, in one pass over CSC(A^T) (per-row dispersion)."""
        _chk(lib().rcppml_hip_nb_size_update_loss(self._h, C.c_int(dt), _dptr(t_col_ptr), _dptr(t_row_idx), _dptr(t_values),
                                                  C.c_int64(m), C.c_int64(nnz), _dptr(W_T), _dptr(d), _dptr(H), C.c_int64(n),
                                                  C.c_int(k), C.c_double(r_min), C.c_double(r_max), _dptr(nb_size), _dptr(out)),
             "nb_size_update_loss")

    def nb_loss(self, dt, col_ptr, row_idx, values, ncols, W_T, d, H, theta_row, k, out):
        _chk(lib().rcppml_hip_nb_loss(self._h, C.c_int(dt), _dptr(col_ptr), _dptr(row_idx), _dptr(values), C.c_int64(ncols),
                                      _dptr(W_T), _dptr(d), _dptr(H), _dptr(theta_row), C.c_int(k), _dptr(out)), "nb_loss")


def spz_info(file_bytes):
    """Header of a .spz v2 byte stream: (status, m, n, nnz, value_type); host only."""
    buf = np.ascontiguousarray(file_bytes, np.uint8)
    m, n, vt, nnz = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int64(0)
    st = lib().rcppml_hip_spz_info(buf.ctypes.data_as(C.c_void_p), C.c_uint64(buf.size), C.byref(m), C.byref(n), C.byref(nnz), C.byref(vt))
    return st, m.value, n.value, nnz.value, vt.value


def sp_read_gpu(path, device=0):
    """reference R/sp_gpu.R sp_read_gpu -> rcppml_sp_read_gpu: dict(status, m, n, nnz, col_ptr, row_idx, values) with the
    three device addresses as floats, exactly as the .C() call returns them."""
    pth = C.c_char_p(os.fsencode(path))
    dev = C.c_int(device)
    a, b, c = C.c_double(0), C.c_double(0), C.c_double(0)
    m, n, st = C.c_int(0), C.c_int(0), C.c_int(-99)
    nnz = C.c_double(0)
    lib().rcppml_sp_read_gpu(C.byref(pth), C.byref(dev), C.byref(a), C.byref(b), C.byref(c), C.byref(m), C.byref(n), C.byref(nnz), C.byref(st))
    return dict(status=st.value, m=m.value, n=n.value, nnz=int(nnz.value), col_ptr=a.value, row_idx=b.value, values=c.value,
                error=last_error() if st.value != 0 else "")


def copy_from_device_address(dst, addr, nbytes):
    """Copy nbytes from a raw device address (as sp_read_gpu returns them: a float) into a torch CUDA tensor."""
    hip = C.CDLL("libamdhip64.so")
    rc = hip.hipMemcpy(C.c_void_p(dst.data_ptr()), C.c_void_p(int(addr)), C.c_size_t(int(nbytes)), C.c_int(3))   # device to device
    if rc != 0:
        raise BackendError("hipMemcpy failed: %d" % rc)


def sp_free_gpu(h):
    a, b, c, st = C.c_double(h["col_ptr"]), C.c_double(h["row_idx"]), C.c_double(h["values"]), C.c_int(-99)
    lib().rcppml_sp_free_gpu(C.byref(a), C.byref(b), C.byref(c), C.byref(st))
    h["col_ptr"], h["row_idx"], h["values"] = a.value, b.value, c.value
    return st.value


def nmf_dense(A, k, W_T, H, *, entry="float", max_iter=100, tol=1e-4, L1_H=0.0, L1_W=0.0, L2_H=0.0, L2_W=0.0, L21_H=0.0,
              L21_W=0.0, ortho_H=0.0, ortho_W=0.0, ub_H=0.0, ub_W=0.0, cd_maxit=100, verbose=0, seed=0, patience=5, nonneg_W=1,
              nonneg_H=1, loss_type=0, norm_type=0, projective=0, symmetric=0, solver_mode=0, robust_delta=0.0, irls_max_iter=5,
              irls_tol=1e-4, dispersion_mode=2, gp_theta_init=0.1, gp_theta_max=5.0, nb_size_init=10.0, nb_size_max=1e6,
              nb_size_min=0.01, tweedie_power=1.5):
    """Call the dense plugin entry as reference gpu/bridge_nmf.hpp:537-690 does.  A: (m, n) float64 array (any layout: it
    is handed over column-major).  W_T (m, k), H (n, k) float64, updated IN PLACE.  entry: "float" | "double"."""
    L = lib()
    A = np.asfortranarray(A, dtype=np.float64)
    m, n = A.shape
    assert W_T.dtype == np.float64 and H.dtype == np.float64 and W_T.flags.c_contiguous and H.flags.c_contiguous
    assert W_T.shape == (m, k) and H.shape == (n, k)
    d = np.ones(k, np.float64)
    theta = np.zeros(max(m, n, 1), np.float64)          # gpu/bridge_nmf.hpp:622 theta_buf(max(m, n))
    # (the build-defined entries read *out_theta_len on input as the capacity of out_theta)
    out_iter, out_conv, out_status, out_theta_len = C.c_int(0), C.c_int(0), C.c_int(-99), C.c_int((theta.shape[0] if theta_capacity is None else int(theta_capacity)) if entry == "ex" else 0)
    out_loss, out_tol = C.c_double(0), C.c_double(0)
    args = [
        A.ctypes.data_as(C.POINTER(C.c_double)), _ci(m), _ci(n), _ci(k), _np_ptr(W_T), _np_ptr(H), _np_ptr(d), _ci(max_iter), _cd(tol),
        _cd(L1_H), _cd(L1_W), _cd(L2_H), _cd(L2_W), _cd(L21_H), _cd(L21_W), _cd(ortho_H), _cd(ortho_W), _cd(ub_H), _cd(ub_W),
        _ci(cd_maxit), _ci(verbose), _ci(seed), _ci(1), _ci(patience), _ci(nonneg_W), _ci(nonneg_H), _ci(loss_type), _cd(1.0),
        _ci(irls_max_iter), _cd(irls_tol), _ci(norm_type), _ci(dispersion_mode), _cd(gp_theta_init), _cd(gp_theta_max), _cd(0.0),
        _cd(nb_size_init), _cd(nb_size_max), _cd(nb_size_min),
        _cd(robust_delta), _cd(tweedie_power), _ci(projective), _ci(symmetric), _ci(solver_mode), _np_ptr(theta), C.byref(out_theta_len),
        C.byref(out_iter), C.byref(out_conv), C.byref(out_loss), C.byref(out_status), C.byref(out_tol),
    ]
    assert len(args) == 50
    getattr(L, "rcppml_gpu_nmf_dense_unified_" + entry)(*args)
    return dict(d=d, iter=out_iter.value, converged=bool(out_conv.value), loss=out_loss.value, tol=out_tol.value,
                theta=theta[:out_theta_len.value].copy(), status=out_status.value, error=last_error() if out_status.value != 0 else "")
