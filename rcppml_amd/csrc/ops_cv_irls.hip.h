// ops_cv_irls.hip.h -- (included at the end of ops_irls.hip: the IRLS kernels it builds on are defined once, in that translation unit)
// cross-validation with IRLS losses (device-level C ABI, include/rcppml_gpu.h layer 2): the per-column
// weighted-Gram half-update, the per-element train / test losses and the GP theta update over the training entries.
// Kernels and the reference lines they follow: kernels_cv_irls.hip.h.
#pragma once
#include "kernels_cv_irls.hip.h"

namespace {
// speckled_cv.hpp:57-68 (as in ops_cv.hip); holdout_fraction <= 0: no entry is held out (threshold 0)
void cvi_mask_params(double holdout_fraction, unsigned long long cv_seed, unsigned long long* seed, unsigned long long* threshold) {
    const unsigned s32 = static_cast<unsigned>(cv_seed);
    *seed = s32 == 0 ? 12345ULL : static_cast<unsigned long long>(s32);
    if (!(holdout_fraction > 0.0)) { *threshold = 0; return; }
    if (!(holdout_fraction < 1.0)) throw std::runtime_error("cv: holdout_fraction must be in [0, 1)");
    const unsigned long long inv_prob = static_cast<unsigned long long>(1.0 / holdout_fraction);
    if (inv_prob == 0) throw std::runtime_error("cv: holdout_fraction too large");
    *threshold = 0xFFFFFFFFFFFFFFFFULL / inv_prob;
}
bool loss_ok(int loss_type, double robust) { return (loss_type >= 4 && loss_type <= 8) || (loss_type == 0 && robust > 0); }

template <class T>
void solve_impl(rcppml_hip_ctx* c, int loss_type, const int* cp, const int* ri, const T* vals, int64_t ncols, int nrows, const T* F,
                const T* Gadd, T* X, int k, double frac, unsigned long long cv_seed, int mask_zeros, int transposed, T l1, int nonneg,
                int maxit, int solver_mode, int irls_max_iter, T irls_tol, T power, T robust) {
    if (ncols <= 0) return;
    unsigned long long seed, thr;
    cvi_mask_params(frac, cv_seed, &seed, &thr);
    const int64_t nblk = (ncols + 3) / 4;
    const int* mp = c->cv_mask_p[transposed ? 1 : 0];          // user mask of the fit, if one is set (rcppml_hip_ctx_set_cv_mask)
    const int* mi = c->cv_mask_i[transposed ? 1 : 0];
    if (k > 64) {          // one wave per column, two features per lane, Gram tile in LDS (kernels_wide.hip.h)
        auto kern = wide_cv_irls_solve_kernel<T>;
        static DynSmemOnce once;
        once.ensure(reinterpret_cast<const void*>(kern), wide_smem_bytes<T>(), c->device);
        hipLaunchKernelGGL(kern, dim3((unsigned)ncols), dim3(64), wide_smem_bytes<T>(), c->stream, cp, ri, vals, ncols, nrows, F, Gadd, X, k,
                           seed, thr, mask_zeros, transposed, l1, nonneg, maxit, solver_mode, loss_type, irls_max_iter, irls_tol, power, robust, mp, mi);
        HIPCHK(hipGetLastError());
        return;
    }
    if (k <= 32) {
        hipLaunchKernelGGL((cv_irls_solve_kernel<T, 32>), dim3((unsigned)nblk), dim3(256), (size_t)4 * 32 * 32 * sizeof(T), c->stream, cp, ri, vals,
                           ncols, nrows, F, Gadd, X, k, seed, thr, mask_zeros, transposed, l1, nonneg, maxit, solver_mode, loss_type,
                           irls_max_iter, irls_tol, power, robust, mp, mi);
    } else {
        const size_t smem = (size_t)4 * 64 * 64 * sizeof(T);
        auto kern = cv_irls_solve_kernel<T, 64>;
        static DynSmemOnce once;
        once.ensure(reinterpret_cast<const void*>(kern), smem, c->device);
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), smem, c->stream, cp, ri, vals, ncols, nrows, F, Gadd, X, k, seed, thr,
                           mask_zeros, transposed, l1, nonneg, maxit, solver_mode, loss_type, irls_max_iter, irls_tol, power, robust, mp, mi);
    }
    HIPCHK(hipGetLastError());
}

template <class T>
void loss_impl(rcppml_hip_ctx* c, int loss_type, const int* cp, const int* ri, const T* vals, int64_t ncols, int nrows, const T* W_T,
               const T* d, const T* H, const T* theta, int k, double frac, unsigned long long cv_seed, int mask_zeros, double power,
               double* out4) {
    unsigned long long seed, thr;
    cvi_mask_params(frac, cv_seed, &seed, &thr);
    const int64_t nblk = ncols > 0 ? (ncols + 3) / 4 : 1;
    char* buf = static_cast<char*>(c->scratch(WS_RED2, (size_t)nblk * 32));
    double* ps = reinterpret_cast<double*>(buf);
    unsigned long long* pn = reinterpret_cast<unsigned long long*>(buf + (size_t)nblk * 16);
    hipLaunchKernelGGL(cv_irls_loss_kernel<T>, dim3((unsigned)nblk), dim3(256), 0, c->stream, cp, ri, vals, ncols, nrows, W_T, d, H, theta, k,
                       seed, thr, mask_zeros, loss_type, power, ps, pn, c->cv_mask_p[0], c->cv_mask_i[0]);        // (the loss walks A itself)
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(cv_irls_loss_final_kernel, dim3(1), dim3(256), 0, c->stream, ps, pn, (int)nblk, out4);
    HIPCHK(hipGetLastError());
}

template <class T>
void theta_impl(rcppml_hip_ctx* c, int dtype, int mode, const int* tp, const int* ti, const T* tx, int64_t m, int64_t nnz, const T* W_T,
                const T* d, const T* H, int64_t n, int k, double frac, unsigned long long cv_seed, double hi, T* theta) {
    if (m <= 0) return;
    unsigned long long seed, thr;
    cvi_mask_params(frac, cv_seed, &seed, &thr);
    const size_t head = ((size_t)k * sizeof(T) + 255) / 256 * 256;
    char* buf = static_cast<char*>(c->scratch(WS_IRLS, head + (size_t)std::max<int64_t>(nnz, 1) * sizeof(T)));
    T* h_rs = reinterpret_cast<T*>(buf);
    T* s_cache = reinterpret_cast<T*>(buf + head);
    if (rcppml_hip_row_norms(c, dtype, H, k, n, 3, h_rs) != 0) throw std::runtime_error(rcppml_err());
    hipLaunchKernelGGL(cv_gp_theta_rows_kernel<T>, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, c->stream, tp, ti, tx, m, n, W_T, d, H, h_rs, k,
                       seed, thr, hi, s_cache, theta);
    HIPCHK(hipGetLastError());
    if (mode == 1 && rcppml_hip_vec_global(c, dtype, 0, theta, m) != 0) throw std::runtime_error(rcppml_err());     // GLOBAL: the mean
}
}  // namespace

extern "C" int rcppml_hip_solve_cv_irls(rcppml_hip_ctx* c, int dtype, int loss_type, const int* col_ptr, const int* row_idx,
                                        const void* values, int64_t ncols, int nrows, const void* F, const void* G_add, void* X, int k,
                                        double holdout_fraction, unsigned long long cv_seed, int mask_zeros, int transposed, double l1,
                                        int nonneg, int cd_maxit, int solver_mode, int irls_max_iter, double irls_tol,
                                        double loss_param, double robust_delta) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (k < 1 || k > 128) throw std::runtime_error("solve_cv_irls: k must be in [1,128]");
        if (!loss_ok(loss_type, robust_delta)) throw std::runtime_error("solve_cv_irls: loss_type must be in 4..8, or 0 with robust_delta > 0");
        if (solver_mode != 0 && solver_mode != 1) throw std::runtime_error("solve_cv_irls: solver_mode must be 0 (CD) or 1 (Cholesky+clip)");
        if (dtype == RCPPML_F32)
            solve_impl<float>(c, loss_type, col_ptr, row_idx, (const float*)values, ncols, nrows, (const float*)F, (const float*)G_add, (float*)X,
                              k, holdout_fraction, cv_seed, mask_zeros, transposed, (float)l1, nonneg, cd_maxit, solver_mode, irls_max_iter,
                              (float)irls_tol, (float)loss_param, (float)robust_delta);
        else
            solve_impl<double>(c, loss_type, col_ptr, row_idx, (const double*)values, ncols, nrows, (const double*)F, (const double*)G_add,
                               (double*)X, k, holdout_fraction, cv_seed, mask_zeros, transposed, l1, nonneg, cd_maxit, solver_mode,
                               irls_max_iter, irls_tol, loss_param, robust_delta);
        return 0;
    }
    RCPPML_CATCH_RET
}

extern "C" int rcppml_hip_cv_irls_loss(rcppml_hip_ctx* c, int dtype, int loss_type, const int* col_ptr, const int* row_idx,
                                       const void* values, int64_t ncols, int nrows, const void* W_T, const void* d, const void* H,
                                       const void* theta_row, int k, double holdout_fraction, unsigned long long cv_seed, int mask_zeros,
                                       double loss_param, double* out4) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (k < 1 || k > 128) throw std::runtime_error("cv_irls_loss: k must be in [1,128]");
        if (!(loss_type == 0 || (loss_type >= 4 && loss_type <= 8))) throw std::runtime_error("cv_irls_loss: loss_type must be 0 or in 4..8");
        if (dtype == RCPPML_F32)
            loss_impl<float>(c, loss_type, col_ptr, row_idx, (const float*)values, ncols, nrows, (const float*)W_T, (const float*)d,
                             (const float*)H, (const float*)theta_row, k, holdout_fraction, cv_seed, mask_zeros, loss_param, out4);
        else
            loss_impl<double>(c, loss_type, col_ptr, row_idx, (const double*)values, ncols, nrows, (const double*)W_T, (const double*)d,
                              (const double*)H, (const double*)theta_row, k, holdout_fraction, cv_seed, mask_zeros, loss_param, out4);
        return 0;
    }
    RCPPML_CATCH_RET
}

extern "C" int rcppml_hip_cv_gp_theta_update(rcppml_hip_ctx* c, int dtype, int mode, const int* t_col_ptr, const int* t_row_idx,
                                             const void* t_values, int64_t m, int64_t nnz, const void* W_T, const void* d, const void* H,
                                             int64_t n, int k, double holdout_fraction, unsigned long long cv_seed, double theta_max,
                                             void* theta) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (k < 1 || k > 128) throw std::runtime_error("cv_gp_theta_update: k must be in [1,128]");
        if (mode != 1 && mode != 2) throw std::runtime_error("cv_gp_theta_update: mode must be 1 (global) or 2 (per row)");
        if (dtype == RCPPML_F32)
            theta_impl<float>(c, dtype, mode, t_col_ptr, t_row_idx, (const float*)t_values, m, nnz, (const float*)W_T, (const float*)d,
                              (const float*)H, n, k, holdout_fraction, cv_seed, theta_max, (float*)theta);
        else
            theta_impl<double>(c, dtype, mode, t_col_ptr, t_row_idx, (const double*)t_values, m, nnz, (const double*)W_T, (const double*)d,
                               (const double*)H, n, k, holdout_fraction, cv_seed, theta_max, (double*)theta);
        return 0;
    }
    RCPPML_CATCH_RET
}
