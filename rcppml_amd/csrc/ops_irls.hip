// ops_irls.hip -- NB-IRLS half-update, NB size update and NB loss (device-level C ABI, include/rcppml_gpu.h layer 2)
#include <type_traits>
#include <cstring>
#include "common.hip.h"
#include "kernels_irls.hip.h"
#include "kernels_wide.hip.h"

using namespace rk;

template <class T>
static void irls_impl(rcppml_hip_ctx* c, const int* cp, const int* ri, const T* vals, int64_t ncols, const T* F,
                      const T* Gbase, T* X, int k, T l1, T l2, int nonneg, int cd_maxit, int irls_max_iter, T irls_tol,
                      const T* theta_row, const T* theta_col, int loss_type, T power, T robust) {
    if (ncols <= 0) return;
    if (!((loss_type >= 4 && loss_type <= 8) || (loss_type == 0 && robust > T(0))))
        throw std::runtime_error("solve_irls: loss_type must be 4 (GP), 5 (NB), 6 (Gamma), 7 (inverse Gaussian), 8 (Tweedie), or 0 (MSE) with robust_delta > 0");
    if (k < 1 || k > 128) throw std::runtime_error("solve_irls: k must be in [1,128]");
    const int64_t nblk = (ncols + 3) / 4;
    unsigned long long* const st = c->opt_cd_count ? c->stats + 4 : nullptr;     // [4] IRLS passes, [5] nonzero-passes
    if (k > 64) {          // one wave per column, two features per lane, Gram tile in LDS (kernels_wide.hip.h)
        auto kern = wide_irls_solve_kernel<T>;
        static DynSmemOnce once;
        once.ensure(reinterpret_cast<const void*>(kern), wide_smem_bytes<T>(), c->device);
        hipLaunchKernelGGL(kern, dim3((unsigned)ncols), dim3(64), wide_smem_bytes<T>(), c->stream, cp, ri, vals, ncols, F, Gbase, X, k,
                           l1, l2, nonneg, cd_maxit, irls_max_iter, irls_tol, theta_row, theta_col, loss_type, power, robust, st);
        HIPCHK(hipGetLastError());
        return;
    }
    if constexpr (std::is_same<T, float>::value) {
        // fp32, k <= 32: weighted Gram on the matrix cores (RCPPML_GPU_IRLS_VARIANT=valu keeps the register form)
        static int use_mfma = -1;
        if (use_mfma < 0) use_mfma = exp_flag("RCPPML_GPU_IRLS_VARIANT", "valu") ? 0 : 1;
        if (use_mfma && k <= 32 && k % 4 == 0 && reinterpret_cast<uintptr_t>(F) % 16 == 0 && reinterpret_cast<uintptr_t>(Gbase) % 16 == 0) {   // (16-byte loads of F rows and of the base Gram; unaligned views take the register kernel)
            // many short columns: four columns per wavefront in the CD solve (irls_nb_mfma32q_kernel); RCPPML_OPT_IRLS_COLUMNS_PER_WAVE forces
            const bool quad = c->opt_irls_cpw > 0 ? c->opt_irls_cpw == 4 : ncols >= (int64_t)64 * c->num_cu;
            if (quad) {
                const size_t qsmem = (size_t)4 * (64 * 36 + 2 * 64 + 2 * 32) * sizeof(float);
                const int64_t qblk = (ncols + 15) / 16;
                if (loss_type == 5 && !(robust > 0)) {
                    static DynSmemOnce once;
                    once.ensure(reinterpret_cast<const void*>(&irls_nb_mfma32q_kernel<5>), qsmem, c->device);
                    hipLaunchKernelGGL(irls_nb_mfma32q_kernel<5>, dim3((unsigned)qblk), dim3(256), qsmem, c->stream, cp, ri, vals, ncols, F,
                                       Gbase, X, k, l1, l2, nonneg, cd_maxit, irls_max_iter, irls_tol, theta_row, theta_col, loss_type, power, robust, st);
                } else {
                    static DynSmemOnce once;
                    once.ensure(reinterpret_cast<const void*>(&irls_nb_mfma32q_kernel<-1>), qsmem, c->device);
                    hipLaunchKernelGGL(irls_nb_mfma32q_kernel<-1>, dim3((unsigned)qblk), dim3(256), qsmem, c->stream, cp, ri, vals, ncols, F,
                                       Gbase, X, k, l1, l2, nonneg, cd_maxit, irls_max_iter, irls_tol, theta_row, theta_col, loss_type, power, robust, st);
                }
                HIPCHK(hipGetLastError());
                return;
            }
            const size_t smem = (size_t)4 * (32 * 36 + 2 * 32 + 32) * sizeof(float);
            if (loss_type == 5 && !(robust > 0))          // negative binomial without the robust modifier: specialised weights
                hipLaunchKernelGGL(irls_nb_mfma32_kernel<5>, dim3((unsigned)nblk), dim3(256), smem, c->stream, cp, ri, vals, ncols, F,
                                   Gbase, X, k, l1, l2, nonneg, cd_maxit, irls_max_iter, irls_tol, theta_row, theta_col, loss_type, power, robust, st);
            else
                hipLaunchKernelGGL(irls_nb_mfma32_kernel<-1>, dim3((unsigned)nblk), dim3(256), smem, c->stream, cp, ri, vals, ncols, F,
                                   Gbase, X, k, l1, l2, nonneg, cd_maxit, irls_max_iter, irls_tol, theta_row, theta_col, loss_type, power, robust, st);
            HIPCHK(hipGetLastError());
            return;
        }
        if (use_mfma && k <= 64 && k % 4 == 0 && reinterpret_cast<uintptr_t>(F) % 16 == 0) {      // 32 < k <= 64: 2 x 2 tiles
            const size_t smem = (size_t)4 * (64 * 64 + 2 * 32 + 64) * sizeof(float);
            static DynSmemOnce once;
            once.ensure(reinterpret_cast<const void*>(&irls_nb_mfma32x2_kernel), smem, c->device);
            hipLaunchKernelGGL(irls_nb_mfma32x2_kernel, dim3((unsigned)nblk), dim3(256), smem, c->stream, cp, ri, vals, ncols, F,
                               Gbase, X, k, l1, l2, nonneg, cd_maxit, irls_max_iter, irls_tol, theta_row, theta_col, loss_type, power, robust, st);
            HIPCHK(hipGetLastError());
            return;
        }
    }
    if constexpr (std::is_same<T, double>::value) {
        static int use_mfma64 = -1;
        if (use_mfma64 < 0) use_mfma64 = exp_flag("RCPPML_GPU_IRLS_VARIANT", "valu") ? 0 : 1;
        if (use_mfma64 && k <= 32 && k % 2 == 0 && reinterpret_cast<uintptr_t>(F) % 16 == 0) {
            const size_t smem = (size_t)4 * (32 * 34 + 2 * 32 + 32) * sizeof(double);
            hipLaunchKernelGGL(irls_nb_mfma64_kernel, dim3((unsigned)nblk), dim3(256), smem, c->stream, cp, ri, vals, ncols, F,
                               Gbase, X, k, l1, l2, nonneg, cd_maxit, irls_max_iter, irls_tol, theta_row, theta_col, loss_type, power, robust, st);
            HIPCHK(hipGetLastError());
            return;
        }
    }
    if (k <= 32) {      // 32-wide instantiation: half the rank-1 work per nonzero, 16 KB of LDS per block (8 waves per SIMD)
        const size_t smem = (size_t)4 * 32 * 32 * sizeof(T);
        hipLaunchKernelGGL((irls_nb_solve_kernel<T, 32>), dim3((unsigned)nblk), dim3(256), smem, c->stream, cp, ri, vals, ncols,
                           F, Gbase, X, k, l1, l2, nonneg, cd_maxit, irls_max_iter, irls_tol, theta_row, theta_col, loss_type, power, robust, st);
    } else {
        const size_t smem = (size_t)4 * 64 * 64 * sizeof(T);
        auto kern = irls_nb_solve_kernel<T, 64>;
        static DynSmemOnce once;
        once.ensure(reinterpret_cast<const void*>(kern), smem, c->device);
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), smem, c->stream, cp, ri, vals, ncols, F, Gbase, X, k, l1, l2,
                           nonneg, cd_maxit, irls_max_iter, irls_tol, theta_row, theta_col, loss_type, power, robust, st);
    }
    HIPCHK(hipGetLastError());
}
extern "C" int rcppml_hip_solve_irls(rcppml_hip_ctx* c, int dtype, int loss_type, const int* col_ptr, const int* row_idx,
                                     const void* values, int64_t ncols, const void* F, const void* G_base, void* X,
                                     int k, double l1, double l2, int nonneg, int cd_maxit, int irls_max_iter,
                                     double irls_tol, const void* theta_row, const void* theta_col, double loss_param,
                                     double robust_delta) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (dtype == RCPPML_F32)
            irls_impl<float>(c, col_ptr, row_idx, (const float*)values, ncols, (const float*)F, (const float*)G_base,
                             (float*)X, k, (float)l1, (float)l2, nonneg, cd_maxit, irls_max_iter, (float)irls_tol,
                             (const float*)theta_row, (const float*)theta_col, loss_type, (float)loss_param, (float)robust_delta);
        else
            irls_impl<double>(c, col_ptr, row_idx, (const double*)values, ncols, (const double*)F, (const double*)G_base,
                              (double*)X, k, l1, l2, nonneg, cd_maxit, irls_max_iter, irls_tol, (const double*)theta_row,
                              (const double*)theta_col, loss_type, loss_param, robust_delta);
        return 0;
    }
    RCPPML_CATCH_RET
}
extern "C" int rcppml_hip_solve_irls_nb(rcppml_hip_ctx* c, int dtype, const int* col_ptr, const int* row_idx,
                                        const void* values, int64_t ncols, const void* F, const void* G_base, void* X,
                                        int k, double l1, double l2, int nonneg, int cd_maxit, int irls_max_iter,
                                        double irls_tol, const void* theta_row, const void* theta_col) {
    return rcppml_hip_solve_irls(c, dtype, 5, col_ptr, row_idx, values, ncols, F, G_base, X, k, l1, l2, nonneg, cd_maxit,
                                 irls_max_iter, irls_tol, theta_row, theta_col, 0.0, 0.0);
}

template <class T>
static void nb_size_impl(rcppml_hip_ctx* c, int dtype, const int* tp, const int* ti, const T* tx, int64_t m, const T* W_T,
                         const T* d, const T* H, int64_t n, int k, double r_min, double r_max, T* nb_size) {
    if (k < 1 || k > 128) throw std::runtime_error("nb_size_update: k must be in [1,128]");
    T* tmp = static_cast<T*>(c->scratch(WS_IRLS, ((size_t)k * k + k) * sizeof(T)));
    T* G_H = tmp;
    T* h_rs = tmp + (size_t)k * k;
    if (rcppml_hip_gram(c, dtype, H, k, n, 1e-15, 0.0, G_H) != 0) throw std::runtime_error(rcppml_err());
    if (rcppml_hip_row_norms(c, dtype, H, k, n, 3, h_rs) != 0) throw std::runtime_error(rcppml_err());
    const int64_t nblk = (m + 3) / 4;
    hipLaunchKernelGGL(nb_size_rows_kernel<T>, dim3((unsigned)nblk), dim3(256), 0, c->stream, tp, ti, tx, m, W_T, d, H, h_rs,
                       G_H, k, r_min, r_max, nb_size);
    HIPCHK(hipGetLastError());
}
extern "C" int rcppml_hip_nb_size_update(rcppml_hip_ctx* c, int dtype, const int* t_col_ptr, const int* t_row_idx,
                                         const void* t_values, int64_t m, const void* W_T, const void* d, const void* H,
                                         int64_t n, int k, double r_min, double r_max, void* nb_size) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (dtype == RCPPML_F32)
            nb_size_impl<float>(c, dtype, t_col_ptr, t_row_idx, (const float*)t_values, m, (const float*)W_T, (const float*)d,
                                (const float*)H, n, k, r_min, r_max, (float*)nb_size);
        else
            nb_size_impl<double>(c, dtype, t_col_ptr, t_row_idx, (const double*)t_values, m, (const double*)W_T,
                                 (const double*)d, (const double*)H, n, k, r_min, r_max, (double*)nb_size);
        return 0;
    }
    RCPPML_CATCH_RET
}

template <class T>
static void nb_size_loss_impl(rcppml_hip_ctx* c, int dtype, const int* tp, const int* ti, const T* tx, int64_t m, int64_t nnz,
                              const T* W_T, const T* d, const T* H, int64_t n, int k, double r_min, double r_max, T* nb_size,
                              double* out) {
    if (k < 1 || k > 128) throw std::runtime_error("nb_size_update_loss: k must be in [1,128]");
    const size_t head = (((size_t)k * k + k) * sizeof(T) + 255) / 256 * 256;
    char* buf = static_cast<char*>(c->scratch(WS_IRLS, head + (size_t)(nnz > 0 ? nnz : 1) * sizeof(T)));
    T* G_H = reinterpret_cast<T*>(buf);
    T* h_rs = G_H + (size_t)k * k;
    T* mu_cache = reinterpret_cast<T*>(buf + head);
    if (rcppml_hip_gram(c, dtype, H, k, n, 1e-15, 0.0, G_H) != 0) throw std::runtime_error(rcppml_err());
    if (rcppml_hip_row_norms(c, dtype, H, k, n, 3, h_rs) != 0) throw std::runtime_error(rcppml_err());
    const int64_t nblk = m > 0 ? (m + 3) / 4 : 1;
    double* partial = static_cast<double*>(c->scratch(WS_RED2, (size_t)nblk * sizeof(double)));
    hipLaunchKernelGGL(nb_size_loss_rows_kernel<T>, dim3((unsigned)nblk), dim3(256), 0, c->stream, tp, ti, tx, m, W_T, d, H, h_rs,
                       G_H, k, r_min, r_max, nb_size, mu_cache, partial);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(sum_partials, dim3(1), dim3(256), 0, c->stream, partial, (int)nblk, out);
    HIPCHK(hipGetLastError());
}
extern "C" int rcppml_hip_nb_size_update_loss(rcppml_hip_ctx* c, int dtype, const int* t_col_ptr, const int* t_row_idx,
                                              const void* t_values, int64_t m, int64_t nnz, const void* W_T, const void* d,
                                              const void* H, int64_t n, int k, double r_min, double r_max, void* nb_size,
                                              double* out) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (dtype == RCPPML_F32)
            nb_size_loss_impl<float>(c, dtype, t_col_ptr, t_row_idx, (const float*)t_values, m, nnz, (const float*)W_T,
                                     (const float*)d, (const float*)H, n, k, r_min, r_max, (float*)nb_size, out);
        else
            nb_size_loss_impl<double>(c, dtype, t_col_ptr, t_row_idx, (const double*)t_values, m, nnz, (const double*)W_T,
                                      (const double*)d, (const double*)H, n, k, r_min, r_max, (double*)nb_size, out);
        return 0;
    }
    RCPPML_CATCH_RET
}

template <class T>
static void vec_global_impl(rcppml_hip_ctx* c, int stat, T* x, int64_t m) {
    if (m <= 0) return;
    hipLaunchKernelGGL(vec_global_fill_kernel<T>, dim3(1), dim3(256), 0, c->stream, x, m, stat);
    HIPCHK(hipGetLastError());
}
extern "C" int rcppml_hip_vec_global(rcppml_hip_ctx* c, int dtype, int stat, void* x, int64_t m) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (stat != 0 && stat != 1) throw std::runtime_error("vec_global: stat must be 0 (mean) or 1 (median = sorted[m/2])");
        if (dtype == RCPPML_F32) vec_global_impl<float>(c, stat, (float*)x, m);
        else vec_global_impl<double>(c, stat, (double*)x, m);
        return 0;
    }
    RCPPML_CATCH_RET
}

template <class T>
static void dispersion_impl(rcppml_hip_ctx* c, int dtype, int loss_type, int mode, const int* tp, const int* ti, const T* tx,
                            int64_t m, int64_t nnz, const T* W_T, const T* d, const T* H, int64_t n, int k, double power,
                            double lo, double hi, T* theta) {
    if (k < 1 || k > 128) throw std::runtime_error("dispersion_update: k must be in [1,128]");
    if (!(loss_type == 4 || (loss_type >= 6 && loss_type <= 8))) throw std::runtime_error("dispersion_update: loss_type must be 4 (GP) or 6 / 7 / 8 (Gamma / inverse Gaussian / Tweedie)");
    if (mode != 1 && mode != 2) throw std::runtime_error("dispersion_update: mode must be 1 (global) or 2 (per row)");
    if (m <= 0) return;
    const size_t head = ((size_t)k * sizeof(T) + 255) / 256 * 256;
    char* buf = static_cast<char*>(c->scratch(WS_IRLS, head + (loss_type == 4 ? (size_t)nnz * sizeof(T) : 0)));
    T* h_rs = reinterpret_cast<T*>(buf);
    T* s_cache = reinterpret_cast<T*>(buf + head);
    if (loss_type == 4 && rcppml_hip_row_norms(c, dtype, H, k, n, 3, h_rs) != 0) throw std::runtime_error(rcppml_err());
    hipLaunchKernelGGL(dispersion_rows_kernel<T>, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, c->stream, tp, ti, tx, m, W_T, d, H,
                       h_rs, k, loss_type, power, lo, hi, s_cache, theta);
    HIPCHK(hipGetLastError());
    if (mode == 1) vec_global_impl<T>(c, loss_type == 4 ? 0 : 1, theta, m);
}
extern "C" int rcppml_hip_dispersion_update(rcppml_hip_ctx* c, int dtype, int loss_type, int mode, const int* t_col_ptr,
                                            const int* t_row_idx, const void* t_values, int64_t m, int64_t nnz, const void* W_T,
                                            const void* d, const void* H, int64_t n, int k, double power, double lo, double hi,
                                            void* theta) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (dtype == RCPPML_F32)
            dispersion_impl<float>(c, dtype, loss_type, mode, t_col_ptr, t_row_idx, (const float*)t_values, m, nnz, (const float*)W_T,
                                   (const float*)d, (const float*)H, n, k, power, lo, hi, (float*)theta);
        else
            dispersion_impl<double>(c, dtype, loss_type, mode, t_col_ptr, t_row_idx, (const double*)t_values, m, nnz,
                                    (const double*)W_T, (const double*)d, (const double*)H, n, k, power, lo, hi, (double*)theta);
        return 0;
    }
    RCPPML_CATCH_RET
}

template <class T>
static void nb_loss_impl(rcppml_hip_ctx* c, const int* cp, const int* ri, const T* vals, int64_t ncols, const T* W_T,
                         const T* d, const T* H, const T* theta_row, int k, double* out, int loss_type, double power, double robust) {
    const int64_t nblk = ncols > 0 ? (ncols + 3) / 4 : 1;
    double* partial = static_cast<double*>(c->scratch(WS_RED2, (size_t)nblk * sizeof(double)));
    constexpr int VEC = 16 / (int)sizeof(T);
    const int vec_ok = (k % VEC == 0 && reinterpret_cast<uintptr_t>(W_T) % 16 == 0) ? 1 : 0;
    hipLaunchKernelGGL(nb_loss_lane_kernel<T>, dim3((unsigned)nblk), dim3(256), 0, c->stream, cp, ri, vals, ncols, W_T, d, H,
                       theta_row, k, vec_ok, loss_type, power, robust, partial);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(sum_partials, dim3(1), dim3(256), 0, c->stream, partial, (int)nblk, out);
    HIPCHK(hipGetLastError());
}
extern "C" int rcppml_hip_irls_loss(rcppml_hip_ctx* c, int dtype, int loss_type, const int* col_ptr, const int* row_idx,
                                    const void* values, int64_t ncols, const void* W_T, const void* d, const void* H,
                                    const void* theta_row, int k, double loss_param, double robust_delta, double* out) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (k < 1 || k > 128) throw std::runtime_error("irls_loss: k must be in [1,128]");
        if (!((loss_type >= 4 && loss_type <= 8) || (loss_type == 0 && robust_delta > 0)))
            throw std::runtime_error("irls_loss: loss_type must be in 4..8, or 0 with robust_delta > 0");
        if (dtype == RCPPML_F32)
            nb_loss_impl<float>(c, col_ptr, row_idx, (const float*)values, ncols, (const float*)W_T, (const float*)d,
                                (const float*)H, (const float*)theta_row, k, out, loss_type, loss_param, robust_delta);
        else
            nb_loss_impl<double>(c, col_ptr, row_idx, (const double*)values, ncols, (const double*)W_T, (const double*)d,
                                 (const double*)H, (const double*)theta_row, k, out, loss_type, loss_param, robust_delta);
        return 0;
    }
    RCPPML_CATCH_RET
}
extern "C" int rcppml_hip_nb_loss(rcppml_hip_ctx* c, int dtype, const int* col_ptr, const int* row_idx, const void* values,
                                  int64_t ncols, const void* W_T, const void* d, const void* H, const void* theta_row, int k,
                                  double* out) {
    return rcppml_hip_irls_loss(c, dtype, 5, col_ptr, row_idx, values, ncols, W_T, d, H, theta_row, k, 0.0, 0.0, out);
}

#include "ops_cv_irls.hip.h"
