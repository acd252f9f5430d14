// ops_misc.hip -- scaling, loss reductions, explicit-mask solve (device-level C ABI)
#include "common.hip.h"
#include "kernels.hip.h"
#include "kernels_wide.hip.h"

using namespace rk;
// ----------------------------------------------------------------------------
// Scaling
// ----------------------------------------------------------------------------
template <class T>
static void row_norms_impl(rcppml_hip_ctx* c, const T* X, int k, int64_t ncols, int norm_type, T* out) {
    int64_t nblk = (ncols + 255) / 256;
    if (nblk > 2 * (int64_t)c->num_cu) nblk = 2 * c->num_cu;
    if (nblk < 1) nblk = 1;
    T* partial = static_cast<T*>(c->scratch(WS_RED, (size_t)nblk * k * sizeof(T)));
    constexpr int VEC = 16 / sizeof(T);
    if (k % VEC == 0 && k / VEC <= 256 && reinterpret_cast<uintptr_t>(X) % 16 == 0) {
        const int slots = 256 / (k / VEC);
        hipLaunchKernelGGL((row_norm_partial_vec<T, VEC>), dim3((unsigned)nblk), dim3(256), (size_t)slots * k * sizeof(T),
                           c->stream, X, k, ncols, norm_type, partial);
    } else {
        hipLaunchKernelGGL(row_norm_partial<T>, dim3((unsigned)nblk), dim3(256), 256 * sizeof(T), c->stream, X, k, ncols,
                           norm_type, partial);
    }
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(row_norm_final<T>, dim3(k), dim3(64), 0, c->stream, partial, (int)nblk, k, out);
    HIPCHK(hipGetLastError());
}
extern "C" int rcppml_hip_row_norms(rcppml_hip_ctx* c, int dtype, const void* X, int k, int64_t ncols,
                                    int norm_type, void* out) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (dtype == RCPPML_F32) row_norms_impl<float>(c, (const float*)X, k, ncols, norm_type, (float*)out);
        else row_norms_impl<double>(c, (const double*)X, k, ncols, norm_type, (double*)out);
        return 0;
    }
    RCPPML_CATCH_RET
}
template <class T>
static void apply_scaling_impl(rcppml_hip_ctx* c, T* X, int k, int64_t ncols, int norm_type, const T* sums, T* d) {
    const int64_t total = (int64_t)k * ncols;
    if (norm_type == 2 || total <= 0) {          // no scaling (d = 1), or only d wanted (the sharded loop's d from the global sums)
        hipLaunchKernelGGL(scaling_finalize<T>, dim3((k + 63) / 64), dim3(64), 0, c->stream, sums, k, norm_type, d);
        HIPCHK(hipGetLastError());
        return;
    }
    constexpr int VEC = 16 / sizeof(T);
    const bool vec = k % VEC == 0 && reinterpret_cast<uintptr_t>(X) % 16 == 0;
    int64_t nblk = ((vec ? total / VEC : total) + 255) / 256;
    if (nblk > 8 * (int64_t)c->num_cu) nblk = 8 * c->num_cu;
    if (nblk < 1) nblk = 1;
    if (vec) hipLaunchKernelGGL((scale_rows_from_sums_vec<T, VEC>), dim3((unsigned)nblk), dim3(256), 0, c->stream, X, k, total, sums, norm_type, d);
    else hipLaunchKernelGGL(scale_rows_from_sums<T>, dim3((unsigned)nblk), dim3(256), 0, c->stream, X, k, total, sums, norm_type, d);
    HIPCHK(hipGetLastError());
}
// ----------------------------------------------------------------------------
// k x k feature layer
// ----------------------------------------------------------------------------
extern "C" int rcppml_hip_row_norms(rcppml_hip_ctx* c, int dtype, const void* X, int k, int64_t ncols, int norm_type, void* out);
extern "C" int rcppml_hip_gram(rcppml_hip_ctx* c, int dtype, const void* F, int k, int64_t r, double eps, double l2, void* G);

template <class T>
static void apply_l21_impl(rcppml_hip_ctx* c, int dtype, T* G, const T* X, int k, int64_t ncols, T lambda) {
    if (!(lambda > T(0))) return;
    T* ss = static_cast<T*>(c->scratch(WS_FEAT, ((size_t)2 * k * k + k) * sizeof(T)));
    if (rcppml_hip_row_norms(c, dtype, X, k, ncols, 1, ss) != 0) throw std::runtime_error(rcppml_err());
    hipLaunchKernelGGL(l21_diag_kernel<T>, dim3((k + 63) / 64), dim3(64), 0, c->stream, G, ss, k, lambda);
    HIPCHK(hipGetLastError());
}
extern "C" int rcppml_hip_apply_l21(rcppml_hip_ctx* c, int dtype, void* G, const void* X, int k, int64_t ncols, double lambda) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (dtype == RCPPML_F32) apply_l21_impl<float>(c, dtype, (float*)G, (const float*)X, k, ncols, (float)lambda);
        else apply_l21_impl<double>(c, dtype, (double*)G, (const double*)X, k, ncols, lambda);
        return 0;
    }
    RCPPML_CATCH_RET
}
template <class T>
static void angular_impl(rcppml_hip_ctx* c, int dtype, T* X, int k, int64_t ncols, T lambda) {
    if (!(lambda > T(0)) || ncols <= 0) return;
    if (k > 128) throw std::runtime_error("angular_posthoc: k > 128 not supported");
    T* buf = static_cast<T*>(c->scratch(WS_FEAT, ((size_t)2 * k * k + k) * sizeof(T)));
    T* Gf = buf + k;
    T* M = Gf + (size_t)k * k;
    if (rcppml_hip_gram(c, dtype, X, k, ncols, 0.0, 0.0, Gf) != 0) throw std::runtime_error(rcppml_err());
    hipLaunchKernelGGL(angular_matrix_kernel<T>, dim3((k * k + 255) / 256), dim3(256), 0, c->stream, Gf, k, M);
    HIPCHK(hipGetLastError());
    int64_t nblk = (ncols + 3) / 4;
    if (nblk > 8 * (int64_t)c->num_cu) nblk = 8 * c->num_cu;
    const size_t smem = (size_t)k * k * sizeof(T);          // up to 128 KiB (fp64, k = 128)
    if (k <= 64) {
        hipLaunchKernelGGL((angular_apply_kernel<T, 1>), dim3((unsigned)nblk), dim3(256), smem, c->stream, X, k, ncols, M, lambda);
    } else {
        static DynSmemOnce once;
        once.ensure(reinterpret_cast<const void*>(&angular_apply_kernel<T, 2>), smem, c->device);
        hipLaunchKernelGGL((angular_apply_kernel<T, 2>), dim3((unsigned)nblk), dim3(256), smem, c->stream, X, k, ncols, M, lambda);
    }
    HIPCHK(hipGetLastError());
}
extern "C" int rcppml_hip_angular_posthoc(rcppml_hip_ctx* c, int dtype, void* X, int k, int64_t ncols, double lambda) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (dtype == RCPPML_F32) angular_impl<float>(c, dtype, (float*)X, k, ncols, (float)lambda);
        else angular_impl<double>(c, dtype, (double*)X, k, ncols, lambda);
        return 0;
    }
    RCPPML_CATCH_RET
}

extern "C" int rcppml_hip_rhs(rcppml_hip_ctx* c, int dtype, const int* col_ptr, const int* row_idx, const void* values, int64_t ncols,
                              const void* F, int k, void* B);
template <class T>
static void graph_reg_impl(rcppml_hip_ctx* c, int dtype, T* G, const int* lp, const int* li, const T* lx, const T* X, int k, int64_t ncols,
                           T lambda) {
    if (!(lambda > T(0)) || ncols <= 0) return;
    if (k > 128) throw std::runtime_error("apply_graph_reg: k > 128 not supported");
    int64_t nblk = (ncols + 511) / 512;
    if (nblk > 2 * (int64_t)c->num_cu) nblk = 2 * c->num_cu;
    if (nblk < 1) nblk = 1;
    // FL (k x ncols) | block partials (nblk x k x k)
    T* FL = static_cast<T*>(c->scratch(WS_GRAPH, ((size_t)k * ncols + (size_t)nblk * k * k) * sizeof(T)));
    T* part = FL + (size_t)k * ncols;
    if (rcppml_hip_rhs(c, dtype, lp, li, lx, ncols, X, k, FL) != 0) throw std::runtime_error(rcppml_err());    // FL(:,j) = sum_i L(i,j) X(:,i)
    if (k <= 64) hipLaunchKernelGGL((cross_gram_partial<T, 64>), dim3((unsigned)nblk), dim3(256), 0, c->stream, FL, X, k, ncols, part);
    else hipLaunchKernelGGL((cross_gram_partial<T, 128>), dim3((unsigned)nblk), dim3(256), 0, c->stream, FL, X, k, ncols, part);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(cross_gram_axpy<T>, dim3((k * k + 255) / 256), dim3(256), 0, c->stream, part, (int)nblk, k * k, lambda, G);
    HIPCHK(hipGetLastError());
}
extern "C" int rcppml_hip_apply_graph_reg(rcppml_hip_ctx* c, int dtype, void* G, const int* lap_p, const int* lap_i, const void* lap_x,
                                          const void* X, int k, int64_t ncols, double lambda) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (dtype == RCPPML_F32) graph_reg_impl<float>(c, dtype, (float*)G, lap_p, lap_i, (const float*)lap_x, (const float*)X, k, ncols, (float)lambda);
        else graph_reg_impl<double>(c, dtype, (double*)G, lap_p, lap_i, (const double*)lap_x, (const double*)X, k, ncols, lambda);
        return 0;
    }
    RCPPML_CATCH_RET
}
extern "C" int rcppml_hip_mul_rows(rcppml_hip_ctx* c, int dtype, const void* X, int k, int64_t ncols, const void* d, void* Y) {
    try {
        HIPCHK(hipSetDevice(c->device));
        const int64_t total = (int64_t)k * ncols;
        if (total <= 0) return 0;
        int64_t nblk = (total + 255) / 256;
        if (nblk > 8 * (int64_t)c->num_cu) nblk = 8 * c->num_cu;
        if (dtype == RCPPML_F32)
            hipLaunchKernelGGL(mul_rows<float>, dim3((unsigned)nblk), dim3(256), 0, c->stream, (const float*)X, k, total, (const float*)d, (float*)Y);
        else
            hipLaunchKernelGGL(mul_rows<double>, dim3((unsigned)nblk), dim3(256), 0, c->stream, (const double*)X, k, total, (const double*)d, (double*)Y);
        HIPCHK(hipGetLastError());
        return 0;
    }
    RCPPML_CATCH_RET
}
extern "C" int rcppml_hip_apply_scaling(rcppml_hip_ctx* c, int dtype, void* X, int k, int64_t ncols, int norm_type,
                                        const void* sums, void* d) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (dtype == RCPPML_F32) apply_scaling_impl<float>(c, (float*)X, k, ncols, norm_type, (const float*)sums, (float*)d);
        else apply_scaling_impl<double>(c, (double*)X, k, ncols, norm_type, (const double*)sums, (double*)d);
        return 0;
    }
    RCPPML_CATCH_RET
}

// ----------------------------------------------------------------------------
// Loss pieces
// ----------------------------------------------------------------------------
template <class T>
static void sumsq_impl(rcppml_hip_ctx* c, const T* x, int64_t len, double* out) {
    int64_t nblk = (len + 256 * 8 - 1) / (256 * 8);
    if (nblk > 4 * (int64_t)c->num_cu) nblk = 4 * c->num_cu;
    if (nblk < 1) nblk = 1;
    double* partial = static_cast<double*>(c->scratch(WS_RED, (size_t)nblk * sizeof(double)));
    hipLaunchKernelGGL(sumsq_partial<T>, dim3((unsigned)nblk), dim3(256), 0, c->stream, x, len, partial);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(sum_partials, dim3(1), dim3(256), 0, c->stream, partial, (int)nblk, out);
    HIPCHK(hipGetLastError());
}
extern "C" int rcppml_hip_sumsq(rcppml_hip_ctx* c, int dtype, const void* x, int64_t len, double* out) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (dtype == RCPPML_F32) sumsq_impl<float>(c, (const float*)x, len, out);
        else sumsq_impl<double>(c, (const double*)x, len, out);
        return 0;
    }
    RCPPML_CATCH_RET
}
template <class T>
static void loss_mse_impl(rcppml_hip_ctx* c, const double* trAtA, const T* d, const T* W_T, const T* B_w, int k,
                          int64_t m, const T* G_wt, const T* G_saved, double* out) {
    const int64_t total = (int64_t)k * m;
    int64_t nblk = (total + 256 * 8 - 1) / (256 * 8);
    if (nblk > 4 * (int64_t)c->num_cu) nblk = 4 * c->num_cu;
    if (nblk < 1) nblk = 1;
    double* partial = static_cast<double*>(c->scratch(WS_RED, (size_t)nblk * sizeof(double)));
    hipLaunchKernelGGL(cross_partial<T>, dim3((unsigned)nblk), dim3(256), 0, c->stream, W_T, B_w, d, k, total, partial);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(loss_mse_final<T>, dim3(1), dim3(256), 0, c->stream, trAtA, partial, (int)nblk, d, G_wt,
                       G_saved, k, out);
    HIPCHK(hipGetLastError());
}
extern "C" int rcppml_hip_loss_mse(rcppml_hip_ctx* c, int dtype, const double* trAtA, const void* d, const void* W_T,
                                   const void* B_w, int k, int64_t m, const void* G_wt, const void* G_saved,
                                   double* out) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (dtype == RCPPML_F32)
            loss_mse_impl<float>(c, trAtA, (const float*)d, (const float*)W_T, (const float*)B_w, k, m, (const float*)G_wt, (const float*)G_saved, out);
        else
            loss_mse_impl<double>(c, trAtA, (const double*)d, (const double*)W_T, (const double*)B_w, k, m, (const double*)G_wt, (const double*)G_saved, out);
        return 0;
    }
    RCPPML_CATCH_RET
}

// ----------------------------------------------------------------------------
// Explicit mask
// ----------------------------------------------------------------------------
template <class T>
static void solve_masked_impl(rcppml_hip_ctx* c, const int* cp, const int* ri, const T* vals, const int* mp,
                              const int* mi, int64_t ncols, const T* F, const T* Gfull, T* X, int k, T l1, T l2,
                              int nonneg, int maxit, T tol, int solver_mode, int warm) {
    if (ncols <= 0) return;
    if (k < 1 || k > 128) throw std::runtime_error("solve_masked: k must be in [1,128]");
    const int64_t nblk = (ncols + 3) / 4;
    if (k > 64) {          // one wave per column, two features per lane, Gram tile in LDS (kernels_wide.hip.h)
        auto kern = wide_masked_solve_kernel<T>;
        static DynSmemOnce once;
        once.ensure(reinterpret_cast<const void*>(kern), wide_smem_bytes<T>(), c->device);
        hipLaunchKernelGGL(kern, dim3((unsigned)ncols), dim3(64), wide_smem_bytes<T>(), c->stream, cp, ri, vals, mp, mi, ncols, F, Gfull, X, k,
                           l1, l2, nonneg, maxit, tol, solver_mode, warm);
        HIPCHK(hipGetLastError());
        return;
    }
    if (k <= 32) {     // 32-wide instantiation: 16 KB of LDS per block instead of 64 KB (8 waves per SIMD)
        const size_t smem = (size_t)4 * 32 * 32 * sizeof(T);
        hipLaunchKernelGGL((masked_solve_kernel<T, 32>), dim3((unsigned)nblk), dim3(256), smem, c->stream, cp, ri, vals, mp, mi,
                           ncols, F, Gfull, X, k, l1, l2, nonneg, maxit, tol, solver_mode, warm);
    } else {
        const size_t smem = (size_t)4 * 64 * 64 * sizeof(T);
        auto kern = masked_solve_kernel<T, 64>;
        static DynSmemOnce once;
        once.ensure(reinterpret_cast<const void*>(kern), smem, c->device);
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), smem, c->stream, cp, ri, vals, mp, mi, ncols, F, Gfull, X,
                           k, l1, l2, nonneg, maxit, tol, solver_mode, warm);
    }
    HIPCHK(hipGetLastError());
}
extern "C" int rcppml_hip_solve_masked(rcppml_hip_ctx* c, int dtype, const int* col_ptr, const int* row_idx,
                                       const void* values, const int* mask_p, const int* mask_i, int64_t ncols,
                                       const void* F, const void* G_full, void* X, int k, double l1, double l2,
                                       int nonneg, int cd_maxit, double cd_tol, int solver_mode, int warm) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (dtype == RCPPML_F32)
            solve_masked_impl<float>(c, col_ptr, row_idx, (const float*)values, mask_p, mask_i, ncols, (const float*)F,
                                     (const float*)G_full, (float*)X, k, (float)l1, (float)l2, nonneg, cd_maxit,
                                     (float)cd_tol, solver_mode, warm);
        else
            solve_masked_impl<double>(c, col_ptr, row_idx, (const double*)values, mask_p, mask_i, ncols, (const double*)F,
                                      (const double*)G_full, (double*)X, k, l1, l2, nonneg, cd_maxit, cd_tol,
                                      solver_mode, warm);
        return 0;
    }
    RCPPML_CATCH_RET
}
template <class T>
static void loss_nonzeros_impl(rcppml_hip_ctx* c, const int* cp, const int* ri, const T* vals, const int* mp,
                               const int* mi, int64_t ncols, const T* W_T, const T* d, const T* H, int k, double* out,
                               int loss_type = 0, double power = 1.5) {
    const int64_t nblk = ncols > 0 ? (ncols + 3) / 4 : 1;
    double* partial = static_cast<double*>(c->scratch(WS_RED2, (size_t)nblk * 2 * sizeof(double)));
    hipLaunchKernelGGL(loss_nonzeros_kernel<T>, dim3((unsigned)nblk), dim3(256), 0, c->stream, cp, ri, vals, mp, mi,
                       ncols, W_T, d, H, k, loss_type, power, partial);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(sum_partials2, dim3(1), dim3(256), 0, c->stream, partial, (int)nblk, out);
    HIPCHK(hipGetLastError());
}
extern "C" int rcppml_hip_loss_nonzeros(rcppml_hip_ctx* c, int dtype, const int* col_ptr, const int* row_idx,
                                        const void* values, const int* mask_p, const int* mask_i, int64_t ncols,
                                        const void* W_T, const void* d, const void* H, int k, double* out) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (dtype == RCPPML_F32)
            loss_nonzeros_impl<float>(c, col_ptr, row_idx, (const float*)values, mask_p, mask_i, ncols, (const float*)W_T,
                                      (const float*)d, (const float*)H, k, out);
        else
            loss_nonzeros_impl<double>(c, col_ptr, row_idx, (const double*)values, mask_p, mask_i, ncols,
                                       (const double*)W_T, (const double*)d, (const double*)H, k, out);
        return 0;
    }
    RCPPML_CATCH_RET
}

// The same pass with the per-element term of a distribution loss (math/loss.hpp:512-536, theta = 0): the loss of a fit with an
// explicit mask under GP / NB / Gamma / inverse-Gaussian / Tweedie (nmf/masked_nnls.hpp:250-282, fit_cpu.hpp:1685-1690).
extern "C" int rcppml_hip_loss_masked(rcppml_hip_ctx* c, int dtype, int loss_type, double tweedie_power, const int* col_ptr,
                                      const int* row_idx, const void* values, const int* mask_p, const int* mask_i, int64_t ncols,
                                      const void* W_T, const void* d, const void* H, int k, double* out) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (loss_type != 0 && (loss_type < 4 || loss_type > 8)) throw std::runtime_error("loss_masked: loss_type must be 0 or 4..8");
        if (dtype == RCPPML_F32)
            loss_nonzeros_impl<float>(c, col_ptr, row_idx, (const float*)values, mask_p, mask_i, ncols, (const float*)W_T,
                                      (const float*)d, (const float*)H, k, out, loss_type, tweedie_power);
        else
            loss_nonzeros_impl<double>(c, col_ptr, row_idx, (const double*)values, mask_p, mask_i, ncols,
                                       (const double*)W_T, (const double*)d, (const double*)H, k, out, loss_type, tweedie_power);
        return 0;
    }
    RCPPML_CATCH_RET
}

// ----------------------------------------------------------------------------
// Small helpers of the k x k feature layer: Y = X + alpha * T (target regularisation, variant_helpers.hpp:107-111:
// B += lambda * target) and G(i,i) += v.
// ----------------------------------------------------------------------------
template <class T>
static __global__ void axpy_kernel(const T* __restrict__ x, const T* __restrict__ t, T alpha, int64_t n, T* __restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = tfma(alpha, t[i], x[i]);
}
template <class T>
static __global__ void add_diag_kernel(T* __restrict__ G, int k, T v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < k) G[(size_t)i * k + i] += v;
}
extern "C" int rcppml_hip_axpy(rcppml_hip_ctx* c, int dtype, const void* X, const void* Tm, double alpha, int64_t n, void* Y) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (n <= 0) return 0;
        int64_t nblk = (n + 255) / 256;
        if (nblk > 8 * (int64_t)c->num_cu) nblk = 8 * c->num_cu;
        if (dtype == RCPPML_F32)
            hipLaunchKernelGGL(axpy_kernel<float>, dim3((unsigned)nblk), dim3(256), 0, c->stream, (const float*)X, (const float*)Tm, (float)alpha, n, (float*)Y);
        else
            hipLaunchKernelGGL(axpy_kernel<double>, dim3((unsigned)nblk), dim3(256), 0, c->stream, (const double*)X, (const double*)Tm, alpha, n, (double*)Y);
        HIPCHK(hipGetLastError());
        return 0;
    }
    RCPPML_CATCH_RET
}
extern "C" int rcppml_hip_add_diag(rcppml_hip_ctx* c, int dtype, void* G, int k, double v) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (k <= 0) return 0;
        if (dtype == RCPPML_F32) hipLaunchKernelGGL(add_diag_kernel<float>, dim3((k + 63) / 64), dim3(64), 0, c->stream, (float*)G, k, (float)v);
        else hipLaunchKernelGGL(add_diag_kernel<double>, dim3((k + 63) / 64), dim3(64), 0, c->stream, (double*)G, k, v);
        HIPCHK(hipGetLastError());
        return 0;
    }
    RCPPML_CATCH_RET
}

// X = min(X, ub) elementwise -- features/bounds.hpp apply_upper_bound, applied after every half-update branch
// (nmf/fit_cpu.hpp:636-637, :884-885); the CD / Cholesky ops fuse it (ub_post), the IRLS and explicit-mask solves call this.
template <class T>
static __global__ void clip_upper_kernel(T* __restrict__ x, int64_t n, T ub) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const T v = x[i];
        if (v > ub) x[i] = ub;
    }
}
extern "C" int rcppml_hip_clip_upper(rcppml_hip_ctx* c, int dtype, void* X, int64_t n, double ub) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (n <= 0 || !(ub > 0)) return 0;
        int64_t nblk = (n + 255) / 256;
        if (nblk > 8 * (int64_t)c->num_cu) nblk = 8 * c->num_cu;
        if (dtype == RCPPML_F32) hipLaunchKernelGGL(clip_upper_kernel<float>, dim3((unsigned)nblk), dim3(256), 0, c->stream, (float*)X, n, (float)ub);
        else hipLaunchKernelGGL(clip_upper_kernel<double>, dim3((unsigned)nblk), dim3(256), 0, c->stream, (double*)X, n, ub);
        HIPCHK(hipGetLastError());
        return 0;
    }
    RCPPML_CATCH_RET
}
