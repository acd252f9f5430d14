// ops_solve.hip -- CD dispatch + Cholesky solve (device-level C ABI)
#include "solve_common.hip.h"
void rcppml_solve_cd_f32(rcppml_hip_ctx* c, const float* G, const float* B, float* X, int k, int64_t ncols, float l1_pre,
                         int warm, int zero_init, float l1_cd, float l2_cd, int nonneg, int maxit, float tol, float ub_cd,
                         float ub_post, int variant, int* sweeps, const int* order);
void rcppml_solve_cd_f64(rcppml_hip_ctx* c, const double* G, const double* B, double* X, int k, int64_t ncols, double l1_pre,
                         int warm, int zero_init, double l1_cd, double l2_cd, int nonneg, int maxit, double tol, double ub_cd,
                         double ub_post, int variant, int* sweeps, const int* order);
extern "C" int rcppml_hip_solve_cd(rcppml_hip_ctx* c, int dtype, const void* G, const void* B, void* X, int k,
                                   int64_t ncols, double l1_pre, int warm, int zero_init, double l1_cd,
                                   double l2_cd, int nonneg, int maxit, double tol, double ub_cd, double ub_post,
                                   int variant, int* sweeps_out, const int* col_order) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (dtype == RCPPML_F32)
            rcppml_solve_cd_f32(c, (const float*)G, (const float*)B, (float*)X, k, ncols, (float)l1_pre, warm,
                                 zero_init, (float)l1_cd, (float)l2_cd, nonneg, maxit, (float)tol, (float)ub_cd,
                                 (float)ub_post, variant, sweeps_out, col_order);
        else
            rcppml_solve_cd_f64(c, (const double*)G, (const double*)B, (double*)X, k, ncols, l1_pre, warm,
                                  zero_init, l1_cd, l2_cd, nonneg, maxit, tol, ub_cd, ub_post, variant, sweeps_out, col_order);
        return 0;
    }
    RCPPML_CATCH_RET
}

// ----------------------------------------------------------------------------
// Cholesky solve + clip
// ----------------------------------------------------------------------------
template <class T, int KP>
static void chol_launch(rcppml_hip_ctx* c, const T* Gp, const T* B, T* X, int k, int64_t ncols, T l1_pre,
                        int nonneg, T ub_post) {
    T* L = static_cast<T*>(c->scratch(WS_CHOL, ((size_t)KP * KP + KP) * sizeof(T)));
    T* invl = L + (size_t)KP * KP;
    hipLaunchKernelGGL((chol_factor_kernel<T, KP>), dim3(1), dim3(64), 0, c->stream, Gp, L, invl);
    HIPCHK(hipGetLastError());
    const int64_t nblk = (ncols + 63) / 64;
    hipLaunchKernelGGL((chol_solve_kernel<T, KP>), dim3((unsigned)nblk), dim3(64), 0, c->stream, L, B, X, k, ncols,
                       l1_pre, nonneg, ub_post);
    HIPCHK(hipGetLastError());
}
template <class T>
static void solve_chol_impl(rcppml_hip_ctx* c, const T* G, const T* B, T* X, int k, int64_t ncols, T l1_pre,
                            int nonneg, T ub_post) {
    if (ncols <= 0) return;
    const int lane_max = std::is_same<T, float>::value ? 128 : 64;
    if (k < 1 || k > lane_max) throw std::runtime_error("solve_chol: k out of supported range");
    const int KP = solve_kp(k);
    T *Gp, *invd;
    pad_impl<T>(c, G, k, KP, &Gp, &invd);
    switch (KP) {
        case 16: chol_launch<T, 16>(c, Gp, B, X, k, ncols, l1_pre, nonneg, ub_post); break;
        case 32: chol_launch<T, 32>(c, Gp, B, X, k, ncols, l1_pre, nonneg, ub_post); break;
        case 64: chol_launch<T, 64>(c, Gp, B, X, k, ncols, l1_pre, nonneg, ub_post); break;
        default:
            if constexpr (std::is_same<T, float>::value) chol_launch<T, 128>(c, Gp, B, X, k, ncols, l1_pre, nonneg, ub_post);
            break;
    }
}
extern "C" int rcppml_hip_solve_chol(rcppml_hip_ctx* c, int dtype, const void* G, const void* B, void* X, int k,
                                     int64_t ncols, double l1_pre, int nonneg, double ub_post) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (dtype == RCPPML_F32)
            solve_chol_impl<float>(c, (const float*)G, (const float*)B, (float*)X, k, ncols, (float)l1_pre, nonneg, (float)ub_post);
        else
            solve_chol_impl<double>(c, (const double*)G, (const double*)B, (double*)X, k, ncols, l1_pre, nonneg, ub_post);
        return 0;
    }
    RCPPML_CATCH_RET
}


// ----------------------------------------------------------------------------
// Column work order (counting sort by descending sweep count)
// ----------------------------------------------------------------------------
extern "C" int rcppml_hip_order_columns(rcppml_hip_ctx* c, const int* sweeps, int64_t ncols, int* order) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (ncols <= 0) return 0;
        int64_t nblk = (ncols + 1023) / 1024;                       // >= 1024 columns per block; both kernels use the same grid
        if (nblk > ORDER_BLOCKS_MAX) nblk = ORDER_BLOCKS_MAX;
        if (nblk < 1) nblk = 1;
        unsigned int* part = static_cast<unsigned int*>(c->scratch(WS_ORDER, (size_t)ORDER_BLOCKS_MAX * 128 * sizeof(unsigned int)));
        hipLaunchKernelGGL(order_hist_kernel, dim3((unsigned)nblk), dim3(256), 0, c->stream, sweeps, ncols, part);
        hipLaunchKernelGGL(order_scatter_kernel, dim3((unsigned)nblk), dim3(256), 0, c->stream, sweeps, ncols, part, order);
        HIPCHK(hipGetLastError());
        return 0;
    }
    RCPPML_CATCH_RET
}
