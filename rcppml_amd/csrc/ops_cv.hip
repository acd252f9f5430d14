// ops_cv.hip -- cross-validation half-update and held-out error (device-level C ABI, include/rcppml_gpu.h layer 2)
#include <type_traits>
#include <cstring>
#include "common.hip.h"
#include "kernels.hip.h"
#include "kernels_wide.hip.h"

using namespace rk;

namespace {
// speckled_cv.hpp:57-68: seed = (uint32) cv_seed, 0 -> 12345; inv_prob = (uint64)(1 / fraction); threshold = UINT64_MAX / inv_prob
void mask_params(double holdout_fraction, unsigned long long cv_seed, unsigned long long* seed, unsigned long long* threshold) {
    if (!(holdout_fraction > 0.0 && holdout_fraction < 1.0)) throw std::runtime_error("cv: holdout_fraction must be in (0, 1)");
    const unsigned s32 = static_cast<unsigned>(cv_seed);
    *seed = s32 == 0 ? 12345ULL : static_cast<unsigned long long>(s32);
    const unsigned long long inv_prob = static_cast<unsigned long long>(1.0 / holdout_fraction);
    if (inv_prob == 0) throw std::runtime_error("cv: holdout_fraction too large");
    *threshold = 0xFFFFFFFFFFFFFFFFULL / inv_prob;
}

template <class T>
void cv_solve_impl(rcppml_hip_ctx* c, const int* cp, const int* ri, const T* vals, int64_t ncols, int nrows, const T* F,
                   const T* G, T* X, int k, double frac, unsigned long long cv_seed, int mask_zeros, int transposed, T l1,
                   int nonneg, int maxit, int solver_mode) {
    if (ncols <= 0) return;
    if (k < 1 || k > 128) throw std::runtime_error("solve_cv: k must be in [1,128]");
    if (solver_mode != 0 && solver_mode != 1) throw std::runtime_error("solve_cv: solver_mode must be 0 (CD) or 1 (Cholesky+clip)");
    unsigned long long seed, thr;
    mask_params(frac, cv_seed, &seed, &thr);
    // user mask of the fit (rcppml_hip_ctx_set_cv_mask; cv_detail.hpp:433-505): the generic kernels take it, the MFMA forms below do not
    const int* mp = c->cv_mask_p[transposed ? 1 : 0];
    const int* mi = c->cv_mask_i[transposed ? 1 : 0];
    if (k > 64) {          // one wave per column, two features per lane, Gram tile in LDS (kernels_wide.hip.h)
        auto kern = wide_cv_solve_kernel<T>;
        static DynSmemOnce once;
        once.ensure(reinterpret_cast<const void*>(kern), wide_smem_bytes<T>(), c->device);
        hipLaunchKernelGGL(kern, dim3((unsigned)ncols), dim3(64), wide_smem_bytes<T>(), c->stream, cp, ri, vals, ncols, nrows, F, G, X, k,
                           seed, thr, mask_zeros, transposed, l1, nonneg, maxit, solver_mode, mp, mi);
        HIPCHK(hipGetLastError());
        return;
    }
    const int64_t nblk = (ncols + 3) / 4;
    if constexpr (std::is_same<T, float>::value) if (!mp) {
        // fp32, k <= 32: Gram correction on the matrix cores (RCPPML_GPU_CV_VARIANT=valu keeps the LDS read-modify-write form)
        static int use_mfma = -1;
        if (use_mfma < 0) use_mfma = exp_flag("RCPPML_GPU_CV_VARIANT", "valu") ? 0 : 1;
        if (use_mfma && k <= 32 && k % 4 == 0 && reinterpret_cast<uintptr_t>(F) % 16 == 0) {
            const size_t smem = (size_t)4 * (32 * 36 + 96) * sizeof(float);
            hipLaunchKernelGGL(cv_solve_mfma32_kernel, dim3((unsigned)nblk), dim3(256), smem, c->stream, cp, ri, vals, ncols, nrows, F, G,
                               X, k, seed, thr, mask_zeros, transposed, l1, nonneg, maxit, solver_mode);
            HIPCHK(hipGetLastError());
            return;
        }
        if (use_mfma && k <= 64 && k % 4 == 0 && reinterpret_cast<uintptr_t>(F) % 16 == 0) {      // 32 < k <= 64: 2 x 2 tiles
            const size_t smem = (size_t)4 * (64 * 64 + 96) * sizeof(float);
            static DynSmemOnce once;
            once.ensure(reinterpret_cast<const void*>(&cv_solve_mfma32x2_kernel), smem, c->device);
            hipLaunchKernelGGL(cv_solve_mfma32x2_kernel, dim3((unsigned)nblk), dim3(256), smem, c->stream, cp, ri, vals, ncols, nrows, F, G,
                               X, k, seed, thr, mask_zeros, transposed, l1, nonneg, maxit, solver_mode);
            HIPCHK(hipGetLastError());
            return;
        }
    }
    if constexpr (std::is_same<T, double>::value) if (!mp) {
        static int use_mfma64 = -1;
        if (use_mfma64 < 0) use_mfma64 = exp_flag("RCPPML_GPU_CV_VARIANT", "valu") ? 0 : 1;
        if (use_mfma64 && k <= 32 && k % 2 == 0 && reinterpret_cast<uintptr_t>(F) % 16 == 0) {
            const size_t smem = (size_t)4 * (32 * 34 + 48) * sizeof(double);
            hipLaunchKernelGGL(cv_solve_mfma64_kernel, dim3((unsigned)nblk), dim3(256), smem, c->stream, cp, ri, vals, ncols, nrows, F, G,
                               X, k, seed, thr, mask_zeros, transposed, l1, nonneg, maxit, solver_mode);
            HIPCHK(hipGetLastError());
            return;
        }
    }
    if (k <= 32) {
        hipLaunchKernelGGL((cv_solve_kernel<T, 32>), dim3((unsigned)nblk), dim3(256), (size_t)4 * 32 * 32 * sizeof(T), c->stream, cp, ri,
                           vals, ncols, nrows, F, G, X, k, seed, thr, mask_zeros, transposed, l1, nonneg, maxit, solver_mode, mp, mi);
    } else {
        const size_t smem = (size_t)4 * 64 * 64 * sizeof(T);
        auto kern = cv_solve_kernel<T, 64>;
        static DynSmemOnce once;
        once.ensure(reinterpret_cast<const void*>(kern), smem, c->device);
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), smem, c->stream, cp, ri, vals, ncols, nrows, F, G, X, k, seed, thr,
                           mask_zeros, transposed, l1, nonneg, maxit, solver_mode, mp, mi);
    }
    HIPCHK(hipGetLastError());
}

__global__ void cv_sum_partials_kernel(const double* __restrict__ ps, const unsigned long long* __restrict__ pn, int n,
                                       double* __restrict__ out2) {
    __shared__ double ss[256];
    __shared__ unsigned long long sn[256];
    double a = 0.0;
    unsigned long long b = 0;
    for (int i = threadIdx.x; i < n; i += 256) { a += ps[i]; b += pn[i]; }
    ss[threadIdx.x] = a; sn[threadIdx.x] = b;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) { ss[threadIdx.x] += ss[threadIdx.x + off]; sn[threadIdx.x] += sn[threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out2[0] = ss[0]; out2[1] = static_cast<double>(sn[0]); }
}

template <class T>
void cv_test_impl(rcppml_hip_ctx* c, const int* cp, const int* ri, const T* vals, int64_t ncols, int nrows, const T* W_T,
                  const T* d, const T* H, int k, double frac, unsigned long long cv_seed, int mask_zeros, double* out2) {
    if (k < 1 || k > 128) throw std::runtime_error("cv_test_error: k must be in [1,128]");
    unsigned long long seed, thr;
    mask_params(frac, cv_seed, &seed, &thr);
    const int64_t nblk = ncols > 0 ? (ncols + 3) / 4 : 1;
    char* buf = static_cast<char*>(c->scratch(WS_RED2, (size_t)nblk * 16));
    double* ps = reinterpret_cast<double*>(buf);
    unsigned long long* pn = reinterpret_cast<unsigned long long*>(buf + (size_t)nblk * 8);
    hipLaunchKernelGGL(cv_test_error_kernel<T>, dim3((unsigned)nblk), dim3(256), 0, c->stream, cp, ri, vals, ncols, nrows, W_T, d, H, k,
                       seed, thr, mask_zeros, ps, pn);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(cv_sum_partials_kernel, dim3(1), dim3(256), 0, c->stream, ps, pn, (int)nblk, out2);
    HIPCHK(hipGetLastError());
}
}  // namespace

extern "C" int rcppml_hip_solve_cv(rcppml_hip_ctx* c, int dtype, const int* col_ptr, const int* row_idx, const void* values,
                                   int64_t ncols, int nrows, const void* F, const void* G, void* X, int k,
                                   double holdout_fraction, unsigned long long cv_seed, int mask_zeros, int transposed,
                                   double l1, int nonneg, int cd_maxit, int solver_mode) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (dtype == RCPPML_F32)
            cv_solve_impl<float>(c, col_ptr, row_idx, (const float*)values, ncols, nrows, (const float*)F, (const float*)G, (float*)X, k,
                                 holdout_fraction, cv_seed, mask_zeros, transposed, (float)l1, nonneg, cd_maxit, solver_mode);
        else
            cv_solve_impl<double>(c, col_ptr, row_idx, (const double*)values, ncols, nrows, (const double*)F, (const double*)G,
                                  (double*)X, k, holdout_fraction, cv_seed, mask_zeros, transposed, l1, nonneg, cd_maxit, solver_mode);
        return 0;
    }
    RCPPML_CATCH_RET
}

extern "C" int rcppml_hip_cv_test_error(rcppml_hip_ctx* c, int dtype, const int* col_ptr, const int* row_idx, const void* values,
                                        int64_t ncols, int nrows, const void* W_T, const void* d, const void* H, int k,
                                        double holdout_fraction, unsigned long long cv_seed, int mask_zeros, double* out2) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (dtype == RCPPML_F32)
            cv_test_impl<float>(c, col_ptr, row_idx, (const float*)values, ncols, nrows, (const float*)W_T, (const float*)d,
                                (const float*)H, k, holdout_fraction, cv_seed, mask_zeros, out2);
        else
            cv_test_impl<double>(c, col_ptr, row_idx, (const double*)values, ncols, nrows, (const double*)W_T, (const double*)d,
                                 (const double*)H, k, holdout_fraction, cv_seed, mask_zeros, out2);
        return 0;
    }
    RCPPML_CATCH_RET
}

// The user mask of a cross-validation fit (reference NMFConfig::mask under nmf_fit_cv, nmf/fit_cv.hpp:327-331): pattern CSC of the mask
// (m x n) and of its transpose, device pointers that stay valid until the mask is cleared (all four NULL).  While set, rcppml_hip_solve_cv,
// rcppml_hip_solve_cv_irls and rcppml_hip_cv_irls_loss treat its entries as the reference does (cv_detail.hpp:433-505, fit_cv.hpp:1391-1427).
extern "C" int rcppml_hip_ctx_set_cv_mask(rcppml_hip_ctx* c, const int* mask_p, const int* mask_i, const int* maskT_p, const int* maskT_i) {
    if (!c) return -1;
    if ((mask_p == nullptr) != (maskT_p == nullptr) || (mask_p && (!mask_i || !maskT_i))) { rcppml_err() = "set_cv_mask: give the mask and its transpose, or neither"; return -1; }
    c->cv_mask_p[0] = mask_p; c->cv_mask_i[0] = mask_i;
    c->cv_mask_p[1] = maskT_p; c->cv_mask_i[1] = maskT_i;
    return 0;
}
