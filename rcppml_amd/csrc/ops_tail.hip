// ops_tail.hip -- the iteration's tail between two solves in fewer launches (kernels_tail.hip.h): scaling + the next solve's work
// order, Gram + loss, and the two together (device-level C ABI, include/rcppml_gpu.h layer 2)
#include "common.hip.h"
#include "kernels.hip.h"
#include "kernels_tail.hip.h"
#include "gram_launch.hip.h"

using namespace rk;
extern "C" int rcppml_hip_row_norms(rcppml_hip_ctx* c, int dtype, const void* X, int k, int64_t ncols, int norm_type, void* out);
extern "C" int rcppml_hip_gram(rcppml_hip_ctx* c, int dtype, const void* F, int k, int64_t r, double eps, double l2, void* G);

// ----------------------------------------------------------------------------
// pieces
// ----------------------------------------------------------------------------
struct TailOrder { unsigned int* part = nullptr; int64_t nbo = 0; };
// the work-order table of `ncols` columns (grid of rcppml_hip_order_columns)
static TailOrder tail_order_setup(rcppml_hip_ctx* c, int64_t ncols, const int* sweeps, const int* order) {
    TailOrder o;
    if (sweeps && order) {
        o.nbo = (ncols + 1023) / 1024;
        if (o.nbo > ORDER_BLOCKS_MAX) o.nbo = ORDER_BLOCKS_MAX;
        if (o.nbo < 1) o.nbo = 1;
        o.part = static_cast<unsigned int*>(c->scratch(WS_ORDER, (size_t)ORDER_BLOCKS_MAX * 128 * sizeof(unsigned int)));
    }
    return o;
}
// row sums of X into `sums` (grid, bodies and summation order of row_norms_impl, ops_misc.hip), the work-order histogram beside the
// partial sums: two launches
template <class T>
static void tail_norms(rcppml_hip_ctx* c, const T* X, int k, int64_t ncols, int norm_type, T* sums, const int* sweeps, const TailOrder& o) {
    int64_t nbn = (ncols + 255) / 256;
    if (nbn > 2 * (int64_t)c->num_cu) nbn = 2 * c->num_cu;
    if (nbn < 1) nbn = 1;
    constexpr int VEC = 16 / sizeof(T);
    const bool vec = k % VEC == 0 && k / VEC <= 256 && reinterpret_cast<uintptr_t>(X) % 16 == 0;       // row_norms_impl's choice
    T* partial = static_cast<T*>(c->scratch(WS_RED, (size_t)nbn * k * sizeof(T)));
    if (vec) {
        const int slots = 256 / (k / VEC);
        hipLaunchKernelGGL((tail_norm_hist_kernel<T, VEC>), dim3((unsigned)(o.nbo + nbn)), dim3(256), (size_t)slots * k * sizeof(T), c->stream,
                           X, k, ncols, norm_type, partial, (unsigned)nbn, sweeps, o.part, (unsigned)o.nbo);
    } else {
        hipLaunchKernelGGL((tail_norm_hist_kernel<T, 0>), dim3((unsigned)(o.nbo + nbn)), dim3(256), 256 * sizeof(T), c->stream,
                           X, k, ncols, norm_type, partial, (unsigned)nbn, sweeps, o.part, (unsigned)o.nbo);
    }
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(row_norm_final<T>, dim3(k), dim3(64), 0, c->stream, partial, (int)nbn, k, sums);
    HIPCHK(hipGetLastError());
}
// X(i,:) /= d_i and d from `sums` (grid and body of apply_scaling_impl), the work-order scatter beside it: one launch
template <class T>
static void tail_scale(rcppml_hip_ctx* c, T* X, int k, int64_t ncols, int norm_type, const T* sums, T* d, const int* sweeps, int* order,
                       const TailOrder& o) {
    const int64_t total = (int64_t)k * ncols;
    constexpr int VEC = 16 / sizeof(T);
    const bool vec_s = k % VEC == 0 && reinterpret_cast<uintptr_t>(X) % 16 == 0;       // apply_scaling_impl's choice
    int64_t nbs = ((vec_s ? total / VEC : total) + 255) / 256;
    if (nbs > 8 * (int64_t)c->num_cu) nbs = 8 * c->num_cu;
    if (nbs < 1) nbs = 1;
    if (vec_s)
        hipLaunchKernelGGL((tail_scale_scatter_kernel<T, VEC>), dim3((unsigned)(o.nbo + nbs)), dim3(256), 0, c->stream, X, k, total, sums, norm_type, d,
                           (unsigned)nbs, sweeps, ncols, o.part, order, (unsigned)o.nbo);
    else
        hipLaunchKernelGGL((tail_scale_scatter_kernel<T, 0>), dim3((unsigned)(o.nbo + nbs)), dim3(256), 0, c->stream, X, k, total, sums, norm_type, d,
                           (unsigned)nbs, sweeps, ncols, o.part, order, (unsigned)o.nbo);
    HIPCHK(hipGetLastError());
}
// fp32 k = 64: scaling inside the Gram's partial-tile kernel (+ the scatter beside it); returns the partial tiles, grid of gram_partials
static const float* tail_scale_gram_k64(rcppml_hip_ctx* c, float* X, int64_t ncols, int norm_type, const float* sums, float* d,
                                        const int* sweeps, int* order, const TailOrder& o, int* nblk_out) {
    int64_t nblk = (ncols / 2 + 4 * 32 - 1) / (4 * 32);          // gram_partials<float>'s grid: the same partial tiles
    if (nblk < 1) nblk = 1;
    if (nblk > 2 * (int64_t)c->num_cu) nblk = 2 * c->num_cu;
    float* partial = static_cast<float*>(c->scratch(WS_GRAM, (size_t)nblk * 64 * 64 * sizeof(float)));
    hipLaunchKernelGGL((tail_scale_gram_k64_kernel<8>), dim3((unsigned)(o.nbo + nblk)), dim3(256), 0, c->stream, X, 64, ncols, sums, norm_type, d,
                       partial, (unsigned)nblk, sweeps, o.part, order, (unsigned)o.nbo);
    HIPCHK(hipGetLastError());
    *nblk_out = (int)nblk;
    return partial;
}
static bool k64_fusable(int dtype, const void* X, int k, int64_t ncols, int norm_type) {
    return dtype == RCPPML_F32 && k == 64 && ncols > 0 && (norm_type == 0 || norm_type == 1) && reinterpret_cast<uintptr_t>(X) % 16 == 0;
}
// cross-term partials beside the Gram's final sum, then the loss's final sum: two launches
template <class T>
static void tail_gramfin_loss(rcppml_hip_ctx* c, const T* partial, int nblk, int KP, const T* W_T, int k, int64_t m, T eps, const double* trAtA,
                              const T* d, const T* B_w, const T* G_saved, T* G_wt, double* out) {
    const int64_t total = (int64_t)k * m;
    int64_t nbc = (total + 256 * 8 - 1) / (256 * 8);          // = loss_mse_impl's grid (ops_misc.hip): the same partial sums
    if (nbc > 4 * (int64_t)c->num_cu) nbc = 4 * c->num_cu;
    if (nbc < 1) nbc = 1;
    double* cpart = static_cast<double*>(c->scratch(WS_RED, (size_t)nbc * sizeof(double)));
    const unsigned nfin = (unsigned)((KP * KP + 7) / 8);
    hipLaunchKernelGGL(tail_gramfin_cross_kernel<T>, dim3((unsigned)nbc + nfin), dim3(256), 0, c->stream, partial, nblk, KP, k, eps, T(0), G_wt,
                       W_T, B_w, d, total, cpart, (unsigned)nbc);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(loss_mse_final<T>, dim3(1), dim3(256), 0, c->stream, trAtA, cpart, (int)nbc, d, G_wt, G_saved, k, out);
    HIPCHK(hipGetLastError());
}

// ----------------------------------------------------------------------------
// rcppml_hip_scale_order: extract_scaling (+ the next solve's work order), three launches instead of five
// ----------------------------------------------------------------------------
extern "C" int rcppml_hip_order_columns(rcppml_hip_ctx* c, const int* sweeps, int64_t ncols, int* order);
extern "C" int rcppml_hip_apply_scaling(rcppml_hip_ctx* c, int dtype, void* X, int k, int64_t ncols, int norm_type, const void* sums, void* d);
static int scale_order_separate(rcppml_hip_ctx* c, int dtype, void* X, int k, int64_t ncols, int norm_type, void* sums, void* d,
                                const int* sweeps, int* order) {
    if (rcppml_hip_row_norms(c, dtype, X, k, ncols, norm_type, sums) != 0) return 1;
    if (rcppml_hip_apply_scaling(c, dtype, X, k, ncols, norm_type, sums, d) != 0) return 1;
    if (sweeps && order && ncols > 0) return rcppml_hip_order_columns(c, sweeps, ncols, order);
    return 0;
}
extern "C" int rcppml_hip_scale_order(rcppml_hip_ctx* c, int dtype, void* X, int k, int64_t ncols, int norm_type, void* sums, void* d,
                                      const int* sweeps, int* order) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (norm_type == 2 || ncols <= 0 || k <= 0)       // no scaling pass to share a launch with: the separate ops
            return scale_order_separate(c, dtype, X, k, ncols, norm_type, sums, d, sweeps, order);
        const TailOrder o = tail_order_setup(c, ncols, sweeps, order);
        if (dtype == RCPPML_F32) {
            tail_norms<float>(c, (const float*)X, k, ncols, norm_type, (float*)sums, sweeps, o);
            tail_scale<float>(c, (float*)X, k, ncols, norm_type, (const float*)sums, (float*)d, sweeps, order, o);
        } else {
            tail_norms<double>(c, (const double*)X, k, ncols, norm_type, (double*)sums, sweeps, o);
            tail_scale<double>(c, (double*)X, k, ncols, norm_type, (const double*)sums, (double*)d, sweeps, order, o);
        }
        return 0;
    }
    RCPPML_CATCH_RET
}
// ----------------------------------------------------------------------------
// rcppml_hip_gram_loss_mse: Gram of W_T (+ eps) into G_wt and the MSE loss by the Gram trick, three launches instead of four
// ----------------------------------------------------------------------------
template <class T>
static void gram_loss_mse_impl(rcppml_hip_ctx* c, const T* W_T, int k, int64_t m, T eps, const double* trAtA, const T* d, const T* B_w,
                               const T* G_saved, T* G_wt, double* out) {
    int nblk = 0, KP = 0;
    const T* partial = gram_partials<T>(c, W_T, k, m, &nblk, &KP);
    tail_gramfin_loss<T>(c, partial, nblk, KP, W_T, k, m, eps, trAtA, d, B_w, G_saved, G_wt, out);
}
extern "C" int rcppml_hip_gram_loss_mse(rcppml_hip_ctx* c, int dtype, const void* W_T, int k, int64_t m, double eps, const double* trAtA,
                                        const void* d, const void* B_w, const void* G_saved, void* G_wt, double* out) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (dtype == RCPPML_F32)
            gram_loss_mse_impl<float>(c, (const float*)W_T, k, m, (float)eps, trAtA, (const float*)d, (const float*)B_w, (const float*)G_saved, (float*)G_wt, out);
        else
            gram_loss_mse_impl<double>(c, (const double*)W_T, k, m, eps, trAtA, (const double*)d, (const double*)B_w, (const double*)G_saved, (double*)G_wt, out);
        return 0;
    }
    RCPPML_CATCH_RET
}
// ----------------------------------------------------------------------------
// rcppml_hip_tail_scale_gram = rcppml_hip_scale_order, then rcppml_hip_gram(X, eps, l2) -> G            (the H side's tail)
// rcppml_hip_tail_scale_gram_loss = rcppml_hip_scale_order, then rcppml_hip_gram_loss_mse               (the W side's tail)
// fp32 k = 64: the scaling pass runs inside the Gram's partial-tile kernel (four and five launches); otherwise the two ops above
// ----------------------------------------------------------------------------
extern "C" int rcppml_hip_tail_scale_gram(rcppml_hip_ctx* c, int dtype, void* X, int k, int64_t ncols, int norm_type, void* sums, void* d,
                                          const int* sweeps, int* order, double eps, double l2, void* G) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (!k64_fusable(dtype, X, k, ncols, norm_type)) {
            if (rcppml_hip_scale_order(c, dtype, X, k, ncols, norm_type, sums, d, sweeps, order) != 0) return 1;
            return rcppml_hip_gram(c, dtype, X, k, ncols, eps, l2, G);
        }
        const TailOrder o = tail_order_setup(c, ncols, sweeps, order);
        tail_norms<float>(c, (const float*)X, k, ncols, norm_type, (float*)sums, sweeps, o);
        int nblk = 0;
        const float* partial = tail_scale_gram_k64(c, (float*)X, ncols, norm_type, (const float*)sums, (float*)d, sweeps, order, o, &nblk);
        hipLaunchKernelGGL(gram_finalize<float>, dim3((64 * 64 + 7) / 8), dim3(256), 0, c->stream, partial, nblk, 64, k, (float)eps, (float)l2, (float*)G);
        HIPCHK(hipGetLastError());
        return 0;
    }
    RCPPML_CATCH_RET
}
extern "C" int rcppml_hip_tail_scale_gram_loss(rcppml_hip_ctx* c, int dtype, void* W_T, int k, int64_t m, int norm_type, void* sums, void* d,
                                               const int* sweeps, int* order, double eps, const double* trAtA, const void* B_w,
                                               const void* G_saved, void* G_wt, double* out) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (!k64_fusable(dtype, W_T, k, m, norm_type)) {
            if (rcppml_hip_scale_order(c, dtype, W_T, k, m, norm_type, sums, d, sweeps, order) != 0) return 1;
            return rcppml_hip_gram_loss_mse(c, dtype, W_T, k, m, eps, trAtA, d, B_w, G_saved, G_wt, out);
        }
        const TailOrder o = tail_order_setup(c, m, sweeps, order);
        tail_norms<float>(c, (const float*)W_T, k, m, norm_type, (float*)sums, sweeps, o);
        int nblk = 0;
        const float* partial = tail_scale_gram_k64(c, (float*)W_T, m, norm_type, (const float*)sums, (float*)d, sweeps, order, o, &nblk);
        tail_gramfin_loss<float>(c, partial, nblk, 64, (const float*)W_T, k, m, (float)eps, trAtA, (const float*)d, (const float*)B_w,
                                 (const float*)G_saved, (float*)G_wt, out);
        return 0;
    }
    RCPPML_CATCH_RET
}
