// ops_gram.hip -- context + Gram (device-level C ABI, include/rcppml_gpu.h layer 2)
#include "common.hip.h"
#include "kernels.hip.h"
#include "kernels_tail.hip.h"

using namespace rk;
// ----------------------------------------------------------------------------
// context
// ----------------------------------------------------------------------------
extern "C" int rcppml_hip_ctx_create(rcppml_hip_ctx** out, int device, void* stream) {
    try {
        int ndev = 0;
        HIPCHK(hipGetDeviceCount(&ndev));
        if (ndev <= 0) throw std::runtime_error("no HIP device visible");
        if (device < 0 || device >= ndev) throw std::runtime_error("device index out of range");
        HIPCHK(hipSetDevice(device));
        rcppml_hip_ctx* c = new rcppml_hip_ctx();
        c->device = device;
        c->stream = static_cast<hipStream_t>(stream);
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, device));
        c->num_cu = prop.multiProcessorCount;
        HIPCHK(hipMalloc(&c->stats, 8 * sizeof(unsigned long long)));
        HIPCHK(hipMemset(c->stats, 0, 8 * sizeof(unsigned long long)));
        *out = c;
        return 0;
    }
    RCPPML_CATCH_RET
}
extern "C" void rcppml_hip_ctx_destroy(rcppml_hip_ctx* c) {
    if (!c) return;
    for (auto& b : c->bufs)
        if (b.ptr) (void)hipFree(b.ptr);
    if (c->stats) (void)hipFree(c->stats);
    delete c;
}
extern "C" int rcppml_hip_ctx_sync(rcppml_hip_ctx* c) {
    try { HIPCHK(hipStreamSynchronize(c->stream)); return 0; }
    RCPPML_CATCH_RET
}
extern "C" int rcppml_hip_ctx_set_option(rcppml_hip_ctx* c, int option, int value) {
    if (!c) return 1;
    switch (option) {
        case RCPPML_OPT_CD_COUNT_NOOP: c->opt_cd_count = value; return 0;
        case RCPPML_OPT_CD_LMF_LANE_GROUPS: c->opt_lmf_lg = value; return 0;
        case RCPPML_OPT_CD_LMF_WAVES_PER_SIMD: c->opt_lmf_wps = value; return 0;
        case RCPPML_OPT_CD_NO_LMF: c->opt_cd_no_lmf = value; return 0;
        case RCPPML_OPT_IRLS_COLUMNS_PER_WAVE: c->opt_irls_cpw = value; return 0;
        default: rcppml_err() = "unknown option"; return 1;
    }
}
extern "C" int rcppml_hip_ctx_stats(rcppml_hip_ctx* c, int reset, unsigned long long* out4) {
    try {
        HIPCHK(hipStreamSynchronize(c->stream));
        if (out4) HIPCHK(hipMemcpy(out4, c->stats, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        if (reset) HIPCHK(hipMemset(c->stats, 0, 4 * sizeof(unsigned long long)));
        return 0;
    }
    RCPPML_CATCH_RET
}
extern "C" int rcppml_hip_ctx_cd_step_stats(rcppml_hip_ctx* c, int reset, unsigned long long* out2) {
    try {
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (out2) HIPCHK(hipMemcpy(out2, c->stats + 6, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        if (reset) HIPCHK(hipMemset(c->stats + 6, 0, 2 * sizeof(unsigned long long)));
        return 0;
    }
    RCPPML_CATCH_RET
}
extern "C" int rcppml_hip_ctx_irls_stats(rcppml_hip_ctx* c, int reset, unsigned long long* out2) {
    try {
        HIPCHK(hipStreamSynchronize(c->stream));
        if (out2) HIPCHK(hipMemcpy(out2, c->stats + 4, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        if (reset) HIPCHK(hipMemset(c->stats + 4, 0, 2 * sizeof(unsigned long long)));
        return 0;
    }
    RCPPML_CATCH_RET
}

// ----------------------------------------------------------------------------
// Gram
// ----------------------------------------------------------------------------
template <class T> static int gram_kp(int k);
template <> int gram_kp<float>(int k) { return ((k + 31) / 32) * 32; }
template <> int gram_kp<double>(int k) { return ((k + 15) / 16) * 16; }

// per-block partial tiles of F F^T into the context's scratch; returns the table, *nblk_out tiles of KP x KP
template <class T>
static T* gram_partials(rcppml_hip_ctx* c, const T* F, int k, int64_t r, int* nblk_out, int* KP_out) {
    const int KP = gram_kp<T>(k);
    if (KP > 256) throw std::runtime_error("gram: k > 256 not supported");
    // number of blocks: enough waves to fill the chip, each wave >= 64 K-steps
    const int64_t step = std::is_same<T, float>::value ? 2 : 4;
    // enough waves to fill the chip; each wave >= 32 K-steps (more, smaller waves: the loop is latency-bound); the cap
    // bounds the partial-tile traffic (nblk * KP^2 values written and re-read by gram_finalize)
    int64_t nblk = (r / step + 4 * 32 - 1) / (4 * 32);
    if (nblk < 1) nblk = 1;
    if (nblk > 2 * (int64_t)c->num_cu) nblk = 2 * c->num_cu;
    T* partial = static_cast<T*>(c->scratch(WS_GRAM, (size_t)nblk * KP * KP * sizeof(T)));
    if constexpr (std::is_same<T, float>::value) {
        const int tt = KP / 32;
        dim3 grid((unsigned)nblk, tt), block(256);
        const bool vl = (tt == 2 || tt == 4) && k % tt == 0 && reinterpret_cast<uintptr_t>(F) % (4 * tt) == 0;
        switch (tt) {
            case 1: hipLaunchKernelGGL((gram_partial_f32<1, false, 8>), grid, block, 0, c->stream, F, k, r, partial); break;
            case 2:
                if (vl) hipLaunchKernelGGL((gram_partial_f32_k64<8>), dim3((unsigned)nblk), block, 0, c->stream, F, k, r, partial);   // all four tiles per block: F read once
                else hipLaunchKernelGGL((gram_partial_f32<2, false, 8>), grid, block, 0, c->stream, F, k, r, partial);
                break;
            case 3: hipLaunchKernelGGL((gram_partial_f32<3, false, 4>), grid, block, 0, c->stream, F, k, r, partial); break;
            case 4:
                if (vl) hipLaunchKernelGGL((gram_partial_f32<4, true, 4>), grid, block, 0, c->stream, F, k, r, partial);
                else hipLaunchKernelGGL((gram_partial_f32<4, false, 4>), grid, block, 0, c->stream, F, k, r, partial);
                break;
            case 5: hipLaunchKernelGGL((gram_partial_f32<5, false, 4>), grid, block, 0, c->stream, F, k, r, partial); break;
            case 6: hipLaunchKernelGGL((gram_partial_f32<6, false, 4>), grid, block, 0, c->stream, F, k, r, partial); break;
            case 7: hipLaunchKernelGGL((gram_partial_f32<7, false, 2>), grid, block, 0, c->stream, F, k, r, partial); break;
            default: hipLaunchKernelGGL((gram_partial_f32<8, false, 2>), grid, block, 0, c->stream, F, k, r, partial); break;
        }
    } else {
        const int tt = KP / 16;
        dim3 grid((unsigned)nblk, tt), block(256);
#define GRAM64_CASE(N) case N: hipLaunchKernelGGL(gram_partial_f64<N>, grid, block, 0, c->stream, F, k, r, partial); break;
        switch (tt) {
            GRAM64_CASE(1) GRAM64_CASE(2) GRAM64_CASE(3) GRAM64_CASE(4) GRAM64_CASE(5) GRAM64_CASE(6)
            GRAM64_CASE(7) GRAM64_CASE(8) GRAM64_CASE(9) GRAM64_CASE(10) GRAM64_CASE(11) GRAM64_CASE(12)
            GRAM64_CASE(13) GRAM64_CASE(14) GRAM64_CASE(15)
            default: hipLaunchKernelGGL(gram_partial_f64<16>, grid, block, 0, c->stream, F, k, r, partial); break;
        }
#undef GRAM64_CASE
    }
    HIPCHK(hipGetLastError());
    *nblk_out = (int)nblk;
    *KP_out = KP;
    return partial;
}
template <class T>
static void gram_impl(rcppml_hip_ctx* c, const T* F, int k, int64_t r, T eps, T l2, T* G) {
    int nblk = 0, KP = 0;
    const T* partial = gram_partials<T>(c, F, k, r, &nblk, &KP);
    hipLaunchKernelGGL(gram_finalize<T>, dim3((KP * KP + 7) / 8), dim3(256), 0, c->stream, partial, nblk, KP, k, eps, l2, G);
    HIPCHK(hipGetLastError());
}
// Gram of W_T (+ eps) into G_wt and the MSE loss by the Gram trick, three launches instead of four: the cross-term partials share the
// launch of the Gram's final sum (kernels_tail.hip.h) -- the results of rcppml_hip_gram + rcppml_hip_loss_mse bit for bit
template <class T>
static void gram_loss_mse_impl(rcppml_hip_ctx* c, const T* W_T, int k, int64_t m, T eps, const double* trAtA, const T* d, const T* B_w,
                               const T* G_saved, T* G_wt, double* out) {
    int nblk = 0, KP = 0;
    const T* partial = gram_partials<T>(c, W_T, k, m, &nblk, &KP);
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
extern "C" int rcppml_hip_gram(rcppml_hip_ctx* c, int dtype, const void* F, int k, int64_t r, double eps,
                               double l2, void* G) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (dtype == RCPPML_F32) gram_impl<float>(c, (const float*)F, k, r, (float)eps, (float)l2, (float*)G);
        else gram_impl<double>(c, (const double*)F, k, r, eps, l2, (double*)G);
        return 0;
    }
    RCPPML_CATCH_RET
}

