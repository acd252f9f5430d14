// ops_gram.hip -- context + Gram (device-level C ABI, include/rcppml_gpu.h layer 2)
#include "common.hip.h"
#include "kernels.hip.h"
#include "gram_launch.hip.h"

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
        HIPCHK(hipMalloc(&c->stats, 12 * sizeof(unsigned long long)));
        HIPCHK(hipMemset(c->stats, 0, 12 * sizeof(unsigned long long)));
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
        case RCPPML_OPT_SMALL_GIVE_UP: c->opt_small_give_up = value; return 0;
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

// [8]: CD sweeps executed inside the IRLS half-updates (per column and pass, until the column's fixed point; counted with the passes)
extern "C" int rcppml_hip_ctx_irls_sweep_stats(rcppml_hip_ctx* c, int reset, unsigned long long* out1) {
    try {
        HIPCHK(hipStreamSynchronize(c->stream));
        if (out1) HIPCHK(hipMemcpy(out1, c->stats + 8, sizeof(unsigned long long), hipMemcpyDeviceToHost));
        if (reset) HIPCHK(hipMemset(c->stats + 8, 0, sizeof(unsigned long long)));
        return 0;
    }
    RCPPML_CATCH_RET
}

// ----------------------------------------------------------------------------
// Gram
// ----------------------------------------------------------------------------
template <class T>
static void gram_impl(rcppml_hip_ctx* c, const T* F, int k, int64_t r, T eps, T l2, T* G) {
    int nblk = 0, KP = 0;
    const T* partial = gram_partials<T>(c, F, k, r, &nblk, &KP);
    hipLaunchKernelGGL(gram_finalize<T>, dim3((KP * KP + 7) / 8), dim3(256), 0, c->stream, partial, nblk, KP, k, eps, l2, G);
    HIPCHK(hipGetLastError());
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

