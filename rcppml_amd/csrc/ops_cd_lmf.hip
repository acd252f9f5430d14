// ops_cd_lmf.hip -- launch logic of the lane = column MFMA coordinate-descent kernel (kernels_cd_lmf.hip.h).  Its own
// translation unit: built with -fno-slp-vectorize (packed f32 VALU next to MFMAs costs more than it saves on gfx950) and
// with the accumulator tiles in VGPR form (single elements are read every coordinate).
#include "solve_common.hip.h"
#include "kernels_cd_lmf.hip.h"

// Lane = column MFMA variant (fp32, k <= 64, SIMPLE steps): persistent waves, LG lane groups per column.
template <int KP, int LG>
static void cd_lmf_launch(rcppml_hip_ctx* c, const float* G, const float* B, float* X, int k, int64_t ncols, float l1_pre,
                          int warm, int zero_init, int maxit, float tol, float ub_post, int* sweeps, const int* order, int wps) {
    typedef LmfGeom<KP, LG> Ge;
    float2* img = static_cast<float2*>(c->scratch(WS_MFMA, (size_t)KP * 64 * sizeof(float2)));
    hipLaunchKernelGGL((cd_lmf_prep_kernel<KP, LG>), dim3((KP * 64 + 255) / 256), dim3(256), 0, c->stream, G, k, img);
    HIPCHK(hipGetLastError());
    size_t smem = (size_t)KP * 64 * sizeof(float2);
    const int64_t waves = (ncols + Ge::CW - 1) / Ge::CW;
    int64_t nblk = (waves + 3) / 4;
    const int64_t cap = (int64_t)(c->num_cu > 0 ? c->num_cu : 256) * wps;
    if (wps < 8 && nblk > cap) {
        nblk = cap;
        // exactly wps blocks per CU: ask for so much LDS that one more block does not fit
        const size_t lds_cu = 160 * 1024;
        const size_t want = lds_cu / (size_t)(wps + 1) + 1024;
        if (want > smem && want <= lds_cu / (size_t)wps) smem = want;
    }
    const bool count = c->opt_cd_count != 0;
    static DynSmemOnce once_a, once_b;
    if (count) once_a.ensure(reinterpret_cast<const void*>(&cd_lmf_kernel<KP, LG, true>), smem, c->device);
    else once_b.ensure(reinterpret_cast<const void*>(&cd_lmf_kernel<KP, LG, false>), smem, c->device);
    if (count)
        hipLaunchKernelGGL((cd_lmf_kernel<KP, LG, true>), dim3((unsigned)nblk), dim3(256), smem, c->stream, img, B, X, k, ncols, l1_pre,
                           warm, zero_init, maxit, tol, ub_post, sweeps, order, c->stats);
    else
        hipLaunchKernelGGL((cd_lmf_kernel<KP, LG, false>), dim3((unsigned)nblk), dim3(256), smem, c->stream, img, B, X, k, ncols, l1_pre,
                           warm, zero_init, maxit, tol, ub_post, sweeps, order, c->stats);
    HIPCHK(hipGetLastError());
}
// Geometry (measured, tools/cd_bench.py / tools/cd_c2_bench.py): one lane group (64 columns per wave, no lane-group selects)
// and two resident waves per SIMD once there are >= 64 columns per SIMD; fewer columns -> more lane groups so that every SIMD
// still gets a wave.  RCPPML_OPT_CD_LMF_* override both.
void rcppml_cd_lmf_dispatch(rcppml_hip_ctx* c, const float* G, const float* B, float* X, int k, int64_t ncols, float l1_pre,
                            int warm, int zero_init, int maxit, float tol, float ub_post, int* sweeps, const int* order) {
    const int64_t simds = (int64_t)(c->num_cu > 0 ? c->num_cu : 256) * 4;
    int lg = ncols >= 64 * simds ? 1 : (ncols >= 32 * simds ? 2 : 4);
    if (c->opt_lmf_lg == 1 || c->opt_lmf_lg == 2 || c->opt_lmf_lg == 4) lg = c->opt_lmf_lg;
    if (const char* e = exp_env("RCPPML_GPU_LMF_LG")) lg = atoi(e);
    int wps = lg == 1 ? 2 : 3;
    if (c->opt_lmf_wps > 0) wps = c->opt_lmf_wps;
    if (const char* e = exp_env("RCPPML_GPU_LMF_WPS")) wps = atoi(e);
    if (wps > 4 && wps < 8) wps = 4;       // wps >= 8: no residency cap -- one wave per CW columns, all launched (tile mode)
#define LMF_ARGS c, G, B, X, k, ncols, l1_pre, warm, zero_init, maxit, tol, ub_post, sweeps, order, wps
    if (k <= 32) {
        if (lg == 1) cd_lmf_launch<32, 1>(LMF_ARGS); else if (lg == 2) cd_lmf_launch<32, 2>(LMF_ARGS); else cd_lmf_launch<32, 4>(LMF_ARGS);
    } else {
        if (lg == 1) cd_lmf_launch<64, 1>(LMF_ARGS); else if (lg == 2) cd_lmf_launch<64, 2>(LMF_ARGS); else cd_lmf_launch<64, 4>(LMF_ARGS);
    }
#undef LMF_ARGS
}
