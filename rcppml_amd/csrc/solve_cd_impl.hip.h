// solve_cd_impl.hip.h -- CD solve launch logic (instantiated per dtype in ops_cd_f32.hip / ops_cd_f64.hip)
#pragma once
#include <cmath>
#include "solve_common.hip.h"
#include "kernels_cd_mfma.hip.h"
#include "kernels_cd_mfma64.hip.h"
// ----------------------------------------------------------------------------
// CD solve
// ----------------------------------------------------------------------------
template <class T, int KP>
static void cd_lane_launch(rcppml_hip_ctx* c, const T* Gp, const T* invd, const T* B, T* X, int k,
                           int64_t ncols, T l1_pre, int warm, int zero_init, T l1_cd, T l2_cd, int nonneg,
                           int maxit, T tol, T ub_cd, T ub_post, int* sweeps, const int* order) {
    constexpr bool EXACT = std::is_same<T, double>::value;
    const int64_t nblk = (ncols + 63) / 64;
    hipLaunchKernelGGL((cd_lane_kernel<T, KP, EXACT>), dim3((unsigned)nblk), dim3(64), 0, c->stream, Gp, invd,
                       B, X, k, ncols, l1_pre, warm, zero_init, l1_cd, l2_cd, nonneg, maxit, tol, ub_cd, ub_post, sweeps, order);
    HIPCHK(hipGetLastError());
}
template <class T, int KP, bool GLDS = true>
static void cd_wave_launch(rcppml_hip_ctx* c, const T* Gp, const T* invd, const T* B, T* X, int k,
                           int64_t ncols, T l1_pre, int warm, int zero_init, T l1_cd, T l2_cd, int nonneg,
                           int maxit, T tol, T ub_cd, T ub_post, int* sweeps, const int* order) {
    constexpr bool EXACT = std::is_same<T, double>::value;
    const size_t smem = GLDS ? (size_t)KP * KP * sizeof(T) : 0;
    auto kern = cd_wave_kernel<T, KP, EXACT, GLDS>;
    static DynSmemOnce once;
    once.ensure(reinterpret_cast<const void*>(kern), smem, c->device);
    // persistent blocks: as many 256-thread blocks per CU as LDS allows (<= 8), capped by the work
    int per_cu = (int)((160 * 1024) / (smem + 256));
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    int64_t nblk = (int64_t)c->num_cu * per_cu;
    const int64_t need = (ncols + 3) / 4;
    if (nblk > need) nblk = need;
    if (nblk < 1) nblk = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), smem, c->stream, Gp, invd, B, X, k, ncols, l1_pre,
                       warm, zero_init, l1_cd, l2_cd, nonneg, maxit, tol, ub_cd, ub_post, sweeps, order);
    HIPCHK(hipGetLastError());
}


// Lane-group variant: LPC lanes per column, 4 waves per block sharing the LDS copy of G.
template <class T, int KP, int LPC>
static void cd_group_launch(rcppml_hip_ctx* c, const T* Gp, const T* invd, const T* B, T* X, int k, int64_t ncols,
                            T l1_pre, int warm, int zero_init, T l1_cd, T l2_cd, int nonneg, int maxit, T tol, T ub_cd,
                            T ub_post, int* sweeps, const int* order) {
    constexpr bool EXACT = std::is_same<T, double>::value;
    const size_t smem = ((size_t)KP * KP + 2 * KP) * sizeof(T);
    auto kern = cd_group_kernel<T, KP, LPC, EXACT>;
    static DynSmemOnce once;
    once.ensure(reinterpret_cast<const void*>(kern), smem, c->device);
    const int64_t per_block = (int64_t)4 * (64 / LPC);
    const int64_t nblk = (ncols + per_block - 1) / per_block;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), smem, c->stream, Gp, invd, B, X, k, ncols, l1_pre, warm,
                       zero_init, l1_cd, l2_cd, nonneg, maxit, tol, ub_cd, ub_post, sweeps, order, c->stats);
    HIPCHK(hipGetLastError());
}

// MFMA variant (fp32, k <= 128): rank-1 residual updates on the matrix cores, 32*CT columns per wave.
template <int RT, int CT>
static void cd_mfma_launch(rcppml_hip_ctx* c, const float* G, const float* /*unused*/, const float* B, float* X, int k,
                           int64_t ncols, float l1_pre, int warm, int zero_init, float l1_cd, float l2_cd, int nonneg,
                           int maxit, float tol, float ub_cd, float ub_post, int* sweeps, const int* order) {
    size_t smem = cd_mfma_lds_bytes(RT);          // operand image of the k x k Gram, KP = 32 RT rows
    const int64_t per_block = 4 * 32 * CT;          // 4 waves per block
    const int64_t nblk = (ncols + per_block - 1) / per_block;
    const bool simple = nonneg && ub_cd <= 0.f && l1_cd == 0.f && l2_cd == 0.f;
    // Residency cap.  A SIMD's waves share its (non-overlapping) VALU + f32-MFMA issue, so with all blocks resident
    // the kernel takes as long as the fullest CU: 782 blocks on 256 CUs = 4 blocks on 14 CUs, 3 on the rest, i.e.
    // 4/3.05 of the balanced time.  Capping residency at floor(blocks per CU) (by asking for more LDS than a further
    // block would leave) makes the surplus blocks -- the cheapest ones under the sweep-sorted order -- start when
    // the first blocks retire.  RCPPML_GPU_CD_CAP overrides in experiment builds (0 = no cap).
    {
        const double per_cu = (double)nblk / (double)(c->num_cu > 0 ? c->num_cu : 256);
        int cap = 0;
        if (per_cu > 1.0 && per_cu < 4.0 && per_cu - std::floor(per_cu) < 0.5) cap = (int)std::floor(per_cu);
        if (const char* e = exp_env("RCPPML_GPU_CD_CAP")) cap = atoi(e);
        if (cap > 0) {
            const size_t lds_cu = 160 * 1024;
            const size_t want = lds_cu / (size_t)(cap + 1) + 1024;      // cap + 1 blocks no longer fit
            if (want > smem && want <= lds_cu / (size_t)cap) smem = want;
        }
    }
    static DynSmemOnce once_simple, once_general;
    if (simple) once_simple.ensure(reinterpret_cast<const void*>(&cd_mfma_kernel<RT, CT, true>), smem, c->device);
    else once_general.ensure(reinterpret_cast<const void*>(&cd_mfma_kernel<RT, CT, false>), smem, c->device);
    if (simple)
        hipLaunchKernelGGL((cd_mfma_kernel<RT, CT, true>), dim3((unsigned)nblk), dim3(256), smem, c->stream, G, B, X,
                           k, ncols, l1_pre, warm, zero_init, l1_cd, l2_cd, nonneg, maxit, tol, ub_cd, ub_post, sweeps, order, c->stats);
    else
        hipLaunchKernelGGL((cd_mfma_kernel<RT, CT, false>), dim3((unsigned)nblk), dim3(256), smem, c->stream, G, B, X,
                           k, ncols, l1_pre, warm, zero_init, l1_cd, l2_cd, nonneg, maxit, tol, ub_cd, ub_post, sweeps, order, c->stats);
    HIPCHK(hipGetLastError());
}


// Lane = column MFMA variant (fp32, k <= 64, SIMPLE steps): ops_cd_lmf.hip (its own translation unit and flags)
void rcppml_cd_lmf_dispatch(rcppml_hip_ctx* c, const float* G, const float* B, float* X, int k, int64_t ncols, float l1_pre,
                            int warm, int zero_init, int maxit, float tol, float ub_post, int* sweeps, const int* order);

// 16-column MFMA variant: v_mfma_f64_16x16x4_f64 (k <= 128) / v_mfma_f32_16x16x4_f32 (k <= 64), four coordinates per instruction.
template <class T, int NT>
static void cd_mfma64_launch(rcppml_hip_ctx* c, const T* G, const T* /*unused*/, const T* B, T* X, int k,
                             int64_t ncols, T l1_pre, int warm, int zero_init, T l1_cd, T l2_cd, int nonneg,
                             int maxit, T tol, T ub_cd, T ub_post, int* sweeps, const int* order) {
    constexpr int KP = 16 * NT;
    const size_t smem = ((size_t)KP * KP + 4 * KP + (sizeof(T) == 8 ? KP : 0)) * sizeof(T);      // fp64: + the diagonal itself
    const int64_t nblk = (ncols + 63) / 64;          // 4 waves x 16 columns per block
    const bool simple = nonneg && ub_cd <= T(0) && l1_cd == T(0) && l2_cd == T(0);
    static DynSmemOnce once_simple, once_general;    // fp64 above k = 64: the LDS copy of G passes 64 KiB (128 KiB at KP = 128)
    if (simple) once_simple.ensure(reinterpret_cast<const void*>(&cd_mfma64_kernel<T, NT, true>), smem, c->device);
    else once_general.ensure(reinterpret_cast<const void*>(&cd_mfma64_kernel<T, NT, false>), smem, c->device);
    if (simple)
        hipLaunchKernelGGL((cd_mfma64_kernel<T, NT, true>), dim3((unsigned)nblk), dim3(256), smem, c->stream, G, B, X, k,
                           ncols, l1_pre, warm, zero_init, l1_cd, l2_cd, nonneg, maxit, tol, ub_cd, ub_post, sweeps, order, c->stats);
    else
        hipLaunchKernelGGL((cd_mfma64_kernel<T, NT, false>), dim3((unsigned)nblk), dim3(256), smem, c->stream, G, B, X, k,
                           ncols, l1_pre, warm, zero_init, l1_cd, l2_cd, nonneg, maxit, tol, ub_cd, ub_post, sweeps, order, c->stats);
    HIPCHK(hipGetLastError());
}

// LPC choice (measured on MI355X, k = 64, 20k..100k columns: 32 fp32 rows per lane beat 16 by 4-15 %, and for
// fp64 16 rows per lane are as good as 32 at half the registers): fp32 -> KP/32 lanes per column, fp64 -> KP/16,
// clamped to {1, 2, 4}.  RCPPML_GPU_CD_LPC overrides (experiments).
template <class T>
static int pick_lpc(int KP) {
    const int ev = 16 / (int)sizeof(T);
    const char* e = exp_env("RCPPML_GPU_CD_LPC");
    int lpc = e ? atoi(e) : (std::is_same<T, float>::value ? KP / 32 : KP / 16);
    if (lpc < 1) lpc = 1;
    if (lpc > 4) lpc = 4;
    while (lpc < 4 && KP / lpc > 32) lpc *= 2;          // at most 32 rows per lane
    while (lpc > 1 && (KP / lpc) % ev != 0) lpc /= 2;   // whole 16-byte vectors per lane
    return lpc;
}

template <class T>
static void solve_cd_impl(rcppml_hip_ctx* c, const T* G, const T* B, T* X, int k, int64_t ncols, T l1_pre,
                          int warm, int zero_init, T l1_cd, T l2_cd, int nonneg, int maxit, T tol, T ub_cd,
                          T ub_post, int variant, int* sweeps, const int* order) {
    if (ncols <= 0) return;
    if (k < 1 || k > 256) throw std::runtime_error("solve_cd: k must be in [1,256]");
    if (k > 128) {
        // general-rank path: one wavefront per column, four coordinates per lane, the Gram read from L2 (cd_wave_kernel
        // with GLDS = false); every `variant` lands here
        T *Gp, *invd;
        pad_impl<T>(c, G, k, 256, &Gp, &invd);
        cd_wave_launch<T, 256, false>(c, Gp, invd, B, X, k, ncols, l1_pre, warm, zero_init, l1_cd, l2_cd, nonneg, maxit, tol, ub_cd,
                                      ub_post, sweeps, order);
        return;
    }
    int KP = solve_kp(k);
    // fp32 NMF half-updates (non-negativity only, k <= 64) can run on the lane = column MFMA kernel (kernels_cd_lmf.hip.h).
    // Measured (tools/cd_bench.py, tools/cd_c2_bench.py): it wins by 25-35 % for k <= 32 once every SIMD gets a 64-column wave
    // (100 000 x k=32: 0.108 vs 0.165 ms per 20 sweeps), and ties or loses at 32 < k <= 64 on C2-sized sides (the 64-row update
    // is matrix-pipe bound either way and 100 000 columns do not fill 2 x 1024 waves of 64) -- so AUTO takes it for k <= 32 only.
    const bool lmf_ok = std::is_same<T, float>::value && k <= 64 && nonneg && ub_cd <= T(0) && l1_cd == T(0) && l2_cd == T(0) && maxit >= 1;
    if (variant == RCPPML_CD_LMF && !lmf_ok) variant = RCPPML_CD_AUTO;
    if (variant == RCPPML_CD_AUTO && lmf_ok && k <= 32 && ncols >= (int64_t)256 * c->num_cu && !c->opt_cd_no_lmf &&
        !exp_env("RCPPML_GPU_CD_VARIANT"))
        variant = RCPPML_CD_LMF;
    if (variant == RCPPML_CD_LMF) {
        if constexpr (std::is_same<T, float>::value)
            rcppml_cd_lmf_dispatch(c, G, B, X, k, ncols, l1_pre, warm, zero_init, maxit, tol, ub_post, sweeps, order);
        return;
    }
    // Small sides (at most ~1.5 columns per SIMD): one wavefront per column with static coordinate sweeps -- every wave runs alone
    // there and the solve lasts as long as one column's dependent chain (cd_wave_static_kernel, kernels.hip.h)
    if (variant == RCPPML_CD_AUTO && k <= 64 && nonneg && ub_cd <= T(0) && l1_cd == T(0) && l2_cd == T(0) && maxit >= 1 &&
        ncols <= (int64_t)6 * (c->num_cu > 0 ? c->num_cu : 256) && !exp_env("RCPPML_GPU_CD_VARIANT")) {
        const unsigned nblk = (unsigned)((ncols + 3) / 4);
        if (k <= 16)
            hipLaunchKernelGGL((cd_wave_static_kernel<T, 16>), dim3(nblk), dim3(256), 0, c->stream, G, B, X, k, ncols, l1_pre, warm, zero_init,
                               maxit, tol, ub_post, sweeps, c->stats);
        else if (k <= 32)
            hipLaunchKernelGGL((cd_wave_static_kernel<T, 32>), dim3(nblk), dim3(256), 0, c->stream, G, B, X, k, ncols, l1_pre, warm, zero_init,
                               maxit, tol, ub_post, sweeps, c->stats);
        else
            hipLaunchKernelGGL((cd_wave_static_kernel<T, 64>), dim3(nblk), dim3(256), 0, c->stream, G, B, X, k, ncols, l1_pre, warm, zero_init,
                               maxit, tol, ub_post, sweeps, c->stats);
        HIPCHK(hipGetLastError());
        return;
    }
    if (variant == RCPPML_CD_AUTO) {
        const char* e = exp_env("RCPPML_GPU_CD_VARIANT");
        if (e && !strcmp(e, "lane")) variant = RCPPML_CD_LANE;
        else if (e && !strcmp(e, "wave")) variant = RCPPML_CD_WAVE;
        else if (e && !strcmp(e, "group")) variant = RCPPML_CD_GROUP;
        else if (e && !strcmp(e, "mfma")) variant = RCPPML_CD_MFMA;
        else if (e && !strcmp(e, "mfma16")) variant = RCPPML_CD_MFMA16;
        else variant = RCPPML_CD_MFMA;          // fp32: 32-column tiles; fp64: 16-column tiles (below)
        // fp32: 32-column tiles leave SIMDs idle when there are fewer tiles than SIMDs (C2's W side: 20 000 columns =
        // 625 tiles on 1024 SIMDs); 16-column tiles double the wavefronts there (RCPPML_GPU_CD_SMALL16=0 disables)
        if (variant == RCPPML_CD_MFMA && std::is_same<T, float>::value && k <= 64 && !e) {
            static int small16 = -1;
            if (small16 < 0) small16 = exp_flag("RCPPML_GPU_CD_SMALL16", "0") ? 0 : 1;
            if (small16 && (ncols + 31) / 32 < (int64_t)4 * c->num_cu) variant = RCPPML_CD_MFMA16;
        }
    }
    if (variant == RCPPML_CD_MFMA && !std::is_same<T, float>::value) variant = RCPPML_CD_MFMA16;      // fp64 MFMA = the 16-column form
    if (variant == RCPPML_CD_MFMA16 && k > 64 && std::is_same<T, float>::value) variant = RCPPML_CD_MFMA;   // fp32 16-column tiles: k <= 64
    if (variant == RCPPML_CD_MFMA) KP = 32 * ((k + 31) / 32);
    if (variant == RCPPML_CD_MFMA16) KP = 16 * ((k + 15) / 16);
    // register-resident lane kernel (SGPR-fed): fp32 up to KP=64, fp64 up to KP=32 without spilling
    const int lane_max = std::is_same<T, float>::value ? 64 : 32;
    if (variant == RCPPML_CD_LANE && KP > lane_max) variant = RCPPML_CD_GROUP;
    if (variant != RCPPML_CD_LANE && variant != RCPPML_CD_WAVE && variant != RCPPML_CD_MFMA && variant != RCPPML_CD_MFMA16) variant = RCPPML_CD_GROUP;
    if (variant == RCPPML_CD_WAVE && KP < 64) KP = 64;   // the wave variant pads to a full 64-lane slab
    // the MFMA variants read the k x k Gram themselves (their prep kernels pad it); the others take pad_gram's copy
    T *Gp = const_cast<T*>(G), *invd = nullptr;
    if (variant != RCPPML_CD_MFMA && variant != RCPPML_CD_MFMA16) pad_impl<T>(c, G, k, KP, &Gp, &invd);
#define CD_ARGS c, Gp, invd, B, X, k, ncols, l1_pre, warm, zero_init, l1_cd, l2_cd, nonneg, maxit, tol, ub_cd, ub_post, sweeps, order
    if (variant == RCPPML_CD_LANE) {
        switch (KP) {
            case 16: cd_lane_launch<T, 16>(CD_ARGS); break;
            case 32: cd_lane_launch<T, 32>(CD_ARGS); break;
            default:
                if constexpr (std::is_same<T, float>::value) cd_lane_launch<T, 64>(CD_ARGS);
                break;
        }
    } else if (variant == RCPPML_CD_MFMA) {
        if constexpr (std::is_same<T, float>::value) {
            switch (KP) {
                case 32: cd_mfma_launch<1, 1>(CD_ARGS); break;
                case 64: cd_mfma_launch<2, 1>(CD_ARGS); break;
                case 96: cd_mfma_launch<3, 1>(CD_ARGS); break;
                default: cd_mfma_launch<4, 1>(CD_ARGS); break;
            }
        }
    } else if (variant == RCPPML_CD_MFMA16) {
        switch (KP) {
            case 16: cd_mfma64_launch<T, 1>(CD_ARGS); break;
            case 32: cd_mfma64_launch<T, 2>(CD_ARGS); break;
            case 48: cd_mfma64_launch<T, 3>(CD_ARGS); break;
            case 64: cd_mfma64_launch<T, 4>(CD_ARGS); break;
            default:
                if constexpr (std::is_same<T, double>::value) {        // fp64, 64 < k <= 128 (C4 in parity mode)
                    switch (KP) {
                        case 80: cd_mfma64_launch<T, 5>(CD_ARGS); break;
                        case 96: cd_mfma64_launch<T, 6>(CD_ARGS); break;
                        case 112: cd_mfma64_launch<T, 7>(CD_ARGS); break;
                        default: cd_mfma64_launch<T, 8>(CD_ARGS); break;
                    }
                } else cd_mfma64_launch<T, 4>(CD_ARGS);
                break;
        }
    } else if (variant == RCPPML_CD_GROUP) {
        const int lpc = pick_lpc<T>(KP);
#define GROUP_CASE(K_, L_) if (KP == K_ && lpc == L_) { cd_group_launch<T, K_, L_>(CD_ARGS); launched = true; }
        bool launched = false;
        GROUP_CASE(16, 1) GROUP_CASE(16, 2) GROUP_CASE(16, 4)
        GROUP_CASE(32, 1) GROUP_CASE(32, 2) GROUP_CASE(32, 4)
        GROUP_CASE(64, 2) GROUP_CASE(64, 4)
        GROUP_CASE(128, 4)
#undef GROUP_CASE
        if (!launched) throw std::runtime_error("solve_cd: no lane-group configuration for this rank");
    } else {
        if (KP == 64) cd_wave_launch<T, 64>(CD_ARGS);
        else cd_wave_launch<T, 128>(CD_ARGS);
    }
#undef CD_ARGS
}
