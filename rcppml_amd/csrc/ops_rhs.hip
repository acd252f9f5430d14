// ops_rhs.hip -- sparse RHS products (device-level C ABI)
#include "common.hip.h"
#include "kernels.hip.h"

using namespace rk;
// ----------------------------------------------------------------------------
// RHS
// ----------------------------------------------------------------------------
static int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

template <class T, int VEC, int LPN>
static void rhs_launch(rcppml_hip_ctx* c, const int* cp, const int* ri, const T* vals, int64_t ncols,
                       const T* F, int k, T* B) {
    const int64_t nblk = (ncols + 3) / 4;
    dim3 grid((unsigned)nblk), block(256);
#ifdef RCPPML_EXPERIMENTS
    // probes only (-DRCPPML_EXPERIMENTS): RCPPML_GPU_RHS_VARIANT = u16 | u4 | u2 | group | stage, RCPPML_GPU_RHS_STAGE_U
    static int mode = -1, staged = -1;
    if (mode < 0) {
        const char* e = getenv("RCPPML_GPU_RHS_VARIANT");
        mode = !e ? 0 : (!strcmp(e, "u16") ? 1 : (!strcmp(e, "u4") ? 2 : (!strcmp(e, "u2") ? 3 : 0)));
        staged = (e && *e) ? (!strcmp(e, "stage") ? 1 : 0) : 1;
    }
    if (!staged || LPN < 8) {
        switch (mode) {
            case 1: hipLaunchKernelGGL((rhs_kernel<T, VEC, LPN, 16, false>), grid, block, 0, c->stream, cp, ri, vals, ncols, F, k, B); break;
            case 2: hipLaunchKernelGGL((rhs_kernel<T, VEC, LPN, 4, false>), grid, block, 0, c->stream, cp, ri, vals, ncols, F, k, B); break;
            case 3: hipLaunchKernelGGL((rhs_kernel<T, VEC, LPN, 2, false>), grid, block, 0, c->stream, cp, ri, vals, ncols, F, k, B); break;
            default: hipLaunchKernelGGL((rhs_kernel<T, VEC, LPN, 8, false>), grid, block, 0, c->stream, cp, ri, vals, ncols, F, k, B); break;
        }
        HIPCHK(hipGetLastError());
        return;
    }
#endif
    if constexpr (LPN >= 8) {
        // staged indices: one coalesced load per 64 (row, value) pairs, handed to the lane groups through ds_bpermute;
        // 16 gathers in flight per lane group where a row spans 16 lanes (measured on C2 fp32: rhs_H 0.41 -> 0.29 ms,
        // rhs_W 0.44 -> 0.31 ms vs the group-uniform index loads of rhs_kernel; with 8 in flight the W side, which gathers
        // from the 25 MB factor, is 20 % SLOWER than rhs_kernel)
        constexpr int SU = LPN >= 16 ? 16 : 8;
        hipLaunchKernelGGL((rhs_stage_kernel<T, VEC, LPN, SU>), grid, block, 0, c->stream, cp, ri, vals, ncols, F, k, B);
    } else {
        hipLaunchKernelGGL((rhs_kernel<T, VEC, LPN, 8, false>), grid, block, 0, c->stream, cp, ri, vals, ncols, F, k, B);
    }
    HIPCHK(hipGetLastError());
}
template <class T>
static void rhs_impl(rcppml_hip_ctx* c, const int* cp, const int* ri, const T* vals, int64_t ncols,
                     const T* F, int k, T* B) {
    if (ncols <= 0) return;
    if (ncols > (int64_t)4 * 0x7fffffff) throw std::runtime_error("rhs: too many columns");
#ifdef RCPPML_EXPERIMENTS
    // RCPPML_GPU_RHS_VARIANT=wave: scalarised-index kernel (one nonzero per wave instruction, reference summation
    // order).  Measured no faster than the lane-group kernel on MI355X (H 0.475 / W 0.552 ms vs 0.468 / 0.403 ms on C2).
    {
        static int wave_mode = -1;
        if (wave_mode < 0) { const char* e = getenv("RCPPML_GPU_RHS_VARIANT"); wave_mode = (e && !strcmp(e, "wave")) ? 1 : 0; }
        const int64_t nblk = (ncols + 3) / 4;
        const bool al8 = (reinterpret_cast<uintptr_t>(F) % 16 == 0) && (reinterpret_cast<uintptr_t>(B) % 16 == 0);
        if (wave_mode == 1 && k > 32 && k <= 64) {
            hipLaunchKernelGGL((rhs_wave_kernel<T, 1, 8>), dim3((unsigned)nblk), dim3(256), 0, c->stream, cp, ri, vals, ncols, F, k, B);
            HIPCHK(hipGetLastError());
            return;
        }
        if (wave_mode == 1 && k > 64 && k <= 128 && k % 2 == 0 && al8) {
            hipLaunchKernelGGL((rhs_wave_kernel<T, 2, 8>), dim3((unsigned)nblk), dim3(256), 0, c->stream, cp, ri, vals, ncols, F, k, B);
            HIPCHK(hipGetLastError());
            return;
        }
    }
#endif
    constexpr int VMAX = 16 / sizeof(T);   // 16-byte loads
    const bool aligned = (reinterpret_cast<uintptr_t>(F) % 16 == 0) && (reinterpret_cast<uintptr_t>(B) % 16 == 0);
    if (k % VMAX == 0 && aligned && k / VMAX <= 64) {
        const int lpn = next_pow2(k / VMAX);
#define RHS_CASE(L) case L: rhs_launch<T, VMAX, L>(c, cp, ri, vals, ncols, F, k, B); break;
        switch (lpn) { RHS_CASE(1) RHS_CASE(2) RHS_CASE(4) RHS_CASE(8) RHS_CASE(16) RHS_CASE(32) RHS_CASE(64) }
#undef RHS_CASE
    } else {
        if (k > 64) {      // odd ranks above 64 (or unaligned factors): general one-wave-per-column kernel
            if (k > 256) throw std::runtime_error("rhs: k > 256 not supported");
            const int64_t nblk = (ncols + 3) / 4;
            if (k <= 128) hipLaunchKernelGGL((rhs_generic_kernel<T, 2>), dim3((unsigned)nblk), dim3(256), 0, c->stream, cp, ri, vals, ncols, F, k, B);
            else hipLaunchKernelGGL((rhs_generic_kernel<T, 4>), dim3((unsigned)nblk), dim3(256), 0, c->stream, cp, ri, vals, ncols, F, k, B);
            HIPCHK(hipGetLastError());
            return;
        }
        const int lpn = next_pow2(k);
#define RHS_CASE(L) case L: rhs_launch<T, 1, L>(c, cp, ri, vals, ncols, F, k, B); break;
        switch (lpn) { RHS_CASE(1) RHS_CASE(2) RHS_CASE(4) RHS_CASE(8) RHS_CASE(16) RHS_CASE(32) RHS_CASE(64) }
#undef RHS_CASE
    }
}
extern "C" int rcppml_hip_rhs(rcppml_hip_ctx* c, int dtype, const int* col_ptr, const int* row_idx,
                              const void* values, int64_t ncols, const void* F, int k, void* B) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (dtype == RCPPML_F32) rhs_impl<float>(c, col_ptr, row_idx, (const float*)values, ncols, (const float*)F, k, (float*)B);
        else rhs_impl<double>(c, col_ptr, row_idx, (const double*)values, ncols, (const double*)F, k, (double*)B);
        return 0;
    }
    RCPPML_CATCH_RET
}

