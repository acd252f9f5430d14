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
    constexpr int U = 8;
    const int64_t nblk = (ncols + 3) / 4;
    hipLaunchKernelGGL((rhs_kernel<T, VEC, LPN, U>), dim3((unsigned)nblk), dim3(256), 0, c->stream, cp, ri,
                       vals, ncols, F, k, B);
    HIPCHK(hipGetLastError());
}
template <class T>
static void rhs_impl(rcppml_hip_ctx* c, const int* cp, const int* ri, const T* vals, int64_t ncols,
                     const T* F, int k, T* B) {
    if (ncols <= 0) return;
    if (ncols > (int64_t)4 * 0x7fffffff) throw std::runtime_error("rhs: too many columns");
    constexpr int VMAX = 16 / sizeof(T);   // 16-byte loads
    const bool aligned = (reinterpret_cast<uintptr_t>(F) % 16 == 0) && (reinterpret_cast<uintptr_t>(B) % 16 == 0);
    if (k % VMAX == 0 && aligned && k / VMAX <= 64) {
        const int lpn = next_pow2(k / VMAX);
#define RHS_CASE(L) case L: rhs_launch<T, VMAX, L>(c, cp, ri, vals, ncols, F, k, B); break;
        switch (lpn) { RHS_CASE(1) RHS_CASE(2) RHS_CASE(4) RHS_CASE(8) RHS_CASE(16) RHS_CASE(32) RHS_CASE(64) }
#undef RHS_CASE
    } else {
        if (k > 64) throw std::runtime_error("rhs: k > 64 requires k % (16/sizeof(T)) == 0");
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

