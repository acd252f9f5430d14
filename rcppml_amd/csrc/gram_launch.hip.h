// gram_launch.hip.h -- launch of the Gram's per-block partial tiles (shared by ops_gram.hip and ops_tail.hip)
#pragma once
#include "common.hip.h"
#include "kernels.hip.h"

using namespace rk;
template <class T> inline int gram_kp(int k);
template <> inline int gram_kp<float>(int k) { return ((k + 31) / 32) * 32; }
template <> inline int gram_kp<double>(int k) { return ((k + 15) / 16) * 16; }

// per-block partial tiles of F F^T into the context's scratch; returns the table, *nblk_out tiles of KP x KP
template <class T>
inline T* gram_partials(rcppml_hip_ctx* c, const T* F, int k, int64_t r, int* nblk_out, int* KP_out) {
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
