// lmf_body_probe.hip -- per-coordinate cost of the lane = column CD sweep body (kernels_cd_lmf.hip.h) in isolation, with switches:
// LG (lane groups), PD (LDS prefetch distance), LDS on/off, BLGP on/off, tolerance term on/off, correction-mode select on/off.
// Prints wall ns per coordinate and SIMD for 1..4 resident waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
template <int CBSZ, int ABID, int BLGP> __device__ __forceinline__ f32x4 mf(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, CBSZ, ABID, BLGP);
}
template <int LG, int PD, bool LDS, bool BLGPON, bool TOL, bool CM>
__global__ __launch_bounds__(1024) void body(float* out, int nsweep, float s, int cmflag) {
    constexpr int KP = 64, CW = 64 / LG, NTL = KP / LG / 4, CBSZ = LG == 1 ? 4 : (LG == 2 ? 3 : 2);
    __shared__ float2 img[KP * 64];
    for (int e = threadIdx.x; e < KP * 64; e += blockDim.x) img[e] = make_float2(-1e-3f * (e & 63) - s, 0.01f + s);
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane / CW;
    f32x4 acc[NTL];
    float xn[NTL][4];
    for (int t = 0; t < NTL; ++t) for (int r = 0; r < 4; ++r) { acc[t][r] = 1.f + t + r + s; xn[t][r] = -0.5f - s; }
    bool ing[LG];
    for (int q = 0; q < LG; ++q) ing[q] = g == q;
    const bool cm = cmflag != 0 && lane == 77;
    const float keep = cm ? 0.f : 1.f;
    float2 ring[PD];
    for (int p = 0; p < PD; ++p) ring[p] = img[p * 64 + lane];
    float tsum = 0.f, areg = 0.f;
    for (int sw = 0; sw < nsweep; ++sw) {
        static_for<0, KP>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int gi = i % LG, sl = i / LG, t = sl / 4, r = sl % 4;
            constexpr int tn = (((i + 1) % KP) / LG) / 4;
            float2 gq;
            if constexpr (LDS) { gq = ring[i % PD]; ring[i % PD] = img[((i + PD) % KP) * 64 + lane]; }
            else gq = make_float2(-1e-3f * lane, 0.01f);
            const float xo = xn[t][r];
            const float diff = acc[t][r] * gq.y;
            float a = __builtin_fmaxf(diff, xo);
            if constexpr (CM) a = cm ? -xo : a;
            if constexpr (LG == 1) areg = a; else areg = ing[gi] ? a : areg;
            static_for<0, NTL>([&](auto uc) {
                constexpr int tt = (tn + decltype(uc)::value) % NTL;
                constexpr int bl = !BLGPON ? 0 : (LG == 1 ? 0 : (LG == 2 ? (gi == 0 ? 1 : 2) : 4 + gi));
                acc[tt] = mf<CBSZ, tt, bl>(gq.x, areg, acc[tt]);
            });
            if constexpr (gi == LG - 1) {
                const float xnew = __builtin_fmaf(-areg, keep, xo);
                xn[t][r] = xnew;
                if constexpr (TOL) tsum = __builtin_fmaf(__builtin_fabsf(areg), __builtin_amdgcn_rcpf(__builtin_fabsf(xnew) + 1e-15f), tsum);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    float sum = tsum;
    for (int t = 0; t < NTL; ++t) for (int r = 0; r < 4; ++r) sum += acc[t][r] + xn[t][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}
template <class K> static double run(K kern, int w, int nsweep) {
    float* d; hipMalloc(&d, 256 * 1024 * 4);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256 * w), 0, 0, d, nsweep, 0.f, 0);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256 * w), 0, 0, d, nsweep, 0.f, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(d);
    return (double)ms * 1e6 / nsweep / 64;     // ns per coordinate for all w waves of a SIMD
}
#define ROW(label, ...) { printf("%-46s", label); for (int w = 1; w <= 4; ++w) { double t = run(body<__VA_ARGS__>, w, 400); printf("  w=%d %6.1f ns (%5.1f /wave)", w, t, t / w); } printf("\n"); }
int main() {
    printf("ns per coordinate per SIMD (all resident waves advance one coordinate each)\n");
    ROW("LG=1 PD=2 full", 1, 2, true, true, true, true)
    ROW("LG=1 PD=2 no tol", 1, 2, true, true, false, true)
    ROW("LG=1 PD=2 no tol no cm", 1, 2, true, true, false, false)
    ROW("LG=1 no LDS no tol no cm", 1, 2, false, true, false, false)
    ROW("LG=2 PD=2 full", 2, 2, true, true, true, true)
    ROW("LG=2 PD=4 full", 2, 4, true, true, true, true)
    ROW("LG=2 PD=4 no blgp", 2, 4, true, false, true, true)
    ROW("LG=2 no LDS", 2, 4, false, true, true, true)
    ROW("LG=4 PD=2 full", 4, 2, true, true, true, true)
    ROW("LG=4 PD=4 full", 4, 4, true, true, true, true)
    ROW("LG=4 PD=6 full", 4, 6, true, true, true, true)
    ROW("LG=4 PD=6 no blgp", 4, 6, true, false, true, true)
    ROW("LG=4 PD=6 no tol", 4, 6, true, true, false, true)
    ROW("LG=4 PD=6 no tol no cm", 4, 6, true, true, false, false)
    ROW("LG=4 no LDS", 4, 6, false, true, true, true)
    ROW("LG=4 no LDS no blgp no tol no cm", 4, 6, false, false, false, false)
    return 0;
}
