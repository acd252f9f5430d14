// issue_probe2: does a bf16 MFMA overlap VALU work of the same / another wave?  (f32 MFMA does not: issue_probe.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ void k(float* out, long long* cyc, int rep, float s) {
    float v[8]; for (int i = 0; i < 8; ++i) v[i] = i + s + threadIdx.x;
    f32x16 acc[2]; for (int t = 0; t < 2; ++t) for (int e = 0; e < 16; ++e) acc[t][e] = s + e;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(s + e * 0.25f); b[e] = (__bf16)(s + 1.f + e); }
    float fa = s + 1.0001f, fb = s + 0.5f;
    long long t0 = clock64();
    for (int r = 0; r < rep; ++r) {
        if constexpr (MODE == 0) {          // 2 independent bf16 MFMAs 32x32x16
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[1], 0, 0, 0);
        } else if constexpr (MODE == 1) {   // 16 independent v_fma
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i & 7] = __builtin_fmaf(v[i & 7], fa, fb);
        } else if constexpr (MODE == 2) {   // both
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[0], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i & 7] = __builtin_fmaf(v[i & 7], fa, fb);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[1], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i & 7] = __builtin_fmaf(v[i & 7], fa, fb);
        } else if constexpr (MODE == 3) {   // 2 f32 MFMAs 32x32x2 + 16 fma (reference point)
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[0], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i & 7] = __builtin_fmaf(v[i & 7], fa, fb);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[1], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i & 7] = __builtin_fmaf(v[i & 7], fa, fb);
        } else if constexpr (MODE == 4) {   // 2 f32 MFMAs only
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[1], 0, 0, 0);
        } else if constexpr (MODE == 5) {   // dependent chain: acc element -> 4 VALU -> bf16 mfma (same tile)
            float d = acc[0][0] * fa; d = __builtin_fmaxf(d, fb); d = d * fa + fb; d = __builtin_fmaxf(d, fb);
            b[0] = (__bf16)d;
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[1], 0, 0, 0);
        } else if constexpr (MODE == 6) {   // the same chain with f32 MFMAs
            float d = acc[0][0] * fa; d = __builtin_fmaxf(d, fb); d = d * fa + fb; d = __builtin_fmaxf(d, fb);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, d, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, d, acc[1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    long long t1 = clock64();
    float sum = 0; for (int i = 0; i < 8; ++i) sum += v[i];
    for (int t = 0; t < 2; ++t) for (int e = 0; e < 16; ++e) sum += acc[t][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE> void run(const char* name, int wps) {
    float* d; long long* c; (void)hipMalloc(&d, 256 * 1024 * 4); (void)hipMalloc(&c, 8);
    const int rep = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, d, c, rep, 0.f);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, d, c, rep, 0.f);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-60s waves/SIMD=%d: %7.2f ns per loop body and wave\n", name, wps, ms * 1e6 / rep);
    (void)hipFree(d); (void)hipFree(c);
}
int main() {
    for (int wps = 1; wps <= 3; ++wps) {
        run<0>("2 x v_mfma_f32_32x32x16_bf16", wps);
        run<1>("16 x v_fma_f32 (independent)", wps);
        run<2>("2 x bf16 MFMA + 16 x v_fma_f32", wps);
        run<4>("2 x v_mfma_f32_32x32x2_f32", wps);
        run<3>("2 x f32 MFMA + 16 x v_fma_f32", wps);
        run<5>("CD-like chain (4 dependent VALU) + 2 bf16 MFMA", wps);
        run<6>("CD-like chain (4 dependent VALU) + 2 f32 MFMA", wps);
    }
    return 0;
}
