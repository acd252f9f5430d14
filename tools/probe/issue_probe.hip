// issue_probe: cycles per instruction for ONE wave per SIMD: independent / dependent VALU, with 4x4 MFMAs in between
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void k(float* out, long long* cyc, int rep, float s) {
    float v[8]; for (int i = 0; i < 8; ++i) v[i] = i + s + threadIdx.x;
    f32x4 acc[4]; for (int t = 0; t < 4; ++t) acc[t] = f32x4{s, 1.f, 2.f, 3.f};
    float a = s + 1.0001f, b = s + 0.5f;
    long long t0 = clock64();
    for (int r = 0; r < rep; ++r) {
        if constexpr (MODE == 0) {       // 8 independent fma
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], a, b);
        } else if constexpr (MODE == 1) {   // 8 dependent fma
#pragma unroll
            for (int i = 0; i < 8; ++i) v[0] = __builtin_fmaf(v[0], a, b);
        } else if constexpr (MODE == 2) {   // 4 independent mfma 4x4
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[t], 4, 0, 0);
        } else if constexpr (MODE == 3) {   // 4 mfma + 4 independent fma
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[t], 4, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = __builtin_fmaf(v[i], a, b);
        } else if constexpr (MODE == 4) {   // chain: acc elem -> mul -> max -> 4 mfma (CD-like, dependent)
            float d = acc[0][0] * a;
            d = __builtin_fmaxf(d, b);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, d, acc[t], 4, 0, 0);
        } else if constexpr (MODE == 5) {   // chain with VALU lookahead: v0 chain only, mfma independent of next step
            float d = v[0] * a;
            d = __builtin_fmaxf(d, b);
            v[0] = __builtin_fmaf(d, a, v[0]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, d, acc[t], 4, 0, 0);
        } else if constexpr (MODE == 6) {   // 1 dependent mfma chain on one tile
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[0], 4, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    long long t1 = clock64();
    float sum = 0; for (int i = 0; i < 8; ++i) sum += v[i];
    for (int t = 0; t < 4; ++t) sum += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE> void run(const char* name, int ninstr, int wps) {
    float* d; long long* c; hipMalloc(&d, 256 * 1024 * 4); hipMalloc(&c, 8);
    const int rep = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, d, c, rep, 0.f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, d, c, rep, 0.f);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    printf("%-48s wps=%d: %7.1f clock64 ticks/iter, %7.2f ns/iter  (%d instr/iter -> %.2f ns/instr)\n", name, wps, (double)cy / rep, ms * 1e6 / rep, ninstr, ms * 1e6 / rep / ninstr);
    hipFree(d); hipFree(c);
}
int main() {
    for (int wps = 1; wps <= 2; ++wps) {
        run<0>("8 independent v_fma", 8, wps);
        run<1>("8 dependent v_fma", 8, wps);
        run<2>("4 independent mfma4x4", 4, wps);
        run<6>("4 dependent mfma4x4 (same tile)", 4, wps);
        run<3>("4 mfma4x4 + 4 independent v_fma", 8, wps);
        run<4>("CD chain: acc->mul->max->4 mfma", 6, wps);
        run<5>("CD chain with VALU lookahead + 4 mfma", 7, wps);
    }
    return 0;
}
