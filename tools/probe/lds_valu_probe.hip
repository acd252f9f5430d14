// lds_valu_probe: do ds_read_b128 returns and VALU work overlap on a SIMD?  (the row-tiled rhs kernel shows VALU 44 % +
// LDS 45 % busy and a tile time equal to their SUM).  One workgroup per CU, wps waves per SIMD, a loop of "steps" shaped
// like the kernel's: [v_add_u32_dpp -> ds_read_b128] and [v_mov_b32_dpp -> 2 v_pk_fma_f32], two batches in flight.
//   mode 0: LDS reads only (results consumed by an empty asm)         mode 1: the VALU part only (no LDS reads)
//   mode 2: both, as the kernel does                                  mode 3: LDS reads + the two pk_fma only (no DPP ops)
//   mode 4: as 2 with the row read as 2 x ds_read_b64                 mode 5: as 2 with 4 x plain v_fma_f32 instead of 2 pk_fma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int I> __device__ __forceinline__ int bc(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + I, 0xf, 0xf, true); }
template <int B, int E, class Fn> __device__ __forceinline__ void sfor(Fn&& fn) {
    if constexpr (B < E) { fn(std::integral_constant<int, B>{}); sfor<B + 1, E>(fn); }
}
constexpr int UB = 4, NB = 8;      // 4 steps per batch, 8 batches per loop body = 32 steps (one tile of the NR = 8, S = 4 shape)
template <int MODE>
__global__ void k(float* out, int rep, int seed) {
    extern __shared__ char lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((float*)lds)[i] = 1e-6f * i;
    __syncthreads();
    const int lane = threadIdx.x & 63, u = lane & 15;
    const int lbase = u * 16;
    int co[2]; float cv[2];
    co[0] = ((lane * 37 + seed + (threadIdx.x >> 6) * 11) & 255) * 256;
    co[1] = ((lane * 53 + seed + (threadIdx.x >> 6) * 7) & 255) * 256;
    cv[0] = 1.0f + 1e-7f * lane; cv[1] = 1.0f - 1e-7f * lane;
    f4 acc[8];
    for (int q = 0; q < 8; ++q) acc[q] = f4{0.f, 0.f, 0.f, 0.f};
    f4 f[2][UB];
    for (int b = 0; b < 2; ++b) for (int i = 0; i < UB; ++i) f[b][i] = f4{1.f, 2.f, 3.f, 4.f};
    for (int r = 0; r < rep; ++r) {
        auto reads = [&](auto QB) {
            constexpr int b = decltype(QB)::value;
            sfor<0, UB>([&](auto I) {
                constexpr int i = decltype(I)::value;
                constexpr int step = b * UB + i;
                if constexpr (MODE == 1) {
                    const int a = lbase + bc<(step & 15)>(co[step >> 4]);
                    asm volatile("" ::"v"(a));
                } else if constexpr (MODE == 0 || MODE == 3) {
                    const int a = lbase + co[step >> 4] + ((step & 15) << 8);
                    f[b & 1][i] = *reinterpret_cast<const f4*>(lds + (a & 0xffff));
                } else if constexpr (MODE == 4) {
                    const int a = lbase + bc<(step & 15)>(co[step >> 4]);
                    const f2 lo = *reinterpret_cast<const f2*>(lds + a);
                    const f2 hi = *reinterpret_cast<const f2*>(lds + a + 8);
                    f[b & 1][i] = f4{lo[0], lo[1], hi[0], hi[1]};
                } else {
                    const int a = lbase + bc<(step & 15)>(co[step >> 4]);
                    f[b & 1][i] = *reinterpret_cast<const f4*>(lds + a);
                }
            });
        };
        if (r == 0) reads(std::integral_constant<int, 0>{});
        sfor<0, NB>([&](auto QB) {
            constexpr int b = decltype(QB)::value;
            reads(std::integral_constant<int, (b + 1) % NB>{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MODE != 1) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MODE == 4 ? 2 * UB : UB) : "memory");
            sfor<0, UB>([&](auto I) {
                constexpr int i = decltype(I)::value;
                constexpr int step = b * UB + i;
                if constexpr (MODE == 0) {
                    const f4 t = f[b & 1][i];
                    asm volatile("" ::"v"(t));
                } else {
                    float val;
                    if constexpr (MODE == 3) val = cv[step >> 4];
                    else val = __int_as_float(bc<(step & 15)>(__float_as_int(cv[step >> 4])));
                    if constexpr (MODE == 5) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float a_ = acc[step / 4][e];
                            const float f_ = f[b & 1][i][e];
                            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a_) : "v"(val), "v"(f_));
                            acc[step / 4][e] = a_;
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[step / 4][e] = __builtin_fmaf(val, f[b & 1][i][e], acc[step / 4][e]);
                    }
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    f4 s = acc[0];
    for (int q = 1; q < 8; ++q) s += acc[q];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
template <int MODE> void run(const char* name, int wps) {
    float* d; (void)hipMalloc(&d, 256 * 1024 * 4);
    const int rep = 4000;
    auto kern = k<MODE>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256 * wps), 131072, 0, d, rep, 3);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256 * wps), 131072, 0, d, rep, 3);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns_body = ms * 1e6 / rep;                 // 32 steps per wave
    printf("%-52s waves/SIMD=%d: %8.1f ns per 32-step body -> %6.2f ns per step and SIMD (%.2f cycles @2.4 GHz per step and wave-slot)\n", name, wps,
           ns_body, ns_body / 32.0 / wps, ns_body / 32.0 / wps * 2.4);
    (void)hipFree(d);
}
int main() {
    for (int wps = 1; wps <= 4; ++wps) {
        run<0>("LDS only: 32 x ds_read_b128", wps);
        run<1>("VALU only: 32 x (add_dpp, mov_dpp, 2 pk_fma)", wps);
        run<2>("both (the kernel's step)", wps);
        run<3>("ds_read_b128 + 2 pk_fma, no DPP", wps);
        run<4>("both, row read as 2 x ds_read_b64", wps);
        run<5>("both, 4 x v_fmac_f32 instead of 2 x v_pk_fma", wps);
    }
    return 0;
}
