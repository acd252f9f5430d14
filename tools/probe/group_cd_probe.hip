// group_cd_probe: body of a 16-lanes-per-column, static-sweep VALU coordinate descent (4 columns per wave, k = 64: lane u of a
// 16-lane row holds rows u, u+16, u+32, u+48 of its column's residual).  Per coordinate: ds_read_b128 of -G(rows of the lane, i),
// mul, max, one masked save, four v_fmac_f32 with the step broadcast by DPP row_newbcast.  Prints ns per coordinate and wave for
// 1..8 waves per SIMD (throughput = that / waves).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
template <int B, int E, class Fn> __device__ __forceinline__ void sfor(Fn&& fn) {
    if constexpr (B < E) { fn(std::integral_constant<int, B>{}); sfor<B + 1, E>(fn); }
}
template <int I> __device__ __forceinline__ float bc(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + I, 0xf, 0xf, true));
}
__global__ void body(float* out, int sweeps, float s) {
    __shared__ float4 G[64 * 16];              // [coordinate][lane u]: the four rows of lane u
    for (int e = threadIdx.x; e < 64 * 16; e += blockDim.x) G[e] = make_float4(-0.001f * (e & 7), -0.002f, -0.0005f, -0.001f);
    __syncthreads();
    const int u = threadIdx.x & 15;
    float b[4], x[4], as[4];
    for (int r = 0; r < 4; ++r) { b[r] = 1.f + s + 0.01f * r + 0.001f * threadIdx.x; x[r] = 0.5f; as[r] = 0.f; }
    const float ginv = 0.9f + s;
    float tsum = 0.f;
    for (int it = 0; it < sweeps; ++it) {
        sfor<0, 4>([&](auto RC) {
            constexpr int r = decltype(RC)::value;
            sfor<0, 16>([&](auto UC) {
                constexpr int uu = decltype(UC)::value;
                constexpr int i = 16 * r + uu;
                const float4 g = G[i * 16 + u];
                const float diff = b[r] * ginv;
                const float ad = __builtin_fmaxf(diff, -x[r]);
                // lanes uu of every row keep their step for the deferred iterate update
                as[r] = (u == uu) ? ad : as[r];
                const float adb = bc<uu>(ad);
                b[0] = __builtin_fmaf(g.x, adb, b[0]);
                b[1] = __builtin_fmaf(g.y, adb, b[1]);
                b[2] = __builtin_fmaf(g.z, adb, b[2]);
                b[3] = __builtin_fmaf(g.w, adb, b[3]);
                __builtin_amdgcn_sched_barrier(0);
            });
            x[r] += as[r];
            tsum = __builtin_fmaf(__builtin_fabsf(as[r]), __builtin_amdgcn_rcpf(__builtin_fabsf(x[r]) + 1e-15f), tsum);
        });
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = b[0] + b[1] + b[2] + b[3] + x[0] + x[1] + x[2] + x[3] + tsum;
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 2048 * 4);
    const int sweeps = 200;
    for (int wps = 1; wps <= 8; ++wps) {
        if (wps == 7) continue;
        const int threads = 256 * (wps <= 4 ? wps : wps / 2), blocks = 256 * (wps <= 4 ? 1 : 2);
        hipLaunchKernelGGL(body, dim3(blocks), dim3(threads), 0, 0, d, sweeps, 0.f);
        (void)hipDeviceSynchronize();
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(body, dim3(blocks), dim3(threads), 0, 0, d, sweeps, 0.f);
        (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double per = ms * 1e6 / (sweeps * 64.0);
        printf("waves/SIMD %d: %.2f ns per coordinate and wave, %.2f ns per coordinate of SIMD time (4 columns) -> %.2f us per sweep\n",
               wps, per, per / wps, per * 64 / 1000.0);
    }
    return 0;
}
