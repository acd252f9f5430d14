// mfma4x4_probe.hip -- what v_mfma_f32_4x4x1_16B_f32 does on this GPU: operand / result lane maps under the CBSZ / ABID (A-block
// broadcast) and BLGP (B lane-group) controls, issue rate of independent and dependent instructions, overlap with VALU work,
// and the latency of the coordinate-descent chain  mul -> max -> MFMA -> (next residual).  Feeds kernels_cd_lane_mfma.hip.h.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma4x4_probe mfma4x4_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>
#include <utility>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

template <int CBSZ, int ABID, int BLGP>
__global__ void sem_kernel(float* out) {
    const int lane = threadIdx.x;
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    // run 1: which A lane feeds (lane, reg);  run 2: which B lane
    f32x4 da = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(lane + 1), 1.f, z, CBSZ, ABID, BLGP);
    f32x4 db = __builtin_amdgcn_mfma_f32_4x4x1f32(1.f, (float)(lane + 1), z, CBSZ, ABID, BLGP);
    for (int v = 0; v < 4; ++v) { out[(v * 64 + lane) * 2] = da[v] - 1.f; out[(v * 64 + lane) * 2 + 1] = db[v] - 1.f; }
}

template <int CBSZ, int ABID, int BLGP>
static void sem(const char* name) {
    float* d; hipMalloc(&d, 512 * 4);
    hipLaunchKernelGGL((sem_kernel<CBSZ, ABID, BLGP>), dim3(1), dim3(64), 0, 0, d);
    float h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%s (cbsz=%d abid=%d blgp=%d): D[lane][reg] = A[laneA] * B[laneB]\n", name, CBSZ, ABID, BLGP);
    for (int lane = 0; lane < 64; ++lane) {
        if (!(lane < 8 || (lane % 16) < 2 || lane >= 60)) continue;
        printf("  lane %2d:", lane);
        for (int v = 0; v < 4; ++v) printf("  r%d A%2d B%2d", v, (int)h[(v * 64 + lane) * 2], (int)h[(v * 64 + lane) * 2 + 1]);
        printf("\n");
    }
    hipFree(d);
}


template <int ABID> __device__ __forceinline__ f32x4 mf(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, ABID, 0); }
// runtime tile index (constant after unrolling) -> immediate ABID
__device__ __forceinline__ f32x4 mfi(int t, float a, float b, f32x4 c) {
    switch (t & 15) {
        case 0: return mf<0>(a, b, c); case 1: return mf<1>(a, b, c); case 2: return mf<2>(a, b, c); case 3: return mf<3>(a, b, c);
        case 4: return mf<4>(a, b, c); case 5: return mf<5>(a, b, c); case 6: return mf<6>(a, b, c); case 7: return mf<7>(a, b, c);
        case 8: return mf<8>(a, b, c); case 9: return mf<9>(a, b, c); case 10: return mf<10>(a, b, c); case 11: return mf<11>(a, b, c);
        case 12: return mf<12>(a, b, c); case 13: return mf<13>(a, b, c); case 14: return mf<14>(a, b, c); default: return mf<15>(a, b, c);
    }
}

// ---- timing kernels: one wave per SIMD unless stated -----------------------------------------------------------------------
// NT independent accumulator tiles, NV independent v_fma between MFMA groups, REP groups
template <int NT, int NV, bool DEP>
__global__ void rate_kernel(float* out, long long* cyc, int rep, float s) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[16];
    for (int t = 0; t < 16; ++t) acc[t] = f32x4{(float)t, 1.f, 2.f, 3.f};
    float a = lane * 0.001f + s, b = 1.0001f + s;
    float va[8];
    for (int i = 0; i < 8; ++i) va[i] = i + s;
    __builtin_amdgcn_s_barrier();
    const long long t0 = clock64();
    for (int r = 0; r < rep; ++r) {
        static_for<0, NT>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            constexpr int tt = DEP ? 0 : t;
            acc[tt] = mf<t & 15>(a, b, acc[tt]);
        });
        static_for<0, NV>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            va[i & 7] = __builtin_fmaf(va[i & 7], b, a);
        });
    }
    const long long t1 = clock64();
    float sum = 0.f;
    for (int t = 0; t < 16; ++t) sum += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    for (int i = 0; i < 8; ++i) sum += va[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// the CD chain: residual element -> v_mul -> v_max -> MFMA on NT tiles (the tile read next goes first) -> next coordinate
template <int NT>
__global__ void chain_kernel(float* out, long long* cyc, int rep, float s) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[16];
    for (int t = 0; t < 16; ++t) acc[t] = f32x4{(float)t + s, 1.f, 2.f, 3.f};
    float g = lane * 0.001f + s, x = 0.5f + s, ginv = 0.9f + s;
    const long long t0 = clock64();
    for (int r = 0; r < rep; ++r) {
        static_for<0, 4 * NT>([&](auto ic) {          // coordinate i lives in tile i/4, element i%4
            constexpr int i = decltype(ic)::value;
            const float bi = acc[i >> 2][i & 3];
            const float diff = bi * ginv;
            const float a = __builtin_fmaxf(diff, -x);
            x = x + a;
            constexpr int tn = ((i + 1) >> 2) % NT;
            static_for<0, NT>([&](auto uc) {
                constexpr int t = (tn + decltype(uc)::value) % NT;
                acc[t] = mf<t & 15>(g, a, acc[t]);
            });
        });
    }
    const long long t1 = clock64();
    float sum = x;
    for (int t = 0; t < 16; ++t) sum += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

static double g_wall_ns_per_group = 0;
template <class K>
static double run(K kern, int waves_per_simd, int rep, int nblocks = 256) {
    float* d; long long* c;
    hipMalloc(&d, (size_t)nblocks * 1024 * 4); hipMalloc(&c, 8);
    const int threads = 256 * waves_per_simd;     // 4 SIMDs x waves
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(threads), 0, 0, d, c, rep, 0.f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(threads), 0, 0, d, c, rep, 0.f);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    hipFree(d); hipFree(c);
    g_wall_ns_per_group = (double)ms * 1e6 / rep;     // wall time of the whole grid per group (all waves run concurrently)
    return (double)h / rep;
}
static double wall() { return g_wall_ns_per_group; }

int main() {
    sem<0, 0, 0>("plain");
    sem<4, 0, 0>("A block 0 to all");
    sem<4, 5, 0>("A block 5 to all");
    sem<2, 1, 0>("A block 1 of each group of 4");
    sem<3, 2, 0>("A block 2 of each group of 8");
    sem<0, 0, 1>("blgp 1");
    sem<0, 0, 2>("blgp 2");
    sem<0, 0, 3>("blgp 3");
    sem<0, 0, 4>("blgp 4");
    sem<0, 0, 5>("blgp 5");
    sem<0, 0, 6>("blgp 6");
    sem<0, 0, 7>("blgp 7");
    sem<2, 3, 6>("cbsz 2 abid 3 blgp 6");
    const int rep = 2000;
    printf("\ncycles per group (clock64 ticks; s_memtime runs at 100 MHz on some parts -- compare ratios)\n");
    for (int w = 1; w <= 4; ++w) {
        double c;
#define SHOW(label, kern, r) c = run(kern, w, r); printf("  w=%d %-34s ticks/group %8.1f  wall ns/group %8.2f\n", w, label, c, wall());
        SHOW("16 indep MFMA", (rate_kernel<16, 0, false>), rep)
        SHOW("8 indep MFMA", (rate_kernel<8, 0, false>), rep)
        SHOW("4 indep MFMA", (rate_kernel<4, 0, false>), rep)
        SHOW("16 dependent MFMA", (rate_kernel<16, 0, true>), rep)
        SHOW("16 MFMA + 8 v_fma", (rate_kernel<16, 8, false>), rep)
        SHOW("16 MFMA + 32 v_fma", (rate_kernel<16, 32, false>), rep)
        SHOW("0 MFMA + 32 v_fma", (rate_kernel<0, 32, false>), rep)
        SHOW("chain NT=16 (64 coords)", (chain_kernel<16>), 200)
        SHOW("chain NT=8 (32 coords)", (chain_kernel<8>), 200)
        SHOW("chain NT=4 (16 coords)", (chain_kernel<4>), 200)
    }
    return 0;
}
