// grid_barrier_probe: what a grid-wide barrier INSIDE one kernel costs on gfx950, and what it needs to be correct.
// (1) workgroup -> XCD mapping (HW_REG_XCC_ID); (2) latency of a barrier among B persistent workgroups, (a) with device-scope
// release / acquire fences (L2 write-back + invalidate: what cross-XCD visibility needs), (b) with relaxed L2 atomics and L1-bypassing
// loads only (enough when every workgroup sits on ONE XCD: one L2); (3) a visibility check for (b): every round each block stores
// its round number into its own slot, the barrier, then reads every other block's slot with an L1-bypassing load.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe/grid_barrier_probe.hip -o /tmp/gbp ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
__device__ __forceinline__ unsigned ld_l2(const unsigned* p) {          // relaxed, agent scope: bypasses the CU's L1
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// MODE 0: device-scope fences (safe across XCDs); MODE 1: relaxed L2 atomics, no cache maintenance (one XCD)
// (a spin that lasts longer than ~0.3 s of the 100 MHz counter gives up: a probe must never hang the box)
__device__ unsigned g_abort;
template <int MODE>
__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned nblocks, unsigned& gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        ++gen;
        const long long t0 = clock64();
        if (MODE == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);          // agent-scope release: buffer_wbl2 sc1 + waits
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen * nblocks) {
                __builtin_amdgcn_s_sleep(1);
                if (clock64() - t0 > 30000000ll || ld_l2(&g_abort)) { g_abort = 1; break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                    // buffer_inv sc1
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); // this block's stores have reached L2
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (ld_l2(counter) < gen * nblocks) {
                __builtin_amdgcn_s_sleep(1);
                if (clock64() - t0 > 30000000ll || ld_l2(&g_abort)) { g_abort = 1; break; }
            }
        }
    }
    __syncthreads();
    // MODE 2: the relaxed barrier, then every wave drops its CU's L1 lines -- ordinary (vectorisable) loads may follow
    if (MODE == 2) asm volatile("buffer_inv sc0" ::: "memory");
}

// only the workgroups that land on XCD `want_xcc` take part (want_xcc < 0: all); participants count themselves in `live` first
template <int MODE>
__global__ __launch_bounds__(256) void probe(unsigned* counter, unsigned* live, unsigned* slots, int want_xcc, unsigned expect, int rounds,
                                             unsigned* bad, long long* cycles, unsigned* xcc_of_block) {
    const unsigned x = xcc_id();
    if (threadIdx.x == 0) xcc_of_block[blockIdx.x] = x;
    if (want_xcc >= 0 && (int)x != want_xcc) return;
    __shared__ unsigned my;
    if (threadIdx.x == 0) my = __hip_atomic_fetch_add(live, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned me = my;
    if (me >= expect) return;                                           // more blocks on the XCD than asked for: leave
    unsigned gen = 0, wrong = 0;
    grid_barrier<MODE>(counter, expect, gen);                           // everybody is resident
    const long long t0 = clock64();
    for (int r = 1; r <= rounds; ++r) {
        // every thread of the block stores (a 1 KiB record per block: 256 words), so that "visible" means the whole record
        slots[(size_t)me * 256 + threadIdx.x] = (unsigned)r * 1000u + threadIdx.x;
        grid_barrier<MODE>(counter, expect, gen);
        for (unsigned b = 0; b < expect; ++b) {
            const unsigned v = MODE != 1 ? slots[(size_t)b * 256 + threadIdx.x] : ld_l2(&slots[(size_t)b * 256 + threadIdx.x]);
            wrong += v != (unsigned)r * 1000u + threadIdx.x;
        }
        grid_barrier<MODE>(counter, expect, gen);                       // nobody overwrites before everybody has read
    }
    const long long t1 = clock64();
    if (wrong) atomicAdd(bad, wrong);
    if (me == 0 && threadIdx.x == 0) *cycles = t1 - t0;
}

// barrier-only latency (no payload)
template <int MODE>
__global__ __launch_bounds__(256) void lat(unsigned* counter, unsigned* live, int want_xcc, unsigned expect, int rounds, long long* cycles) {
    const unsigned x = xcc_id();
    if (want_xcc >= 0 && (int)x != want_xcc) return;
    __shared__ unsigned my;
    if (threadIdx.x == 0) my = __hip_atomic_fetch_add(live, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (my >= expect) return;
    unsigned gen = 0;
    grid_barrier<MODE>(counter, expect, gen);
    const long long t0 = clock64();
    for (int r = 0; r < rounds; ++r) grid_barrier<MODE>(counter, expect, gen);
    const long long t1 = clock64();
    if (my == 0 && threadIdx.x == 0) *cycles = t1 - t0;
}

int main() {
    unsigned *counter, *live, *slots, *bad, *xcc;
    long long* cyc;
    hipMalloc(&counter, 4); hipMalloc(&live, 4); hipMalloc(&bad, 4); hipMalloc(&cyc, 8);
    hipMalloc(&slots, 256 * 256 * 4); hipMalloc(&xcc, 4096 * 4);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    printf("device %s, %d CUs, clock64 runs at 100 MHz on this part (ticks of 10 ns)\n", prop.name, prop.multiProcessorCount);
    // (1) mapping
    {
        hipMemset(counter, 0, 4); hipMemset(live, 0, 4); hipMemset(bad, 0, 4);
        hipLaunchKernelGGL(probe<1>, dim3(64), dim3(256), 0, 0, counter, live, slots, 99, 1u, 0, bad, cyc, xcc);   // nobody participates
        hipDeviceSynchronize();
        std::vector<unsigned> h(64);
        hipMemcpy(h.data(), xcc, 64 * 4, hipMemcpyDeviceToHost);
        printf("XCC_ID of blocks 0..63:");
        for (int i = 0; i < 64; ++i) printf(" %u", h[i]);
        printf("\n");
    }
    struct Case { const char* name; int mode, want, grid; unsigned expect; };
    const Case cases[] = {
        {"fences, all XCDs, 256 blocks", 0, -1, 256, 256}, {"fences, all XCDs, 64 blocks", 0, -1, 64, 64},
        {"fences, XCD 0 only, 32 blocks", 0, 0, 256, 32},
        {"relaxed L2, XCD 0 only, 32 blocks", 1, 0, 256, 32}, {"relaxed L2, XCD 0 only, 16 blocks", 1, 0, 256, 16},
        {"relaxed L2 + buffer_inv sc0 + plain loads, XCD 0 only, 32 blocks", 2, 0, 256, 32},
        {"relaxed L2 + buffer_inv sc0 + plain loads, XCD 0 only, 32 blocks (again)", 2, 0, 256, 32},
        {"relaxed L2, all XCDs, 64 blocks (expected to FAIL the visibility check or not)", 1, -1, 64, 64},
    };
    const int rounds = 2000;
    for (const Case& c : cases) {
        for (int what = 0; what < 2; ++what) {
            { unsigned z = 0; hipMemcpyToSymbol(HIP_SYMBOL(g_abort), &z, 4); }
            hipMemset(counter, 0, 4); hipMemset(live, 0, 4); hipMemset(bad, 0, 4); hipMemset(cyc, 0, 8);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (what == 0) {
                if (c.mode == 0) hipLaunchKernelGGL(lat<0>, dim3(c.grid), dim3(256), 0, 0, counter, live, c.want, c.expect, rounds, cyc);
                else if (c.mode == 2) hipLaunchKernelGGL(lat<2>, dim3(c.grid), dim3(256), 0, 0, counter, live, c.want, c.expect, rounds, cyc);
                else hipLaunchKernelGGL(lat<1>, dim3(c.grid), dim3(256), 0, 0, counter, live, c.want, c.expect, rounds, cyc);
            } else {
                if (c.mode == 0) hipLaunchKernelGGL(probe<0>, dim3(c.grid), dim3(256), 0, 0, counter, live, slots, c.want, c.expect, rounds, bad, cyc, xcc);
                else if (c.mode == 2) hipLaunchKernelGGL(probe<2>, dim3(c.grid), dim3(256), 0, 0, counter, live, slots, c.want, c.expect, rounds, bad, cyc, xcc);
                else hipLaunchKernelGGL(probe<1>, dim3(c.grid), dim3(256), 0, 0, counter, live, slots, c.want, c.expect, rounds, bad, cyc, xcc);
            }
            hipEventRecord(e1);
            hipError_t err = hipDeviceSynchronize();
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            unsigned hb = 0, hl = 0; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(&hl, live, 4, hipMemcpyDeviceToHost);
            { unsigned ab = 0; hipMemcpyFromSymbol(&ab, HIP_SYMBOL(g_abort), 4); if (ab) printf("  !! spin timed out (participants never all arrived)\n"); }
            if (what == 0)
                printf("%-80s barrier only : %8.3f us per barrier (kernel %.3f ms, %u participants) %s\n", c.name, ms * 1e3 / rounds, ms, hl, hipGetErrorString(err));
            else
                printf("%-80s store+2 barriers+read-all: %8.3f us per round, wrong reads %u %s\n", c.name, ms * 1e3 / rounds, hb, hipGetErrorString(err));
        }
    }
    return 0;
}
