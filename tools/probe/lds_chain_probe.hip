// tools/probe/lds_chain_probe.hip -- what does one step of a scalar dependent chain through LDS cost on gfx950?
// (the rANS decoder of ops_spz.hip is such a chain).  Prints shader-clock cycles per step, the shader clock the launch ran
// at (clock64 vs the 100 MHz wall_clock64) for 22 and for 4096 single-wave blocks.
// Build: hipcc --offload-arch=gfx950 -O3 lds_chain_probe.hip -o lds_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(64) void chain(const unsigned* init, int steps, unsigned long long* out, int mode) {
    extern __shared__ unsigned long long tab[];
    for (int i = threadIdx.x; i < 16384; i += 64) tab[i] = ((unsigned long long)(i * 2654435761u % 16384u)) | (1ull << 40);
    __syncthreads();
    unsigned x = __builtin_amdgcn_readfirstlane((int)init[blockIdx.x]);
    const long long c0 = clock64(), w0 = wall_clock64();
    if (mode == 0) {            // LDS read -> readfirstlane -> mask
        for (int s = 0; s < steps; ++s) {
            const unsigned long long e = tab[x & 16383u];
            x = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)e) + (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(e >> 32)) - 256u;
        }
    } else {                    // pure scalar ALU chain of the same length (mul, shifts, add)
        for (int s = 0; s < steps; ++s) x = (x >> 14) * 40503u + (x & 16383u) + 7u;
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[3 * blockIdx.x] = c1 - c0; out[3 * blockIdx.x + 1] = w1 - w0; out[3 * blockIdx.x + 2] = x; }
}
int main() {
    const int steps = 200000;
    for (int blocks : {22, 4096}) for (int mode : {0, 1}) {
        std::vector<unsigned> h(blocks); for (int i = 0; i < blocks; ++i) h[i] = 12345u + 77u * i;
        unsigned* d; unsigned long long* o;
        hipMalloc(&d, blocks * 4); hipMalloc(&o, blocks * 24); hipMemcpy(d, h.data(), blocks * 4, hipMemcpyHostToDevice);
        hipFuncSetAttribute((const void*)chain, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        chain<<<blocks, 64, 131072>>>(d, steps, o, mode); hipDeviceSynchronize();
        chain<<<blocks, 64, 131072>>>(d, steps, o, mode); hipDeviceSynchronize();
        std::vector<unsigned long long> r(3 * blocks); hipMemcpy(r.data(), o, blocks * 24, hipMemcpyDeviceToHost);
        printf("blocks %5d mode %s: %.1f shader cycles / step, %.1f ns / step, shader clock %.0f MHz\n", blocks, mode ? "salu" : "lds ",
               (double)r[0] / steps, (double)r[1] * 10.0 / steps, (double)r[0] / ((double)r[1] * 10.0) * 1e3);
        hipFree(d); hipFree(o);
    }
}
