// Probe: does global_load_lds_dword honour LDS addresses >= 128 KiB in M0, and how does the ushort form lay lanes out?
// hipcc --offload-arch=gfx950 -O2 tools/probe/glds_addr_probe.hip -o /tmp/glds_probe && /tmp/glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned* src, unsigned* out, unsigned ldsaddr, int mode) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < 40960; i += 64) lds[i] = 0xdeadbeef;
    __syncthreads();
    unsigned lane_off = mode == 0 ? threadIdx.x * 4 : threadIdx.x * 2;
    if (mode == 0)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1\n\ts_waitcnt vmcnt(0)" ::"v"(lane_off), "s"(src), "s"(ldsaddr) : "memory");
    else
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_ushort %0, %1\n\ts_waitcnt vmcnt(0)" ::"v"(lane_off), "s"(src), "s"(ldsaddr) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 40960; i += 64) out[i] = lds[i];
}
int main() {
    unsigned *src, *out;
    hipMalloc(&src, 4096); hipMalloc(&out, 163840);
    std::vector<unsigned> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 0x1000 + i;
    hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    std::vector<unsigned> o(40960);
    for (unsigned addr : {0u, 65536u, 131072u - 256u, 131072u, 131072u + 4096u, 163840u - 256u}) {
        for (int mode = 0; mode < 2; ++mode) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 163840, 0, src, out, addr, mode);
            hipMemcpy(o.data(), out, 163840, hipMemcpyDeviceToHost);
            int first = -1, last = -1, cnt = 0;
            for (int i = 0; i < 40960; ++i) if (o[i] != 0xdeadbeef) { if (first < 0) first = i; last = i; ++cnt; }
            printf("m0=%u mode=%d: %d dwords changed, first byte %d last byte %d, first values %08x %08x %08x\n", addr, mode, cnt, first * 4, last * 4,
                   first >= 0 ? o[first] : 0, first >= 0 ? o[first + 1] : 0, first >= 0 ? o[first + 2] : 0);
        }
    }
    return 0;
}
