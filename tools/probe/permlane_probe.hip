// permlane_probe.hip -- prints what v_permlane16_swap_b32 / v_permlane32_swap_b32 do on this GPU (lane id patterns).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    const unsigned lane = threadIdx.x;
    unsigned a = 100 + lane, b = 200 + lane;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    out[lane] = a; out[64 + lane] = b;
    unsigned c = 100 + lane;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %0" : "+v"(c));
    out[128 + lane] = c;
    unsigned d = 100 + lane, e = 200 + lane;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(d), "+v"(e));
    out[192 + lane] = d; out[256 + lane] = e;
    unsigned f = 100 + lane;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %0" : "+v"(f));
    out[320 + lane] = f;
}
int main() {
    unsigned* d; hipMalloc(&d, 384 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[384]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[6] = {"p16 vdst(a=100+l)", "p16 src (b=200+l)", "p16 in-place", "p32 vdst", "p32 src", "p32 in-place"};
    for (int r = 0; r < 6; ++r) { printf("%-18s:", names[r]); for (int l = 0; l < 64; l += 8) printf(" [%u..]", h[r * 64 + l]); printf("\n"); }
    return 0;
}
