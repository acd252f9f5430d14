// scan.hip.h -- exclusive prefix sum of 32-bit counts on the device (set-up code: CSC pointers of the transpose, of the
// overflow lists of the row-tiled plans).  Three small kernels: per-tile sums, one workgroup scanning the tile sums, per-tile
// rescan with the tile's offset.  Own code: nothing from rocPRIM / hipCUB is on the call path.
#pragma once
#include "common.hip.h"

namespace rk {

constexpr int SCAN_TILE = 2048;      // items per workgroup: 256 threads x 8

__device__ __forceinline__ int scan_wave_incl(int v) {          // inclusive scan across the 64 lanes of a wavefront
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(v, d, 64);
        if ((int)(threadIdx.x & 63) >= d) v += o;
    }
    return v;
}
// exclusive scan of one value per thread over a 256-thread workgroup; *total (optional) receives the workgroup's sum
__device__ __forceinline__ int scan_block_excl(int v, int* sh /* 4 ints */, int* total) {
    const int incl = scan_wave_incl(v);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) sh[w] = incl;
    __syncthreads();
    int base = 0;
    for (int i = 0; i < w; ++i) base += sh[i];
    if (total) *total = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return base + incl - v;
}

static __global__ __launch_bounds__(256) void scan_tile_sums_kernel(const int* __restrict__ in, int64_t n, int* __restrict__ sums) {
    __shared__ int sh[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * 8;
    int s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (base + i < n) ? in[base + i] : 0;
    int tot;
    (void)scan_block_excl(s, sh, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}
static __global__ __launch_bounds__(256) void scan_sums_kernel(int* __restrict__ sums, int64_t ntiles) {       // one workgroup, in place
    __shared__ int sh[4];
    int carry = 0;
    for (int64_t b = 0; b < ntiles; b += 256) {
        const int64_t i = b + threadIdx.x;
        const int v = i < ntiles ? sums[i] : 0;
        int tot;
        const int ex = scan_block_excl(v, sh, &tot);
        if (i < ntiles) sums[i] = carry + ex;
        carry += tot;
    }
}
static __global__ __launch_bounds__(256) void scan_apply_kernel(const int* __restrict__ in, int64_t n, const int* __restrict__ sums,
                                                         int* __restrict__ out) {
    __shared__ int sh[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * 8;
    int v[8], s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = (base + i < n) ? in[base + i] : 0; s += v[i]; }
    int run = sums[blockIdx.x] + scan_block_excl(s, sh, nullptr);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (base + i < n) out[base + i] = run;
        run += v[i];
    }
}

// out[i] = in[0] + ... + in[i-1] for i < n (in and out may alias).  Stream-ordered; `sums` scratch comes from the context's
// arena or hipMalloc (then the stream is synchronised before it is freed).
inline void exclusive_scan_i32(rcppml_hip_ctx* c, const int* in, int* out, int64_t n) {
    if (n <= 0) return;
    const int64_t nt = (n + SCAN_TILE - 1) / SCAN_TILE;
    DevTmp sums(c, (size_t)nt * sizeof(int));
    hipLaunchKernelGGL(scan_tile_sums_kernel, dim3((unsigned)nt), dim3(256), 0, c->stream, in, n, (int*)sums.p);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(256), 0, c->stream, (int*)sums.p, nt);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nt), dim3(256), 0, c->stream, in, n, (const int*)sums.p, out);
    HIPCHK(hipGetLastError());
    if (sums.owned) HIPCHK(hipStreamSynchronize(c->stream));
}

}  // namespace rk
