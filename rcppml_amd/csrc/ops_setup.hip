// ops_setup.hip -- one-time setup of a fit on the device: CSC transpose (A^T for the W half-update, reference
// nmf/fit_cpu.hpp:251-253 `At = A.transpose()`) and precision casts of the host's double buffers.
//
// The transpose is a STABLE sort of the nonzero positions by row index (own kernels, no library sort).  Up to 32 768 rows a counting
// sort with one LDS counter per row: the columns are cut into B chunks; chunk b counts its nonzeros per row in LDS (transpose_count_kernel), a prefix over the chunks turns the
// B x rows table into "where chunk b's first nonzero of row r goes" (transpose_prefix_kernel + the row-pointer scan), and
// chunk b then walks its columns IN ORDER, handing out positions from LDS counters (transpose_scatter_kernel).  Positions of
// one row stay in increasing order, i.e. column indices of A^T come out sorted exactly as Eigen's transpose produces them.
#include "common.hip.h"
#include "scan.hip.h"

namespace {

constexpr int TR_RCH = 32768;        // rows the LDS-counter form handles (128 KiB of counters); taller inputs: radix passes below
// Chunks are cut by NONZEROS, on column boundaries (round 5; until then by column count, which left a matrix whose nonzeros sit in
// a few columns to a few workgroups): chunk b = columns [bound(b), bound(b+1)), bound(b) = the first column j with
// p[j] >= floor(b nnz / B) (bound(0) = 0, bound(B) = cols) -- non-decreasing in b, so the chunks tile the columns; both kernels
// evaluate the same function.  A single column heavier than nnz / B is still one chunk's.
__device__ __forceinline__ int tr_bound(const int* __restrict__ p, int cols, int b, int B) {
    if (b <= 0) return 0;
    if (b >= B) return cols;
    const int target = (int)((int64_t)p[cols] * b / B);
    int lo = 0, hi = cols;                     // first j in [0, cols] with p[j] >= target (p[cols] = nnz >= target)
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (p[mid] < target) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// chunk b: nonzeros per row -> cnt[b][row]
__global__ __launch_bounds__(256) void transpose_count_kernel(const int* __restrict__ p, const int* __restrict__ ri, int cols, int rows,
                                                              int* __restrict__ cnt) {
    extern __shared__ int tr_lc[];
    __shared__ int bnd[2];
    const int b = blockIdx.x;
    if (threadIdx.x < 2) bnd[threadIdx.x] = tr_bound(p, cols, b + (int)threadIdx.x, (int)gridDim.x);
    __syncthreads();
    const int c0 = bnd[0], c1 = bnd[1];
    const int e0 = p[c0], e1 = p[c1];
    // rows <= TR_RCH here (taller inputs take the radix passes below): one LDS counter per row
    for (int i = threadIdx.x; i < rows; i += 256) tr_lc[i] = 0;
    __syncthreads();
    for (int e = e0 + (int)threadIdx.x; e < e1; e += 256) atomicAdd(&tr_lc[ri[e]], 1);
    __syncthreads();
    for (int i = threadIdx.x; i < rows; i += 256) cnt[(size_t)b * rows + i] = tr_lc[i];
}
// cnt[b][r] <- sum of cnt[b'][r] over b' < b; rowcnt[r] <- the row's total (rowcnt[rows] = 0 feeds the pointer scan)
__global__ void transpose_prefix_kernel(int* __restrict__ cnt, int rows, int B, int* __restrict__ rowcnt) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > rows) return;
    int run = 0;
    if (r < rows)
        for (int b = 0; b < B; ++b) {
            const int t = cnt[(size_t)b * rows + r];
            cnt[(size_t)b * rows + r] = run;
            run += t;
        }
    rowcnt[r] = run;
}
// chunk b hands out the positions of its nonzeros: column by column (a barrier between columns keeps the positions of one row
// in column order), inside a column the nonzeros of different rows in parallel
__global__ __launch_bounds__(256) void transpose_scatter_kernel(const int* __restrict__ p, const int* __restrict__ ri, int cols, int rows,
                                                                const int* __restrict__ cnt, const int* __restrict__ tp,
                                                                int* __restrict__ pos_out) {
    extern __shared__ int tr_lc[];
    __shared__ int bnd[2];
    const int b = blockIdx.x;
    if (threadIdx.x < 2) bnd[threadIdx.x] = tr_bound(p, cols, b + (int)threadIdx.x, (int)gridDim.x);
    __syncthreads();
    const int c0 = bnd[0], c1 = bnd[1];
    for (int i = threadIdx.x; i < rows; i += 256) tr_lc[i] = tp[i] + cnt[(size_t)b * rows + i];      // rows <= TR_RCH
    __syncthreads();
    for (int j = c0; j < c1; ++j) {
        const int e1 = p[j + 1];
        for (int e = p[j] + (int)threadIdx.x; e < e1; e += 256) pos_out[atomicAdd(&tr_lc[ri[e]], 1)] = e;
        __syncthreads();
    }
}
// ---------------------------------------------------------------------------
// Tall inputs (rows > TR_RCH): the passes above would re-read every chunk's nonzeros once per 32 768 rows (a 1M-row matrix: 31
// times) and the B x rows table caps the number of chunks.  Instead: a STABLE least-significant-digit radix sort of the nonzero
// positions by row index, 8 bits per pass (2 passes up to 65 536 rows, 3 up to 16.7 M), each pass = per-block digit counts
// (radix_hist_kernel) + ranked scatter (radix_scatter_kernel: blocks take contiguous chunks of the current order, the four
// wavefronts of a block contiguous quarters, inside a 64-element round the rank is the number of lower lanes with the same digit --
// eight ballots), so equal digits keep their order and positions of one row end in increasing order, as the reference's
// transpose leaves them.  O(passes * nnz) whatever the shape; no atomics on global memory except the row histogram (integer counts:
// order-independent), nothing to zero but that histogram.
// ---------------------------------------------------------------------------
constexpr int RX_BINS = 256, RX_BLOCKS_MAX = 1024;
__global__ __launch_bounds__(256) void radix_hist_kernel(const int* __restrict__ pos_in, const int* __restrict__ ri, int64_t n, int shift,
                                                         unsigned int* __restrict__ part /* gridDim.x x 256 */) {
    __shared__ unsigned int sh[RX_BINS];
    sh[threadIdx.x] = 0;
    __syncthreads();
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t i0 = (int64_t)blockIdx.x * per;
    const int64_t i1 = i0 + per < n ? i0 + per : n;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
        const int e = pos_in ? pos_in[i] : (int)i;
        atomicAdd(&sh[((unsigned)ri[e] >> shift) & 255u], 1u);
    }
    __syncthreads();
    part[(size_t)blockIdx.x * RX_BINS + threadIdx.x] = sh[threadIdx.x];
}
__global__ __launch_bounds__(256) void radix_scatter_kernel(const int* __restrict__ pos_in, const int* __restrict__ ri, int64_t n, int shift,
                                                            const unsigned int* __restrict__ part, int* __restrict__ pos_out) {
    __shared__ unsigned int tot[RX_BINS], base[RX_BINS], wcnt[4][RX_BINS];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t i0 = (int64_t)blockIdx.x * per;
    const int64_t i1 = i0 + per < n ? i0 + per : n;
    const int64_t wper = (per + 3) / 4;                       // this wavefront's contiguous quarter [j0, j1)
    const int64_t j0 = i0 + w * wper < i1 ? i0 + w * wper : i1;
    const int64_t j1 = j0 + wper < i1 ? j0 + wper : i1;
    for (int v = 0; v < 4; ++v) wcnt[v][threadIdx.x] = 0;
    {   // digit x = threadIdx.x: total over all blocks, and the part of it in the blocks before this one
        unsigned int t = 0, before = 0;
        for (unsigned int b = 0; b < gridDim.x; ++b) {
            const unsigned int c = part[(size_t)b * RX_BINS + threadIdx.x];
            before += b < blockIdx.x ? c : 0u;
            t += c;
        }
        tot[threadIdx.x] = t;
        base[threadIdx.x] = before;
    }
    __syncthreads();
    for (int64_t i = j0 + lane; i < j1; i += 64) {            // per-wavefront digit counts of the quarter
        const int e = pos_in ? pos_in[i] : (int)i;
        atomicAdd(&wcnt[w][((unsigned)ri[e] >> shift) & 255u], 1u);
    }
    if (threadIdx.x == 0) {                                    // exclusive scan of the totals over the 256 digits
        unsigned int run = 0;
        for (int x = 0; x < RX_BINS; ++x) { const unsigned int t = tot[x]; tot[x] = run; run += t; }
    }
    __syncthreads();
    {   // wcnt[v][x] <- where wavefront v's first element of digit x goes
        unsigned int run = tot[threadIdx.x] + base[threadIdx.x];
        for (int v = 0; v < 4; ++v) { const unsigned int t = wcnt[v][threadIdx.x]; wcnt[v][threadIdx.x] = run; run += t; }
    }
    __syncthreads();
    volatile unsigned int* wb = wcnt[w];                       // this wavefront's running digit starts (one wavefront = program order)
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int64_t r = j0; r < j1; r += 64) {
        const int64_t i = r + lane;
        const bool valid = i < j1;
        int e = 0;
        unsigned int dg = 0;
        if (valid) {
            e = pos_in ? pos_in[i] : (int)i;
            dg = ((unsigned)ri[e] >> shift) & 255u;
        }
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const unsigned long long bm = __ballot((dg >> bit) & 1u);
            same &= ((dg >> bit) & 1u) ? bm : ~bm;
        }
        if (valid) {
            const unsigned int start = wb[dg];                  // every lane of the group reads the same word ...
            pos_out[start + (unsigned int)__popcll(same & lt)] = e;
            if ((same & lt) == 0ull) wb[dg] = start + (unsigned int)__popcll(same);      // ... its lowest lane moves it on
        }
    }
}
__global__ void row_count_kernel(const int* __restrict__ ri, int64_t nnz, int* __restrict__ rowcnt) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nnz; t += (int64_t)gridDim.x * blockDim.x) atomicAdd(&rowcnt[ri[t]], 1);
}

// column of the nonzero at position pos: largest j with p[j] <= pos
__device__ __forceinline__ int col_of(const int* __restrict__ p, int cols, int pos) {
    int lo = 0, hi = cols;      // invariant: p[lo] <= pos < p[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (p[mid] <= pos) lo = mid; else hi = mid;
    }
    return lo;
}
template <class T>
__global__ void transpose_gather_kernel(const int* __restrict__ p, int cols, const int* __restrict__ pos, const T* __restrict__ x,
                                        int64_t nnz, int* __restrict__ ti, T* __restrict__ tx) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nnz; t += (int64_t)gridDim.x * blockDim.x) {
        const int q = pos[t];
        ti[t] = col_of(p, cols, q);
        if (tx) tx[t] = x[q];
    }
}
template <class T>
__global__ void gather_values_kernel(const int* __restrict__ pos, const T* __restrict__ x, int64_t nnz, T* __restrict__ tx) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nnz; t += (int64_t)gridDim.x * blockDim.x) tx[t] = x[pos[t]];
}
template <class S, class D>
__global__ void cast_kernel(const S* __restrict__ src, D* __restrict__ dst, int64_t n) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
        dst[t] = static_cast<D>(src[t]);
}

unsigned grid_for(int64_t n, int num_cu) {
    int64_t b = (n + 255) / 256;
    const int64_t cap = (int64_t)num_cu * 16;
    if (b > cap) b = cap;
    return (unsigned)(b < 1 ? 1 : b);
}

// The transpose in two steps, so that a caller can overlap the upload of the VALUES with the sort (which needs the row
// indices only): begin = positions sorted by row + row pointers, finish = column indices and values gathered through the
// sorted positions.  The sorted positions live in `pos_out` (nnz ints, caller-provided).
void transpose_sort(rcppml_hip_ctx* c, int rows, int cols, int64_t nnz, const int* p, const int* ri, int* tp, int* pos_out) {
    HIPCHK(hipMemsetAsync(tp, 0, ((size_t)rows + 1) * sizeof(int), c->stream));
    if (nnz == 0 || rows <= 0 || cols <= 0) return;
    if (rows > TR_RCH) {          // tall: stable LSD radix sort of the positions by row index (above)
        int bits = 1;
        while (((int64_t)1 << bits) < rows) ++bits;
        const int passes = (bits + 7) / 8;
        int64_t nblk = (nnz + 4095) / 4096;
        if (nblk > RX_BLOCKS_MAX) nblk = RX_BLOCKS_MAX;
        if (nblk < 1) nblk = 1;
        DevTmp part(c, (size_t)nblk * RX_BINS * sizeof(unsigned int)), other(c, (size_t)nnz * sizeof(int)), rowcnt(c, ((size_t)rows + 1) * sizeof(int));
        // ping-pong so that the LAST pass writes pos_out
        int* bufs[2] = {passes % 2 ? pos_out : (int*)other.p, passes % 2 ? (int*)other.p : pos_out};
        const int* in = nullptr;
        for (int ps = 0; ps < passes; ++ps) {
            int* out = bufs[ps % 2];
            hipLaunchKernelGGL(radix_hist_kernel, dim3((unsigned)nblk), dim3(256), 0, c->stream, in, ri, nnz, 8 * ps, (unsigned int*)part.p);
            hipLaunchKernelGGL(radix_scatter_kernel, dim3((unsigned)nblk), dim3(256), 0, c->stream, in, ri, nnz, 8 * ps, (const unsigned int*)part.p, out);
            HIPCHK(hipGetLastError());
            in = out;
        }
        HIPCHK(hipMemsetAsync(rowcnt.p, 0, ((size_t)rows + 1) * sizeof(int), c->stream));
        hipLaunchKernelGGL(row_count_kernel, dim3(grid_for(nnz, c->num_cu)), dim3(256), 0, c->stream, ri, nnz, (int*)rowcnt.p);
        HIPCHK(hipGetLastError());
        rk::exclusive_scan_i32(c, (const int*)rowcnt.p, tp, (int64_t)rows + 1);
        if (part.owned || other.owned || rowcnt.owned) HIPCHK(hipStreamSynchronize(c->stream));
        return;
    }
    // chunks: enough workgroups to fill the chip, the B x rows table at most 256 MB
    int B = cols < 512 ? cols : 512;
    const int64_t cap = (int64_t)(256u << 20) / ((int64_t)rows * 4);
    if (B > cap) B = cap < 1 ? 1 : (int)cap;
    if (B < 1) B = 1;
    DevTmp table(c, (size_t)B * rows * sizeof(int)), rowcnt(c, ((size_t)rows + 1) * sizeof(int));
    const size_t lds = (size_t)(rows < TR_RCH ? rows : TR_RCH) * sizeof(int);
    static DynSmemOnce once_count, once_scatter;
    once_count.ensure(reinterpret_cast<const void*>(transpose_count_kernel), lds, c->device);
    once_scatter.ensure(reinterpret_cast<const void*>(transpose_scatter_kernel), lds, c->device);
    hipLaunchKernelGGL(transpose_count_kernel, dim3(B), dim3(256), lds, c->stream, p, ri, cols, rows, (int*)table.p);
    hipLaunchKernelGGL(transpose_prefix_kernel, dim3((unsigned)((rows + 1 + 255) / 256)), dim3(256), 0, c->stream, (int*)table.p, rows, B,
                       (int*)rowcnt.p);
    HIPCHK(hipGetLastError());
    rk::exclusive_scan_i32(c, (const int*)rowcnt.p, tp, (int64_t)rows + 1);          // row pointers (the last one = nnz)
    hipLaunchKernelGGL(transpose_scatter_kernel, dim3(B), dim3(256), lds, c->stream, p, ri, cols, rows, (const int*)table.p,
                       (const int*)tp, pos_out);
    HIPCHK(hipGetLastError());
    // temporaries from the context's per-fit arena outlive this call; hipMalloc'ed ones must not be freed under the kernels
    if (table.owned || rowcnt.owned) HIPCHK(hipStreamSynchronize(c->stream));
}
template <class T>
void transpose_gather(rcppml_hip_ctx* c, int cols, int64_t nnz, const int* p, const int* pos, const T* x, int* ti, T* tx) {
    if (nnz == 0) return;
    // ti == NULL: values only (the column indices were gathered by an earlier call with values == NULL, before the values arrived)
    if (ti) hipLaunchKernelGGL(transpose_gather_kernel<T>, dim3(grid_for(nnz, c->num_cu)), dim3(256), 0, c->stream, p, cols, pos, x, nnz, ti, tx);
    else if (tx) hipLaunchKernelGGL(gather_values_kernel<T>, dim3(grid_for(nnz, c->num_cu)), dim3(256), 0, c->stream, pos, x, nnz, tx);
    HIPCHK(hipGetLastError());
}

template <class T>
void transpose_impl(rcppml_hip_ctx* c, int rows, int cols, const int* p, const int* ri, const T* x, int* tp, int* ti, T* tx) {
    int nnz_i = 0;
    HIPCHK(hipMemcpyAsync(&nnz_i, p + cols, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    const int64_t nnz = nnz_i;
    DevTmp pos_out(c, (size_t)std::max<int64_t>(nnz, 1) * sizeof(int));
    transpose_sort(c, rows, cols, nnz, p, ri, tp, static_cast<int*>(pos_out.p));
    transpose_gather<T>(c, cols, nnz, p, static_cast<const int*>(pos_out.p), x, ti, tx);
    HIPCHK(hipStreamSynchronize(c->stream));       // temporaries die here
}

}  // namespace

extern "C" int rcppml_hip_transpose_csc(rcppml_hip_ctx* c, int dtype, int rows, int cols, const int* col_ptr,
                                        const int* row_idx, const void* values, int* t_col_ptr, int* t_row_idx,
                                        void* t_values) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (rows < 0 || cols < 0) throw std::runtime_error("transpose_csc: negative dimension");
        if (dtype == RCPPML_F32)
            transpose_impl<float>(c, rows, cols, col_ptr, row_idx, (const float*)values, t_col_ptr, t_row_idx, (float*)t_values);
        else
            transpose_impl<double>(c, rows, cols, col_ptr, row_idx, (const double*)values, t_col_ptr, t_row_idx, (double*)t_values);
        return 0;
    }
    RCPPML_CATCH_RET
}

extern "C" int rcppml_hip_transpose_csc_sort(rcppml_hip_ctx* c, int rows, int cols, int64_t nnz, const int* col_ptr, const int* row_idx,
                                             int* t_col_ptr, int* sorted_pos) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (rows < 0 || cols < 0 || nnz < 0) throw std::runtime_error("transpose_csc_sort: negative dimension");
        transpose_sort(c, rows, cols, nnz, col_ptr, row_idx, t_col_ptr, sorted_pos);
        return 0;
    }
    RCPPML_CATCH_RET
}
extern "C" int rcppml_hip_transpose_csc_gather(rcppml_hip_ctx* c, int dtype, int cols, int64_t nnz, const int* col_ptr,
                                               const int* sorted_pos, const void* values, int* t_row_idx, void* t_values) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (dtype == RCPPML_F32) transpose_gather<float>(c, cols, nnz, col_ptr, sorted_pos, (const float*)values, t_row_idx, (float*)t_values);
        else transpose_gather<double>(c, cols, nnz, col_ptr, sorted_pos, (const double*)values, t_row_idx, (double*)t_values);
        return 0;
    }
    RCPPML_CATCH_RET
}

// dst[t] = (dst type) src[t]; dtype_src / dtype_dst in {RCPPML_F32, RCPPML_F64}
extern "C" int rcppml_hip_cast(rcppml_hip_ctx* c, int dtype_src, const void* src, int dtype_dst, void* dst, int64_t n) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (n <= 0) return 0;
        const dim3 grid(grid_for(n, c->num_cu)), block(256);
        if (dtype_src == RCPPML_F64 && dtype_dst == RCPPML_F32)
            hipLaunchKernelGGL((cast_kernel<double, float>), grid, block, 0, c->stream, (const double*)src, (float*)dst, n);
        else if (dtype_src == RCPPML_F32 && dtype_dst == RCPPML_F64)
            hipLaunchKernelGGL((cast_kernel<float, double>), grid, block, 0, c->stream, (const float*)src, (double*)dst, n);
        else if (dtype_src == dtype_dst)
            HIPCHK(hipMemcpyAsync(dst, src, (size_t)n * (dtype_src == RCPPML_F32 ? 4 : 8), hipMemcpyDeviceToDevice, c->stream));
        else
            throw std::runtime_error("cast: unknown dtype");
        HIPCHK(hipGetLastError());
        return 0;
    }
    RCPPML_CATCH_RET
}
