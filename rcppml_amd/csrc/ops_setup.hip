// ops_setup.hip -- one-time setup of a fit on the device: CSC transpose (A^T for the W half-update, reference
// nmf/fit_cpu.hpp:251-253 `At = A.transpose()`) and precision casts of the host's double buffers.
//
// The transpose is a STABLE radix sort of the nonzero positions by row index (rocPRIM through hipCUB: library sort, not
// a hand-written one -- it runs once per fit): positions of one row stay in increasing order, i.e. column indices of
// A^T come out sorted exactly as Eigen's transpose produces them.  Row pointers are a histogram + exclusive scan.
#include <hipcub/hipcub.hpp>
#include "common.hip.h"

namespace {

__global__ void iota_kernel(int* __restrict__ v, int64_t n) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) v[t] = (int)t;
}
__global__ void row_hist_kernel(const int* __restrict__ ri, int64_t nnz, int* __restrict__ counts) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nnz; t += (int64_t)gridDim.x * blockDim.x)
        atomicAdd(&counts[ri[t]], 1);
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
void transpose_sort(rcppml_hip_ctx* c, int rows, int cols, int64_t nnz, const int* ri, int* tp, int* pos_out) {
    HIPCHK(hipMemsetAsync(tp, 0, ((size_t)rows + 1) * sizeof(int), c->stream));
    if (nnz == 0) return;
    DevTmp keys_out(c, (size_t)nnz * sizeof(int)), pos_in(c, (size_t)nnz * sizeof(int));
    DevTmp counts(c, ((size_t)rows + 1) * sizeof(int));
    int* kout = static_cast<int*>(keys_out.p);
    int* pin = static_cast<int*>(pos_in.p);
    int* cnt = static_cast<int*>(counts.p);
    hipLaunchKernelGGL(iota_kernel, dim3(grid_for(nnz, c->num_cu)), dim3(256), 0, c->stream, pin, nnz);
    HIPCHK(hipGetLastError());
    int end_bit = 1;
    while ((1ll << end_bit) < rows) ++end_bit;
    size_t tmp_bytes = 0;
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, ri, kout, pin, pos_out, (int)nnz, 0, end_bit, c->stream));
    DevTmp tmp(c, tmp_bytes);
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, ri, kout, pin, pos_out, (int)nnz, 0, end_bit, c->stream));
    // row pointers: counts -> exclusive scan (rows + 1 entries, the last one = nnz)
    HIPCHK(hipMemsetAsync(cnt, 0, ((size_t)rows + 1) * sizeof(int), c->stream));
    hipLaunchKernelGGL(row_hist_kernel, dim3(grid_for(nnz, c->num_cu)), dim3(256), 0, c->stream, ri, nnz, cnt);
    HIPCHK(hipGetLastError());
    size_t scan_bytes = 0;
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, cnt, tp, rows + 1, c->stream));
    DevTmp stmp(c, scan_bytes);
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(stmp.p, scan_bytes, cnt, tp, rows + 1, c->stream));
    // temporaries from the context's per-fit arena outlive this call; hipMalloc'ed ones must not be freed under the kernels
    if (keys_out.owned || pos_in.owned || counts.owned || tmp.owned || stmp.owned) HIPCHK(hipStreamSynchronize(c->stream));
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
    transpose_sort(c, rows, cols, nnz, ri, tp, static_cast<int*>(pos_out.p));
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

extern "C" int rcppml_hip_transpose_csc_sort(rcppml_hip_ctx* c, int rows, int cols, int64_t nnz, const int* row_idx, int* t_col_ptr,
                                             int* sorted_pos) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (rows < 0 || cols < 0 || nnz < 0) throw std::runtime_error("transpose_csc_sort: negative dimension");
        transpose_sort(c, rows, cols, nnz, row_idx, t_col_ptr, sorted_pos);
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
