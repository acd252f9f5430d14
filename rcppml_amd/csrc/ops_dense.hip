// ops_dense.hip -- dense-input right-hand sides of the ALS update (device-level C ABI, include/rcppml_gpu.h layer 2).
// Reference: primitives::rhs<CPU> on a dense A (B = W_T * A, Eigen GEMM) and detail::rhs_transpose (B = H * A^T),
// nmf/fit_cpu.hpp:547-549 / :783; the reference's GPU build calls cuBLAS for them (nmf/fit_gpu_dense.cuh).  These are
// fp32 (the precision the reference computes in): hand-written skinny MFMA GEMMs (kernels_dense.hip.h) that stream A once
// per product; fp64 (the parity mode) and RCPPML_GPU_DENSE_VARIANT=blas: rocBLAS.  Everything around them (Gram, features,
// NNLS solve, scaling, loss) is the hand-written path shared with the sparse input.
#include <rocblas/rocblas.h>
#include "common.hip.h"
#include "kernels_dense.hip.h"
#include <cstring>

namespace {
rocblas_handle blas_of(rcppml_hip_ctx* c) {
    if (!c->blas) {
        rocblas_handle h = nullptr;
        if (rocblas_create_handle(&h) != rocblas_status_success) throw std::runtime_error("rocblas_create_handle failed");
        if (rocblas_set_stream(h, c->stream) != rocblas_status_success) { rocblas_destroy_handle(h); throw std::runtime_error("rocblas_set_stream failed"); }
        c->blas = h;
        c->blas_destroy = [](void* p) { (void)rocblas_destroy_handle(static_cast<rocblas_handle>(p)); };
    }
    return static_cast<rocblas_handle>(c->blas);
}
}  // namespace

// transposed = 0:  B (k x n) = F (k x m) * A (m x n)       transposed = 1:  B (k x m) = F (k x n) * A^T
// A is column-major m x n; F and B are column-major with leading dimension k.
extern "C" int rcppml_hip_rhs_dense(rcppml_hip_ctx* c, int dtype, const void* A, int64_t m, int64_t n, int transposed,
                                    const void* F, int k, void* B) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (m <= 0 || n <= 0 || k <= 0) return 0;
        if (m > 0x7FFFFFFF || n > 0x7FFFFFFF) throw std::runtime_error("rhs_dense: dimension exceeds int32");
        static int use_blas = -1;
        if (use_blas < 0) { const char* e = getenv("RCPPML_GPU_DENSE_VARIANT"); use_blas = (e && !strcmp(e, "blas")) ? 1 : 0; }
        if (dtype == RCPPML_F32 && k <= 128 && !use_blas) {
            const float* Af = (const float*)A; const float* Ff = (const float*)F; float* Bf = (float*)B;
            const int RT = (k + 31) / 32;
            if (!transposed) {
                // 4 waves per SIMD in flight: split the reduction over i when n alone does not supply them
                const int64_t col_blocks = (n + 127) / 128;
                int64_t slices = (4096 + col_blocks * 4 - 1) / (col_blocks * 4);
                const int64_t chunks = (m + rk::DENSE_KC - 1) / rk::DENSE_KC;
                const char* es = getenv("RCPPML_GPU_DENSE_SLICES");
                if (es) slices = atoi(es);
                if (slices > chunks) slices = chunks;
                if (slices < 1) slices = 1;
                const int64_t ichunk = (chunks + slices - 1) / slices * rk::DENSE_KC;
                slices = (m + ichunk - 1) / ichunk;
                float* part = slices == 1 ? Bf : static_cast<float*>(c->scratch(WS_GRAPH, (size_t)slices * k * n * sizeof(float)));
                const dim3 grid((unsigned)col_blocks, (unsigned)slices);
                switch (RT) {
                    case 1: hipLaunchKernelGGL((rk::dense_rhs_fwd_f32<1>), grid, dim3(256), 0, c->stream, Af, m, n, Ff, k, ichunk, part); break;
                    case 2: hipLaunchKernelGGL((rk::dense_rhs_fwd_f32<2>), grid, dim3(256), 0, c->stream, Af, m, n, Ff, k, ichunk, part); break;
                    case 3: hipLaunchKernelGGL((rk::dense_rhs_fwd_f32<3>), grid, dim3(256), 0, c->stream, Af, m, n, Ff, k, ichunk, part); break;
                    default: hipLaunchKernelGGL((rk::dense_rhs_fwd_f32<4>), grid, dim3(256), 0, c->stream, Af, m, n, Ff, k, ichunk, part); break;
                }
                if (slices > 1) {
                    HIPCHK(hipGetLastError());
                    const int64_t count = (int64_t)k * n;
                    hipLaunchKernelGGL(rk::dense_reduce_f32, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream, part, count, (int)slices, Bf);
                }
            } else {
                // split the reduction over j so that about 2 waves per SIMD are in flight; slices are multiples of the chunk
                const int64_t row_blocks = (m + 511) / 512;
                int64_t slices = (4096 + row_blocks * 4 - 1) / (row_blocks * 4);
                const int64_t chunks = (n + rk::DENSE_KC_BWD - 1) / rk::DENSE_KC_BWD;
                if (slices > chunks) slices = chunks;
                if (slices < 1) slices = 1;
                const int64_t jchunk = (chunks + slices - 1) / slices * rk::DENSE_KC_BWD;
                slices = (n + jchunk - 1) / jchunk;
                float* part = slices == 1 ? Bf : static_cast<float*>(c->scratch(WS_GRAPH, (size_t)slices * k * m * sizeof(float)));
                const dim3 grid((unsigned)row_blocks, (unsigned)slices);
                switch (RT) {
                    case 1: hipLaunchKernelGGL((rk::dense_rhs_bwd_f32<1>), grid, dim3(256), 0, c->stream, Af, m, n, Ff, k, jchunk, part); break;
                    case 2: hipLaunchKernelGGL((rk::dense_rhs_bwd_f32<2>), grid, dim3(256), 0, c->stream, Af, m, n, Ff, k, jchunk, part); break;
                    case 3: hipLaunchKernelGGL((rk::dense_rhs_bwd_f32<3>), grid, dim3(256), 0, c->stream, Af, m, n, Ff, k, jchunk, part); break;
                    default: hipLaunchKernelGGL((rk::dense_rhs_bwd_f32<4>), grid, dim3(256), 0, c->stream, Af, m, n, Ff, k, jchunk, part); break;
                }
                if (slices > 1) {
                    HIPCHK(hipGetLastError());
                    const int64_t count = (int64_t)k * m;
                    hipLaunchKernelGGL(rk::dense_reduce_f32, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream, part, count, (int)slices, Bf);
                }
            }
            HIPCHK(hipGetLastError());
            return 0;
        }
        rocblas_handle h = blas_of(c);
        const rocblas_operation tb = transposed ? rocblas_operation_transpose : rocblas_operation_none;
        const rocblas_int N = (rocblas_int)(transposed ? m : n), K = (rocblas_int)(transposed ? n : m);
        rocblas_status st;
        if (dtype == RCPPML_F32) {
            const float one = 1.f, zero = 0.f;
            st = rocblas_sgemm(h, rocblas_operation_none, tb, k, N, K, &one, (const float*)F, k, (const float*)A, (rocblas_int)m, &zero, (float*)B, k);
        } else {
            const double one = 1.0, zero = 0.0;
            st = rocblas_dgemm(h, rocblas_operation_none, tb, k, N, K, &one, (const double*)F, k, (const double*)A, (rocblas_int)m, &zero, (double*)B, k);
        }
        if (st != rocblas_status_success) throw std::runtime_error(std::string("rocblas gemm failed: ") + rocblas_status_to_string(st));
        return 0;
    }
    RCPPML_CATCH_RET
}
