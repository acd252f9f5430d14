// ops_dense.hip -- dense-input right-hand sides of the ALS update (device-level C ABI, include/rcppml_gpu.h layer 2).
// Reference: primitives::rhs<CPU> on a dense A (B = W_T * A, Eigen GEMM) and detail::rhs_transpose (B = H * A^T),
// nmf/fit_cpu.hpp:547-549 / :783; the reference's GPU build calls cuBLAS for them (nmf/fit_gpu_dense.cuh).
// Hand-written skinny MFMA GEMMs (kernels_dense.hip.h: fp32 on 32x32x2 tiles, fp64 on 16x16x4 tiles) that stream A once per
// product; no BLAS library is linked (tools/probe/dense_check.py times torch.matmul = rocBLAS/hipBLASLt beside them).  Everything around them (Gram, features,
// NNLS solve, scaling, loss) is the hand-written path shared with the sparse input.
#include "common.hip.h"
#include "kernels_dense.hip.h"
#include <cstring>


// transposed = 0:  B (k x n) = F (k x m) * A (m x n)       transposed = 1:  B (k x m) = F (k x n) * A^T
// A is column-major m x n; F and B are column-major with leading dimension k.
extern "C" int rcppml_hip_rhs_dense(rcppml_hip_ctx* c, int dtype, const void* A, int64_t m, int64_t n, int transposed,
                                    const void* F, int k, void* B) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (m <= 0 || n <= 0 || k <= 0) return 0;
        if (m > 0x7FFFFFFF || n > 0x7FFFFFFF) throw std::runtime_error("rhs_dense: dimension exceeds int32");
        if (k > 128) throw std::runtime_error("rhs_dense: k must be <= 128");
        if (dtype == RCPPML_F32) {
            const float* Af = (const float*)A; const float* Ff = (const float*)F; float* Bf = (float*)B;
            const int RT = (k + 31) / 32;
            if (!transposed) {
                // 4 waves per SIMD in flight: split the reduction over i when n alone does not supply them
                const int64_t col_blocks = (n + 127) / 128;
                int64_t slices = (4096 + col_blocks * 4 - 1) / (col_blocks * 4);
                const int64_t chunks = (m + rk::DENSE_KC - 1) / rk::DENSE_KC;
                const char* es = exp_env("RCPPML_GPU_DENSE_SLICES");
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
        {
            const double* Ad = (const double*)A; const double* Fd = (const double*)F; double* Bd = (double*)B;
            const int RT = k <= 16 ? 1 : k <= 32 ? 2 : k <= 64 ? 4 : 8;
            const int64_t out_cols = transposed ? m : n, red = transposed ? n : m;
            const int64_t per_block = transposed ? 128 : 64, kc = transposed ? rk::DENSE_KC_BWD : rk::DENSE_KC;
            const int64_t blocks = (out_cols + per_block - 1) / per_block, chunks = (red + kc - 1) / kc;
            int64_t slices = (4096 + blocks * 4 - 1) / (blocks * 4);
            if (slices > chunks) slices = chunks;
            if (slices < 1) slices = 1;
            const int64_t rchunk = (chunks + slices - 1) / slices * kc;
            slices = (red + rchunk - 1) / rchunk;
            double* part = slices == 1 ? Bd : static_cast<double*>(c->scratch(WS_GRAPH, (size_t)slices * k * out_cols * sizeof(double)));
            const dim3 grid((unsigned)blocks, (unsigned)slices);
#define RCPPML_DENSE64(KERN)                                                                                                   \
            switch (RT) {                                                                                                      \
                case 1: hipLaunchKernelGGL((rk::KERN<1>), grid, dim3(256), 0, c->stream, Ad, m, n, Fd, k, rchunk, part); break; \
                case 2: hipLaunchKernelGGL((rk::KERN<2>), grid, dim3(256), 0, c->stream, Ad, m, n, Fd, k, rchunk, part); break; \
                case 4: hipLaunchKernelGGL((rk::KERN<4>), grid, dim3(256), 0, c->stream, Ad, m, n, Fd, k, rchunk, part); break; \
                default: hipLaunchKernelGGL((rk::KERN<8>), grid, dim3(256), 0, c->stream, Ad, m, n, Fd, k, rchunk, part); break; \
            }
            if (!transposed) { RCPPML_DENSE64(dense_rhs_fwd_f64) } else { RCPPML_DENSE64(dense_rhs_bwd_f64) }
#undef RCPPML_DENSE64
            HIPCHK(hipGetLastError());
            if (slices > 1) {
                const int64_t count = (int64_t)k * out_cols;
                hipLaunchKernelGGL(rk::dense_reduce<double>, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream, part, count, (int)slices, Bd);
                HIPCHK(hipGetLastError());
            }
            return 0;
        }
    }
    RCPPML_CATCH_RET
}
