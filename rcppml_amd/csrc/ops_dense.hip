// ops_dense.hip -- dense-input right-hand sides of the ALS update (device-level C ABI, include/rcppml_gpu.h layer 2).
// Reference: primitives::rhs<CPU> on a dense A (B = W_T * A, Eigen GEMM) and detail::rhs_transpose (B = H * A^T),
// nmf/fit_cpu.hpp:547-549 / :783; the reference's GPU build calls cuBLAS for them (nmf/fit_gpu_dense.cuh).  These are
// plain k x (m or n) x (n or m) library GEMMs with nothing to fuse into them, so they go to rocBLAS; everything around
// them (Gram, features, NNLS solve, scaling, loss) is the hand-written path shared with the sparse input.
#include <rocblas/rocblas.h>
#include "common.hip.h"

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
