// ops_small.hip -- launcher of the one-kernel fit of small sparse matrices (kernels_small.hip.h): eligibility rule, scratch, launch.
#include "common.hip.h"
#include "kernels_small.hip.h"

namespace {

using namespace rk;

// What the persistent kernel CAN take (scratch and index ranges): ranks up to 32, 2^20 nonzeros, 65 536 columns on both sides together.
bool small_can(int m, int n, int64_t nnz, int k) {
    return k >= 1 && k <= 32 && m >= 1 && n >= 1 && nnz <= ((int64_t)1 << 20) && (int64_t)m + n <= 65536;
}
// ... and where it PAYS -- measured, tools/small_threshold.py -> profiles/r06_small_threshold.txt (us per iteration against graph replays of
// the multi-launch iteration): with k <= 16 (four columns per wavefront) 2.0-2.3 x at 700 .. 1 400 columns, 1.2-1.5 x at 2 800 columns /
// 76 000 nonzeros, break-even at 4 000 columns / 205 000 nonzeros (three rounds of 1 536 columns per half-update); with 16 < k <= 32 (one
// column per wavefront, 32 coordinates per sweep) it loses at every size (0.75-0.9 x at 700 columns, 0.2-0.4 x at 7 500).  So: k <= 16,
// at most 3 072 columns on both sides together and 2^17 nonzeros (hawaiibirds: 1 366 columns, 30 815 nonzeros; movielens has k = 32).
bool small_ok(int m, int n, int64_t nnz, int k) {
    if (!small_can(m, n, nnz, k)) return false;
    return k <= 16 && (int64_t)m + n <= 3072 && nnz <= (int64_t)1 << 17;
}

template <class T, int KP, bool CHOL>
void launch(rcppml_hip_ctx* c, const SmallFit<T>& P) {
    hipLaunchKernelGGL((als_small_kernel<T, KP, CHOL>), dim3(8 * SM_NB), dim3(64 * SM_WPB), 0, c->stream, P);
    HIPCHK(hipGetLastError());
}

template <class T>
void run(rcppml_hip_ctx* c, const int* Ap, const int* Ai, const void* Ax, const int* Tp, const int* Ti, const void* Tx, int m, int n, int k,
         void* W, void* H, void* d, const double* trAtA, double L1_H, double L1_W, double L2_H, double L2_W, double ub_H, double ub_W, int nonneg_H,
         int nonneg_W, int norm_type, int solver_mode, int cd_maxit, double cd_tol, int max_iter, double tol, int patience, int iter0, double* loss_hist,
         double* result) {
    const int KP = k <= 16 ? 16 : 32;
    const size_t PS = (size_t)KP + (size_t)KP * KP;
    // scratch: Bw | part | cross partials | sync words (one block, zeroed sync)
    const size_t b_bw = (((size_t)k * m * sizeof(T)) + 255) & ~(size_t)255;
    const size_t b_part = ((SM_NB * PS * sizeof(T)) + 255) & ~(size_t)255;
    const size_t b_cross = 256;
    char* blk = (char*)c->scratch(WS_FEAT, b_bw + b_part + b_cross + 256);
    SmallFit<T> P;
    P.Ap = Ap; P.Ai = Ai; P.Ax = (const T*)Ax; P.Tp = Tp; P.Ti = Ti; P.Tx = (const T*)Tx;
    P.m = m; P.n = n; P.k = k;
    P.W = (T*)W; P.H = (T*)H; P.d = (T*)d;
    P.Bw = (T*)blk; P.part = (T*)(blk + b_bw); P.crossp = (double*)(blk + b_bw + b_part); P.sync = (unsigned*)(blk + b_bw + b_part + b_cross);
    P.trAtA = trAtA;
    P.L1_H = (T)L1_H; P.L1_W = (T)L1_W; P.L2_H = (T)L2_H; P.L2_W = (T)L2_W; P.ub_H = (T)ub_H; P.ub_W = (T)ub_W; P.cd_tol = (T)cd_tol;
    P.nonneg_H = nonneg_H; P.nonneg_W = nonneg_W; P.norm_type = norm_type; P.cd_maxit = cd_maxit;
    P.max_iter = max_iter; P.patience = patience; P.tol = tol; P.iter0 = iter0;
    P.loss_hist = loss_hist; P.result = result;
    HIPCHK(hipMemsetAsync(P.sync, 0, 256, c->stream));
    if (c->opt_small_give_up) HIPCHK(hipMemsetAsync(P.sync + 2, 1, 1, c->stream));      // test switch: abort flag preset (low byte = 1)
    HIPCHK(hipMemsetAsync(result, 0, 8 * sizeof(double), c->stream));
    const bool chol = solver_mode == 1;
    if (KP == 16) { if (chol) launch<T, 16, true>(c, P); else launch<T, 16, false>(c, P); }
    else { if (chol) launch<T, 32, true>(c, P); else launch<T, 32, false>(c, P); }
}

}  // namespace

extern "C" int rcppml_hip_als_small_eligible(int m, int n, int64_t nnz, int k) { return small_ok(m, n, nnz, k) ? 1 : 0; }

extern "C" int rcppml_hip_als_small_fit(rcppml_hip_ctx* c, int dtype, const int* col_ptr, const int* row_idx, const void* values,
                                        const int* t_col_ptr, const int* t_row_idx, const void* t_values, int m, int n, int64_t nnz, int k,
                                        void* W, void* H, void* d, const double* trAtA, double L1_H, double L1_W, double L2_H, double L2_W,
                                        double ub_H, double ub_W, int nonneg_H, int nonneg_W, int norm_type, int solver_mode, int cd_maxit,
                                        double cd_tol, int max_iter, double tol, int patience, int iter0, double* loss_history, double* result8) {
    try {
        HIPCHK(hipSetDevice(c->device));
        if (!small_can(m, n, nnz, k)) throw std::runtime_error("als_small_fit: k <= 32, nnz <= 2^20, m + n <= 65536 (rcppml_hip_als_small_eligible says where the kernel pays)");
        if (solver_mode != 0 && solver_mode != 1) throw std::runtime_error("als_small_fit: solver_mode must be 0 or 1");
        if (norm_type < 0 || norm_type > 2) throw std::runtime_error("als_small_fit: bad norm_type");
        if (max_iter < 1 || iter0 < 0) throw std::runtime_error("als_small_fit: max_iter < 1 or iter0 < 0");
        if (dtype == RCPPML_F32)
            run<float>(c, col_ptr, row_idx, values, t_col_ptr, t_row_idx, t_values, m, n, k, W, H, d, trAtA, L1_H, L1_W, L2_H, L2_W, ub_H, ub_W,
                       nonneg_H, nonneg_W, norm_type, solver_mode, cd_maxit, cd_tol, max_iter, tol, patience, iter0, loss_history, result8);
        else
            run<double>(c, col_ptr, row_idx, values, t_col_ptr, t_row_idx, t_values, m, n, k, W, H, d, trAtA, L1_H, L1_W, L2_H, L2_W, ub_H, ub_W,
                        nonneg_H, nonneg_W, norm_type, solver_mode, cd_maxit, cd_tol, max_iter, tol, patience, iter0, loss_history, result8);
        return 0;
    }
    RCPPML_CATCH_RET
}
