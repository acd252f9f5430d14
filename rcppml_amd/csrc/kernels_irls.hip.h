// ============================================================================
// kernels_irls.hip.h -- NB (negative-binomial) IRLS path of the ALS-NNLS NMF update for gfx950
// (BASELINE config 5).
//
//   irls_nb_solve_kernel  primitives/cpu/nnls_batch_irls.hpp:202-329,465-520 + math/loss.hpp:248-256
//   nb_size_rows_kernel   nmf/fit_cpu.hpp:1094-1265 (PER_ROW / GLOBAL, sparse branch)
//   nb_loss_kernel        nmf/explicit_loss.hpp:53-77 + math/loss.hpp:415-426
//
// One wavefront per column, lane r = feature r (k <= 64).  Per IRLS pass the wave walks the column's nonzeros once:
// the reconstruction f_i . x is a wave reduction, the NB weight is evaluated in fp64 as the reference does, and the
// weighted Gram  G_w = G + sum (w-1) f f^T  is accumulated ROW-WISE IN REGISTERS (lane r keeps row r, k VGPRs;
// f_c is broadcast with v_readlane) -- no LDS traffic in the O(nnz_j k^2) part.  G_w is then parked in LDS
// ([c][r], this wave's tile) for the coordinate-descent solve, which is the exact sequential sweep of
// cd_nnls_col_fixed with ballot skipping of no-op coordinates (all cd_maxit sweeps: the reference passes cd_tol = 0).
// ============================================================================
#pragma once
#include "kernels.hip.h"

namespace rk {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
template <class T> __device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const T o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// math/loss.hpp:248-256  irls_weight_nb: computed in double, eps = tiny_num<Scalar>() = Scalar(1e-15)
template <class T> __device__ __forceinline__ T irls_weight_nb_dev(T predicted, T nb_size) {
    double mu = static_cast<double>(predicted);
    const double eps = static_cast<double>(static_cast<T>(1e-15));
    mu = mu > eps ? mu : eps;
    double r = static_cast<double>(nb_size);
    r = r > 1e-10 ? r : 1e-10;
    double w = r / (mu * (r + mu));
    w = w < 1e6 ? w : 1e6;
    return static_cast<T>(w);
}

template <class T, int KP>   // KP in {32, 64}: features padded to KP (k <= KP), lane r = feature r
__global__ __launch_bounds__(256) void irls_nb_solve_kernel(
    const int* __restrict__ colptr, const int* __restrict__ rowidx, const T* __restrict__ vals, int64_t ncols,
    const T* __restrict__ F, const T* __restrict__ Gbase, T* __restrict__ X, int k, T l1, T l2, int nonneg,
    int cd_maxit, int irls_max_iter, T irls_tol, const T* __restrict__ theta_row, const T* __restrict__ theta_col) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    T* Gl = reinterpret_cast<T*>(smem_raw) + (size_t)wave * KP * KP;    // [c][r]
    const int64_t j = (int64_t)blockIdx.x * 4 + wave;
    if (j >= ncols) return;
    const bool fok = lane < k;
    const bool lin = lane < KP;                  // KP = 32: the upper half of the wave only takes part in shuffles
    const int ll = lin ? lane : 0;
    // base Gram row r in registers (padded with identity)
    T gb[KP];
#pragma unroll
    for (int c = 0; c < KP; ++c) gb[c] = (fok && c < k) ? Gbase[(int64_t)c * k + lane] : (c == lane ? T(1) : T(0));
    const int as = colptr[j], ae = colptr[j + 1];
    const T th_col = theta_col ? theta_col[j] : T(0);
    T x = T(0);                                   // nnls_batch_irls.hpp:482-483  H.setZero(): no warm start
    for (int irls = 0; irls < irls_max_iter; ++irls) {
        T gw[KP];
#pragma unroll
        for (int c = 0; c < KP; ++c) gw[c] = gb[c];
        T bw = T(0);
        for (int t = as; t < ae; ++t) {
            const int row = rowidx[t];
            const T a = vals[t];
            const T fr = fok ? F[(int64_t)row * k + lane] : T(0);
            const T recon = wave_sum(fr * x);                                   // W_T.col(row).dot(x)
            const T th = theta_col ? th_col : (theta_row ? theta_row[row] : T(0));
            const T w = irls_weight_nb_dev<T>(recon, th);
            const T dw = w - T(1);
            const T wv = w * a;
            const T frd = fr * dw;                                              // W_nnz_scaled.col = W_block.col * dw
#pragma unroll
            for (int c = 0; c < KP; ++c) {
                const T fc = __shfl(fr, c, 64);
                gw[c] = tfma(frd, fc, gw[c]);                                   // G_w(r,c) += (f_r dw) f_c
            }
            bw = tfma(fr, wv, bw);                                              // b_w += f * (w a)
        }
        if (l2 > T(0)) {
#pragma unroll
            for (int c = 0; c < KP; ++c)
                if (c == lane && fok) gw[c] += l2;
        }
        // park G_w in LDS; residual b_c = b_w - G_w x_old
        const T x_old = x;
        T b = bw;
#pragma unroll
        for (int c = 0; c < KP; ++c) {
            if (lin) Gl[c * KP + lane] = gw[c];
            const T xc = __shfl(x_old, c, 64);
            b = tfma(-gw[c], xc, b);
        }
        const T gd = Gl[ll * KP + ll];
        // cd_nnls_col_fixed(G_w, b_c, x, L1 inside, L2 = 0, nonneg, cd_maxit, ub = 0, tol = 0): all sweeps
        for (int it = 0; it < cd_maxit; ++it) {
            int cur = 0;
            bool any = false;
            while (true) {
                T diff = b / gd;
                if (l1 != T(0)) diff -= l1;
                const T nv = x + diff;
                T ad = diff, nx = nv;
                if (nonneg && nv < T(0)) { ad = -x; nx = T(0); }
                const bool moves = fok && (gd > T(0)) && (ad != T(0)) && (lane >= cur);
                const unsigned long long mask = __ballot(moves);
                if (mask == 0ull) break;
                any = true;
                const int i = __builtin_ctzll(mask);
                const T ad_i = __shfl(ad, i, 64), nx_i = __shfl(nx, i, 64);
                if (lane == i) x = nx_i;
                b = tfma(-Gl[i * KP + ll], ad_i, b);
                cur = i + 1;
                if (cur >= KP) break;
            }
            if (!any) break;      // a sweep without any effective step: all remaining sweeps are no-ops too
        }
        // IRLS convergence: max_i |x_i - x_old_i| / (|x_old_i| + 1e-12) < irls_tol
        T rel = fok ? tabs(x - x_old) / (tabs(x_old) + T(1e-12)) : T(0);
        rel = wave_max(rel);
        if (rel < irls_tol) break;
    }
    if (fok) X[j * (int64_t)k + lane] = x;
}

// NB size (r) method-of-moments update, one wavefront per ROW i of A (= column i of A^T):
//   nonzero sums of mu^2 and (y-mu)^2 in fp64, totals via  Wd_i . h_rs  and  Wd_i^T G_H Wd_i.
template <class T>
__global__ __launch_bounds__(256) void nb_size_rows_kernel(
    const int* __restrict__ tp, const int* __restrict__ ti, const T* __restrict__ tx, int64_t m,
    const T* __restrict__ W_T, const T* __restrict__ d, const T* __restrict__ H, const T* __restrict__ h_rs,
    const T* __restrict__ G_H, int k, double r_min, double r_max, T* __restrict__ nb_size) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    if (i >= m) return;
    const bool fok = lane < k;
    const T wd = fok ? W_T[i * (int64_t)k + lane] * d[lane] : T(0);     // apply_scaling(W_Td, d)
    double s_mu2 = 0.0, s_res2 = 0.0;
    for (int t = tp[i]; t < tp[i + 1]; ++t) {
        const int col = ti[t];
        const T hv = fok ? H[(int64_t)col * k + lane] : T(0);
        const T dot = wave_sum(wd * hv);
        const double y = static_cast<double>(tx[t]);
        double mu = static_cast<double>(dot);
        mu = mu > 1e-10 ? mu : 1e-10;
        const double resid = y - mu;
        s_mu2 += mu * mu;
        s_res2 += resid * resid;
    }
    const T tm = wave_sum(fok ? wd * h_rs[lane] : T(0));
    const double total_mu = static_cast<double>(tm);
    // total_mu_sq = sum_ab Wd_a G_H(a,b) Wd_b  (fp64 accumulation as the reference)
    double acc = 0.0;
    for (int b2 = 0; b2 < k; ++b2) {
        const double wb = static_cast<double>(__shfl(wd, b2, 64));
        if (fok) acc += static_cast<double>(wd) * static_cast<double>(G_H[(int64_t)b2 * k + lane]) * wb;
    }
    const double total_mu_sq = wave_sum(acc);
    if (lane == 0) {
        const double total_resid_sq = s_res2 + (total_mu_sq - s_mu2);
        const double excess = total_resid_sq - total_mu;
        if (excess > 1e-10 && total_mu_sq > 1e-10) {
            double r_new = total_mu_sq / excess;
            r_new = r_new < r_max ? r_new : r_max;
            r_new = r_new > r_min ? r_new : r_min;
            if (isfinite(r_new)) nb_size[i] = static_cast<T>(r_new);
        } else {
            nb_size[i] = static_cast<T>(r_max);
        }
    }
}

// NB negative log-likelihood over the NONZEROS of A (explicit_loss.hpp:53-77), per-row theta, fp64 partials.
template <class T>
__global__ __launch_bounds__(256) void nb_loss_kernel(
    const int* __restrict__ colptr, const int* __restrict__ rowidx, const T* __restrict__ vals, int64_t ncols,
    const T* __restrict__ W_T, const T* __restrict__ d, const T* __restrict__ H, const T* __restrict__ theta_row, int k,
    double* __restrict__ partial) {
    __shared__ double sh[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * 4 + wave;
    double acc = 0.0;
    if (j < ncols) {
        const bool fok = lane < k;
        const T hv = fok ? H[j * (int64_t)k + lane] : T(0);
        for (int t = colptr[j]; t < colptr[j + 1]; ++t) {
            const int row = rowidx[t];
            const T wd = fok ? W_T[(int64_t)row * k + lane] * d[lane] : T(0);
            const T pred = wave_sum(wd * hv);
            const double y = static_cast<double>(vals[t]);
            double mu = static_cast<double>(pred);
            mu = mu > 1e-10 ? mu : 1e-10;
            double r = static_cast<double>(theta_row ? theta_row[row] : T(0));
            r = r > 1e-10 ? r : 1e-10;
            const double nll = -lgamma(y + r) + lgamma(r) - r * log(r / (r + mu)) - y * log(mu / (r + mu));
            acc += static_cast<double>(static_cast<T>(nll));      // the reference casts each term to Scalar
        }
    }
    if (lane == 0) sh[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

}  // namespace rk
