// ============================================================================
// kernels_cv_irls.hip.h -- cross-validation half-updates and losses for the IRLS losses (GP, NB, Gamma, inverse Gaussian,
// Tweedie; robust MSE): reference nmf/cv_detail.hpp:101-292 (irls_solve_col_cv / irls_solve_row_cv), nmf/fit_cv.hpp:446-456,
// :670-689 (callers), :866-961 (GP theta over the training entries), :1377-1443 (per-element losses).
//
// What differs from the non-CV IRLS kernels (kernels_irls.hip.h) -- all of it the reference's behaviour, restated:
//   * the weighted Gram is built FROM ZERO over the column's TRAINING entries, G_w = sum_train w f f^T (no base Gram, no w - 1),
//     plus the additive CV features (L2, graph, L21: `G_add`) and 1e-15 on the diagonal;
//   * training entries: the nonzeros that are not held out (mask_zeros) or EVERY row that is not held out, zeros included --
//     then a column costs O(rows k^2) per pass, as it does in the reference;
//   * the weight is compute_irls_weight(residual, predicted, loss) with the default observed = 0 and theta = 0: dispersion
//     estimates never reach the CV weights, and GP takes irls_weight_gp (math/loss.hpp:197-229), not the KL weight;
//   * the solve starts from the current column and b_w is NOT residual-corrected for it (CD: L1 inside, cd_maxit sweeps, no
//     tolerance; or Cholesky + clip);
//   * the loss is compute_loss(value, prediction, loss, theta) per element, summed separately over training and held-out
//     entries; theta = GP's per-row theta, 0 for every other loss.
// One wavefront per column, lane r = feature r (k <= 64), the per-column Gram in LDS: the layout of cv_solve_kernel.
// Correct first: the rank-1 accumulation runs on the vector ALU (k LDS read-modify-writes per training entry).
// ============================================================================
#pragma once
#include "kernels.hip.h"
#include "kernels_irls.hip.h"

namespace rk {

// math/loss.hpp:197-229  irls_weight_gp(observed, predicted, theta, blend), fp64 inside, eps = tiny_num<Scalar>() = Scalar(1e-15)
template <class T> __device__ __forceinline__ T irls_weight_gp_dev(T observed, T predicted, T theta, T blend) {
    const double eps = static_cast<double>(static_cast<T>(1e-15));
    double s = static_cast<double>(predicted);
    s = s > eps ? s : eps;
    const double y = static_cast<double>(observed), th = static_cast<double>(theta), bl = static_cast<double>(blend);
    const double eff_blend = bl * (s < 1.0 ? s : 1.0);
    double w_gp = 1.0 / (s * s);
    if (y >= 1.0) {
        double denom = s + th * y;
        denom = denom > eps ? denom : eps;
        w_gp += (y - 1.0) / (denom * denom);
    }
    if (eff_blend < 0.999) {
        const double log_w_kl = -log(s);
        const double log_w_gp = log(w_gp > 1e-300 ? w_gp : 1e-300);
        double weight = exp((1.0 - eff_blend) * log_w_kl + eff_blend * log_w_gp);
        weight = weight < 1e6 ? weight : 1e6;
        return static_cast<T>(weight);
    }
    w_gp = w_gp < 1e6 ? w_gp : 1e6;
    return static_cast<T>(w_gp);
}

// the weight as the CV solves evaluate it (observed = 0, theta = 0; cv_detail.hpp:154, :255)
template <class T> __device__ __forceinline__ T cv_irls_weight_dev(int loss_type, T residual, T predicted, T power, T robust) {
    T w_dist;
    if (loss_type == 0) w_dist = T(1);
    else if (loss_type == 4) w_dist = irls_weight_gp_dev<T>(T(0), predicted, T(0), T(1));
    else w_dist = irls_weight_dev<T>(loss_type, predicted, T(0), power);
    if (robust > T(0)) {
        const T wd = w_dist > T(1e-15) ? w_dist : T(1e-15);
        const T abs_r = tabs(residual * sqrt(wd));
        return abs_r <= robust ? w_dist : w_dist * (robust / (abs_r + T(1e-15)));
    }
    return w_dist;
}

// compute_loss(observed, predicted, loss, theta), math/loss.hpp:511-535 with the per-distribution terms of :382-505, each in fp64
// and cast to Scalar as the reference's Scalar-typed functions return them (loss_type 0: squared error)
template <class T> __device__ __forceinline__ T cv_loss_term_dev(int loss_type, T observed, T predicted, T theta, double power) {
    if (loss_type == 0) { const T df = observed - predicted; return df * df; }
    const double y = static_cast<double>(observed);
    double mu = static_cast<double>(predicted);
    mu = mu > 1e-10 ? mu : 1e-10;
    const double th = static_cast<double>(theta);
    double nll;
    if (loss_type == 4) {                  // loss_contribution_gp
        const double opt = 1.0 + th;
        nll = -log(mu / opt);
        if (y >= 1.0) {
            double inner = (mu + th * y) / opt;
            inner = inner > 1e-10 ? inner : 1e-10;
            nll -= (y - 1.0) * log(inner);
        }
        nll += (mu + th * y) / opt;
    } else if (loss_type >= 6) {           // Gamma / inverse Gaussian / Tweedie deviance terms
        const double yy = y > 1e-10 ? y : 1e-10;
        const double pp = loss_type == 6 ? 2.0 : (loss_type == 7 ? 3.0 : power);
        if (loss_type == 7) {
            const double df = yy - mu;
            nll = df * df / (mu * mu * yy);
        } else if (fabs(pp - 1.0) < 1e-6) {
            nll = 2.0 * (yy * log(yy / mu) - (yy - mu));
        } else if (fabs(pp - 2.0) < 1e-6) {
            nll = 2.0 * (-log(yy / mu) + (yy - mu) / mu);
        } else {
            const double omp = 1.0 - pp, tmp = 2.0 - pp;
            nll = 2.0 * (pow(yy, tmp) / (omp * tmp) - yy * pow(mu, omp) / omp + pow(mu, tmp) / tmp);
        }
    } else {                               // loss_contribution_nb
        const double r = th > 1e-10 ? th : 1e-10;
        nll = -lgamma(y + r) + lgamma(r) - r * log(r / (r + mu)) - y * log(mu / (r + mu));
    }
    return static_cast<T>(nll);
}

template <class T, int KP>   // KP in {32, 64}
__global__ __launch_bounds__(256) void cv_irls_solve_kernel(
    const int* __restrict__ colptr, const int* __restrict__ rowidx, const T* __restrict__ vals, int64_t ncols, int nrows,
    const T* __restrict__ F, const T* __restrict__ Gadd, T* __restrict__ X, int k, unsigned long long seed,
    unsigned long long threshold, int mask_zeros, int transposed, T l1, int nonneg, int maxit, int solver_mode, int loss_type,
    int irls_max_iter, T irls_tol, T power, T robust, const int* __restrict__ mp = nullptr, const int* __restrict__ mi = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    T* Gl = reinterpret_cast<T*>(smem_raw) + (size_t)wave * KP * KP;   // [c][r]
    const int64_t j = (int64_t)blockIdx.x * 4 + wave;
    if (j >= ncols) return;
    const bool fok = lane < k;
    const bool lin = lane < KP;
    const int ll = lin ? lane : 0;
    const unsigned col = (unsigned)j;
    const int ts = colptr[j], te = colptr[j + 1];
    T x = fok ? X[j * (int64_t)k + lane] : T(0);
    auto held_row = [&](int row) {          // excluded rows: held out, or user-masked (fit_cv.hpp:491-501 -> cv_detail.hpp:116-117)
        if (cv_user_masked(mp, mi, j, row)) return true;
        return (transposed ? cv_hash_dev(seed, col, (unsigned)row) : cv_hash_dev(seed, (unsigned)row, col)) < threshold;
    };
    for (int it = 0; it < irls_max_iter; ++it) {
        for (int c = 0; c < KP; ++c)
            if (lin) Gl[c * KP + lane] = T(0);
        T b = T(0);
        // one training entry: weight from the current iterate, b_w += (w a) f, G_w += w f f^T (upper triangle formed as
        // (w f_a) f_b and mirrored, cv_detail.hpp:158-166)
        auto entry = [&](int row, T a) {
            const T fr = fok ? F[(int64_t)row * k + lane] : T(0);
            const T pred = wave_sum(fr * x);
            const T w = cv_irls_weight_dev<T>(loss_type, a - pred, pred, power, robust);
            b += fr * (w * a);
            const T wfr = w * fr;
            for (int c = 0; c < k; ++c) {
                const T fc = __shfl(fr, c, 64), wfc = __shfl(wfr, c, 64);
                if (lin) Gl[c * KP + lane] += lane <= c ? wfr * fc : wfc * fr;
            }
        };
        if (mask_zeros) {
            for (int t = ts; t < te; ++t) {
                const int row = rowidx[t];
                if (!held_row(row)) entry(row, vals[t]);
            }
        } else {
            int t = ts;
            for (int r0 = 0; r0 < nrows; r0 += 64) {
                const int r = r0 + lane;
                unsigned long long m = __ballot(r < nrows && !held_row(r));
                while (m) {
                    const int bit = __builtin_ctzll(m);
                    m &= m - 1;
                    const int row = r0 + bit;
                    while (t < te && rowidx[t] < row) ++t;
                    const T a = (t < te && rowidx[t] == row) ? vals[t] : T(0);
                    entry(row, a);
                }
            }
        }
        for (int c = 0; c < KP; ++c) {
            if (!lin) continue;
            T v = Gl[c * KP + lane];
            if (fok && c < k) { v += Gadd ? Gadd[(int64_t)c * k + lane] : T(0); if (c == lane) v += T(1e-15); }
            else if (c == lane) v = T(1);          // identity padding beyond k
            Gl[c * KP + lane] = v;
        }
        RK_WAVE_SYNC();
        const T xo = x;
        if (solver_mode == 1) {
            if (l1 > T(0) && fok) b -= l1;
            for (int c = 0; c < KP; ++c) {
                T s = Gl[c * KP + ll];
                for (int p = 0; p < c; ++p) s -= Gl[p * KP + ll] * Gl[p * KP + c];
                T dcc = __shfl(s, c, 64);
                if (!(dcc > T(0))) dcc = tabs(dcc) + T(1e-30);
                const T lcc = sqrt(dcc);
                if (lin) Gl[c * KP + lane] = lane == c ? lcc : (lane > c ? s / lcc : T(0));
                RK_WAVE_SYNC();
            }
            T y = fok ? b : T(0);
            for (int i = 0; i < k; ++i) {
                const T yi = __shfl(y, i, 64) / Gl[i * KP + i];
                if (lane == i) y = yi;
                else if (lane > i) y -= Gl[i * KP + ll] * yi;
            }
            for (int i = k - 1; i >= 0; --i) {
                const T xi = __shfl(y, i, 64) / Gl[i * KP + i];
                if (lane == i) y = xi;
                else if (lane < i) y -= Gl[ll * KP + i] * xi;
            }
            x = y;
            if (nonneg) x = x > T(0) ? x : T(0);
            if (!fok) x = T(0);
        } else {
            const T gd = Gl[ll * KP + ll];
            // static coordinate sweeps (cd_static_sweeps, kernels.hip.h): the lane's Gram column read from the wave's LDS tile at
            // compile-time offsets
            cd_static_sweeps<T, KP>(b, x, gd, fok, l1, nonneg, maxit, [&](auto IC) { return Gl[decltype(IC)::value * KP + ll]; });
        }
        RK_WAVE_SYNC();
        const T rel = fok ? tabs(x - xo) / (tabs(xo) + T(1e-12)) : T(0);
        if (wave_max(rel) < irls_tol) break;
    }
    if (fok) X[j * (int64_t)k + lane] = x;
}

// Per-element losses, one wavefront per column j of A, one lane per entry (mask_zeros: the column's nonzeros; otherwise every
// row, the value found by bisection in the column's sorted rows).  partial[block] = {train sum, test sum}, counts likewise.
template <class T>
__global__ __launch_bounds__(256) void cv_irls_loss_kernel(
    const int* __restrict__ colptr, const int* __restrict__ rowidx, const T* __restrict__ vals, int64_t ncols, int nrows,
    const T* __restrict__ W_T, const T* __restrict__ d, const T* __restrict__ H, const T* __restrict__ theta_row, int k,
    unsigned long long seed, unsigned long long threshold, int mask_zeros, int loss_type, double power,
    double* __restrict__ psum /*2 per block*/, unsigned long long* __restrict__ pcnt /*2 per block*/,
    const int* __restrict__ mp = nullptr, const int* __restrict__ mi = nullptr) {
    __shared__ double sh[4][2];
    __shared__ unsigned long long shn[4][2];
    __shared__ T hs[4][128];
    __shared__ T dsh[128];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * 4 + wave;
    if (threadIdx.x < 128) dsh[threadIdx.x] = (int)threadIdx.x < k ? d[threadIdx.x] : T(0);
    hs[wave][lane] = (j < ncols && lane < k) ? H[j * (int64_t)k + lane] : T(0);
    hs[wave][lane + 64] = (j < ncols && lane + 64 < k) ? H[j * (int64_t)k + lane + 64] : T(0);
    __syncthreads();
    double tr = 0.0, te = 0.0;
    unsigned long long ntr = 0, nte = 0;
    if (j < ncols) {
        const int as = colptr[j], ae = colptr[j + 1];
        auto term = [&](int row, T a) {
            if (cv_user_masked(mp, mi, j, row)) return;              // fit_cv.hpp:1391, :1407: user-masked entries are in neither sum
            const T* wr = W_T + (int64_t)row * k;
            T pred = T(0);
            for (int c = 0; c < k; ++c) pred = tfma(wr[c] * dsh[c], hs[wave][c], pred);
            const T th = (loss_type == 4 && theta_row) ? theta_row[row] : T(0);
            const double lv = static_cast<double>(cv_loss_term_dev<T>(loss_type, a, pred, th, power));
            if (cv_hash_dev(seed, (unsigned)row, (unsigned)j) < threshold) { te += lv; ++nte; } else { tr += lv; ++ntr; }
        };
        if (mask_zeros) {
            for (int t = as + lane; t < ae; t += 64) term(rowidx[t], vals[t]);
        } else {
            for (int row = lane; row < nrows; row += 64) {
                int lo = as, hi = ae;               // first position with rowidx >= row
                while (lo < hi) { const int mid = lo + ((hi - lo) >> 1); if (rowidx[mid] < row) lo = mid + 1; else hi = mid; }
                const T a = (lo < ae && rowidx[lo] == row) ? vals[lo] : T(0);
                term(row, a);
            }
        }
    }
    tr = wave_sum(tr); te = wave_sum(te);
    for (int off = 32; off > 0; off >>= 1) { ntr += __shfl_xor(ntr, off, 64); nte += __shfl_xor(nte, off, 64); }
    if (lane == 0) { sh[wave][0] = tr; sh[wave][1] = te; shn[wave][0] = ntr; shn[wave][1] = nte; }
    __syncthreads();
    if (threadIdx.x == 0) {
        psum[2 * blockIdx.x] = sh[0][0] + sh[1][0] + sh[2][0] + sh[3][0];
        psum[2 * blockIdx.x + 1] = sh[0][1] + sh[1][1] + sh[2][1] + sh[3][1];
        pcnt[2 * blockIdx.x] = shn[0][0] + shn[1][0] + shn[2][0] + shn[3][0];
        pcnt[2 * blockIdx.x + 1] = shn[0][1] + shn[1][1] + shn[2][1] + shn[3][1];
    }
}
// out4 = {train sum, n_train, test sum, n_test}
static __global__ __launch_bounds__(256) void cv_irls_loss_final_kernel(const double* __restrict__ psum, const unsigned long long* __restrict__ pcnt,
                                                                  int nblk, double* __restrict__ out4) {
    __shared__ double ss[256][2];
    __shared__ unsigned long long sn[256][2];
    double a0 = 0.0, a1 = 0.0;
    unsigned long long c0 = 0, c1 = 0;
    for (int i = threadIdx.x; i < nblk; i += 256) { a0 += psum[2 * i]; a1 += psum[2 * i + 1]; c0 += pcnt[2 * i]; c1 += pcnt[2 * i + 1]; }
    ss[threadIdx.x][0] = a0; ss[threadIdx.x][1] = a1; sn[threadIdx.x][0] = c0; sn[threadIdx.x][1] = c1;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            ss[threadIdx.x][0] += ss[threadIdx.x + off][0]; ss[threadIdx.x][1] += ss[threadIdx.x + off][1];
            sn[threadIdx.x][0] += sn[threadIdx.x + off][0]; sn[threadIdx.x][1] += sn[threadIdx.x + off][1];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out4[0] = ss[0][0]; out4[1] = static_cast<double>(sn[0][0]); out4[2] = ss[0][1]; out4[3] = static_cast<double>(sn[0][1]); }
}

// GP theta by the MM update over the TRAINING entries (nmf/fit_cv.hpp:866-961): dispersion_rows_kernel's GP branch with the held-out
// nonzeros skipped and the held-out pairs' predictions -- zeros included, whatever mask_zeros says (:886-893) -- taken out of
// sum_s.  One wavefront per row i of A (= column i of A^T).
template <class T>
__global__ __launch_bounds__(256) void cv_gp_theta_rows_kernel(
    const int* __restrict__ tp, const int* __restrict__ ti, const T* __restrict__ tx, int64_t m, int64_t n,
    const T* __restrict__ W_T, const T* __restrict__ d, const T* __restrict__ H, const T* __restrict__ h_rs, int k,
    unsigned long long seed, unsigned long long threshold, double hi, T* __restrict__ s_cache, T* __restrict__ theta) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    if (i >= m) return;
    const bool fok = lane < k, fok2 = lane + 64 < k;                     // lane holds features lane and lane + 64 (k <= 128)
    const T wd = fok ? W_T[i * (int64_t)k + lane] * d[lane] : T(0);
    const T wd2 = fok2 ? W_T[i * (int64_t)k + lane + 64] * d[lane + 64] : T(0);
    __shared__ T wds[4][128];
    wds[wave][lane] = wd;
    wds[wave][lane + 64] = wd2;
    __builtin_amdgcn_wave_barrier();
    const int ts = tp[i], te = tp[i + 1];
    double sum_y = 0.0, n_nz = 0.0;
    for (int t = ts + lane; t < te; t += 64) {
        const int col = ti[t];
        if (cv_hash_dev(seed, (unsigned)i, (unsigned)col) < threshold) { s_cache[t] = T(-1); continue; }     // held out: not a training entry
        const T* hr = H + (int64_t)col * k;
        T dot = T(0);
        for (int c = 0; c < k; ++c) dot = tfma(wds[wave][c], hr[c], dot);
        dot = dot > T(0) ? dot : T(0);            // (s_cache < 0 marks held-out entries; predictions are floored at 1e-10 below anyway)
        const double y = static_cast<double>(tx[t]);
        s_cache[t] = dot;
        sum_y += y;
        if (y >= 1.0) n_nz += 1.0;
    }
    sum_y = wave_sum(sum_y);
    n_nz = wave_sum(n_nz);
    double sum_s = static_cast<double>(wave_sum((fok ? wd * h_rs[lane] : T(0)) + (fok2 ? wd2 * h_rs[lane + 64] : T(0))));
    double held_s = 0.0;
    for (int64_t j0 = 0; j0 < n; j0 += 64) {
        const int64_t jj = j0 + lane;
        if (jj < n && cv_hash_dev(seed, (unsigned)i, (unsigned)jj) < threshold) {
            const T* hr = H + jj * k;
            T dot = T(0);
            for (int c = 0; c < k; ++c) dot = tfma(wds[wave][c], hr[c], dot);
            held_s += static_cast<double>(dot);
        }
    }
    sum_s -= wave_sum(held_s);
    T th_s = theta[i];
    for (int mm = 0; mm < 5; ++mm) {
        const double th = static_cast<double>(th_s);
        double alpha = 0.0, gamma = 0.0;
        for (int t = ts + lane; t < te; t += 64) {
            const double y = static_cast<double>(tx[t]);
            const T sc = s_cache[t];
            if (y >= 1.0 && !(sc < T(0))) {
                double sv = static_cast<double>(sc);
                sv = sv > 1e-10 ? sv : 1e-10;
                double denom = sv + th * y;
                denom = denom > 1e-10 ? denom : 1e-10;
                const double eta1 = sv / denom;
                alpha += (y - 1.0) * eta1;
                gamma += (y - 1.0) * (1.0 - eta1);
            }
        }
        alpha = wave_sum(alpha);
        gamma = wave_sum(gamma);
        const double a = alpha + n_nz;
        const double b = (sum_y - sum_s) - gamma + a;
        if (a > 1e-15) {
            const double disc = b * b + 4.0 * a * gamma;
            if (disc > 0.0 && isfinite(disc)) {
                const double nt = (-b + sqrt(disc)) / (2.0 * a);
                if (isfinite(nt) && nt >= 0.0) th_s = static_cast<T>(nt < hi ? nt : hi);
            }
        }
    }
    if (lane == 0) theta[i] = th_s;
}

}  // namespace rk
