// ============================================================================
// kernels_wide.hip.h -- the per-column-Gram solves for 64 < k <= 128: IRLS half-update, cross-validation half-updates (MSE and
// IRLS losses) and the explicit-mask half-update.  Same semantics, statement for statement, as their k <= 64 kernels
// (irls_nb_solve_kernel, cv_solve_kernel, cv_irls_solve_kernel, masked_solve_kernel); what changes is the layout:
//
//   one wavefront per column (64-thread workgroups), lane l holds features l and l + 64 ("slots" 0 and 1), the column's
//   k x k system G_loc in LDS as a 128 x 128 tile [c][r] (64 KiB fp32, 128 KiB fp64: one or two workgroups per CU).
//
// Entries of a column are staged in chunks of WCH rows of F (LDS), so a sweep over the Gram tile applies WCH rank-1 terms per
// read-modify-write of an element -- the tile is LDS-resident, not register-resident as for k <= 64 (a lane would need 256
// registers for its two rows).  The coordinate-descent and Cholesky solves walk the coordinates slot by slot (features 0..63,
// then 64..127): the same sequential order as the reference's loops.
//
// Reference routines: primitives/cpu/nnls_batch_irls.hpp:202-329,465-520; nmf/fit_cv.hpp:420-478,591-830 + nmf/cv_detail.hpp:
// 66-85,101-292,304-405; nmf/masked_nnls.hpp; primitives/cpu/nnls_batch.hpp:70-132 (cd_nnls_col_fixed); cholesky_clip.
// ============================================================================
#pragma once
#include "kernels.hip.h"
#include "kernels_irls.hip.h"
#include "kernels_cv_irls.hip.h"

namespace rk {

constexpr int WKP = 128;      // padded rank of the tile
constexpr int WCH = 8;        // staged entries per Gram sweep

// LDS of one workgroup: Gram tile | WCH staged rows of F | two coefficients per staged row
template <class T> constexpr size_t wide_smem_bytes() { return (size_t)(WKP * WKP + WCH * WKP + 2 * WCH) * sizeof(T); }

template <class T> struct WideCol {
    T* Gl;        // [c][r]
    T* fs;        // [e][feature]
    T* cf;        // [2][e]
    int k, lane;
    bool ok[2];
    __device__ WideCol(char* smem, int k_) : k(k_), lane(threadIdx.x & 63) {
        Gl = reinterpret_cast<T*>(smem);
        fs = Gl + WKP * WKP;
        cf = fs + WCH * WKP;
        ok[0] = lane < k; ok[1] = lane + 64 < k;
    }
    // feature c of a two-slot lane vector (c uniform)
    __device__ __forceinline__ T feat(const T (&v)[2], int c) const { return __shfl(c < 64 ? v[0] : v[1], c & 63, 64); }
    __device__ __forceinline__ T dot(const T (&a)[2], const T (&b)[2]) const { return wave_sum(a[0] * b[0] + a[1] * b[1]); }
    __device__ __forceinline__ void load_row(const T* __restrict__ F, int row, T (&fr)[2]) const {
        fr[0] = ok[0] ? F[(int64_t)row * k + lane] : T(0);
        fr[1] = ok[1] ? F[(int64_t)row * k + lane + 64] : T(0);
    }
    // tile <- base (k x k, column-major; nullptr = zero) with identity padding; diag_add on the leading k diagonal entries
    __device__ void set_base(const T* __restrict__ base, T diag_add) {
        for (int c = 0; c < WKP; ++c)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int r = lane + 64 * s;
                T v;
                if (r < k && c < k) { v = base ? base[(int64_t)c * k + r] : T(0); if (r == c) v += diag_add; }
                else v = r == c ? T(1) : T(0);
                Gl[c * WKP + r] = v;
            }
    }
    // G(r, c) += add(r, c) (k x k, column-major; nullptr = nothing), then diag on the leading k diagonal entries
    __device__ void add_base(const T* __restrict__ add, T diag) {
        for (int c = 0; c < k; ++c)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int r = lane + 64 * s;
                if (r >= k) continue;
                T v = Gl[c * WKP + r];
                if (add) v += add[(int64_t)c * k + r];
                if (r == c) v += diag;
                Gl[c * WKP + r] = v;
            }
    }
    __device__ __forceinline__ void stage(int e, const T (&fr)[2], T c0, T c1) {
        fs[e * WKP + lane] = fr[0];
        fs[e * WKP + lane + 64] = fr[1];
        if (lane == 0) { cf[e] = c0; cf[WCH + e] = c1; }
    }
    // G(r, c) += sum_e cf[e] * f_e[r] * f_e[c] over the ne staged rows; MIRROR: the term is formed as (cf f_min) f_max on both
    // sides of the diagonal (cv_detail.hpp:158-166), otherwise as (f_r cf) f_c (nnls_batch_irls.hpp: W_nnz_scaled.col * f^T)
    template <bool MIRROR> __device__ void flush(int ne) {
        RK_WAVE_SYNC();
        for (int c = 0; c < k; ++c) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int r = lane + 64 * s;
                if (r >= k) continue;
                T g = Gl[c * WKP + r];
                for (int e = 0; e < ne; ++e) {
                    const T fr = fs[e * WKP + r], fc = fs[e * WKP + c], w = cf[e];
                    if (MIRROR) g += r <= c ? (w * fr) * fc : (w * fc) * fr;
                    else g = tfma(fr * w, fc, g);
                }
                Gl[c * WKP + r] = g;
            }
        }
        RK_WAVE_SYNC();
    }
    // cd_nnls_col_fixed on the tile: sequential sweeps with ballot skipping of coordinates that do not move.  l1_in: subtracted
    // from every quotient (the CV / IRLS callers); tol > 0: the masked solver's relative-change stop; otherwise a sweep without
    // any effective step ends the solve (all later sweeps are no-ops too)
    __device__ void cd(T (&b)[2], T (&x)[2], T l1_in, int nonneg, int maxit, T tol) {
        T gd[2], ginv[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) { gd[s] = Gl[(lane + 64 * s) * WKP + lane + 64 * s]; ginv[s] = gd[s] > T(0) ? T(1) / gd[s] : T(0); }
        const bool check = tol > T(0);
        const T inv_k = T(1) / static_cast<T>(k);
        if (!check) {
            // no relative-change stop (the CV / IRLS callers): static coordinate sweeps, as cd_static_sweeps (kernels.hip.h) --
            // wave-uniform control flow, fma -> max -> readlane -> two fmas per coordinate, the tile's column read at
            // compile-time LDS offsets, the steps collected per lane and the iterate moved once per sweep; the step multiplies by 1/G_ii
            const T pinf = static_cast<T>(__builtin_inff());
            const T inf_rt = maxit >= 0 ? pinf : T(0);
            T gi[2], nl1[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bool alive = ok[s] && gd[s] > T(0);
                gi[s] = alive ? ginv[s] : T(0);
                nl1[s] = alive ? -l1_in : T(0);
            }
            for (int it = 0; it < maxit; ++it) {
                T xe[2], aown[2] = {T(0), T(0)};       // a coordinate is visited once per sweep: the steps are collected, x moves after the sweep
#pragma unroll
                for (int s = 0; s < 2; ++s) xe[s] = nonneg ? x[s] : pinf;
                cd_static_for<0, 2>([&](auto SC) {
                    constexpr int s = decltype(SC)::value;
                    cd_static_for<0, 64>([&](auto IC) {
                        constexpr int i = decltype(IC)::value;
                        constexpr int ci = i + 64 * s;
                        if (ci < k) {          // wave-uniform
                            const T diff = tfma(b[s], gi[s], nl1[s]);
                            const T ad = cd_static_max(diff, -xe[s], inf_rt);
                            const T ad_i = lane_value(ad, i);
                            aown[s] = cd_write_lane<i>(aown[s], ad_i);
                            b[0] = tfma(-Gl[ci * WKP + lane], ad_i, b[0]);
                            b[1] = tfma(-Gl[ci * WKP + lane + 64], ad_i, b[1]);
                        }
                    });
                });
                const T x0n = x[0] + aown[0], x1n = x[1] + aown[1];
                const bool moved = x0n != x[0] || x1n != x[1];
                x[0] = x0n; x[1] = x1n;
                if (!__any(moved)) break;   // no step, or the iterate is at its floating-point fixed point
            }
            return;
        }
        for (int it = 0; it < maxit; ++it) {
            T tol_sum = T(0);
            bool any = false;
            const T xs0 = x[0], xs1 = x[1];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (64 * s >= k) break;
                int cur = 0;
                while (true) {
                    T diff = sweep_quotient(b[s], gd[s], ginv[s]);
                    if (l1_in != T(0)) diff -= l1_in;
                    const T nv = x[s] + diff;
                    T ad = diff, nx = nv;
                    if (nonneg && nv < T(0)) { ad = -x[s]; nx = T(0); }
                    const bool moves = ok[s] && (gd[s] > T(0)) && (ad != T(0)) && (lane >= cur);
                    const unsigned long long mask = __ballot(moves);
                    if (mask == 0ull) break;
                    any = true;
                    const int i = __builtin_ctzll(mask);
                    const T ad_i = lane_value(ad, i), nx_i = lane_value(nx, i);
                    if (lane == i) x[s] = nx_i;
                    if (check) tol_sum += tabs(ad_i) / (tabs(nx_i) + T(1e-15));
                    const int ci = i + 64 * s;
                    b[0] = tfma(-Gl[ci * WKP + lane], ad_i, b[0]);
                    b[1] = tfma(-Gl[ci * WKP + lane + 64], ad_i, b[1]);
                    cur = i + 1;
                    if (cur >= 64) break;
                }
            }
            if (check) { if (tol_sum * inv_k < tol) break; }
            else if (!any || !__any(x[0] != xs0 || x[1] != xs1)) break;   // no step, or the iterate is at its floating-point fixed point
        }
    }
    // in-LDS Cholesky (left-looking) of the leading k x k block, forward / back substitution, clip: x = max(G^-1 b, 0)
    __device__ void chol(const T (&b)[2], T (&x)[2], int nonneg) {
        for (int c = 0; c < k; ++c) {
            T sv[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int r = lane + 64 * s;
                T acc = Gl[c * WKP + r];
                for (int p = 0; p < c; ++p) acc -= Gl[p * WKP + r] * Gl[p * WKP + c];
                sv[s] = acc;
            }
            T dcc = feat(sv, c);
            if (!(dcc > T(0))) dcc = tabs(dcc) + T(1e-30);
            const T lcc = sqrt(dcc);
            RK_WAVE_SYNC();
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int r = lane + 64 * s;
                Gl[c * WKP + r] = r == c ? lcc : (r > c ? sv[s] / lcc : T(0));
            }
            RK_WAVE_SYNC();
        }
        T y[2] = {ok[0] ? b[0] : T(0), ok[1] ? b[1] : T(0)};
        for (int i = 0; i < k; ++i) {
            const T yi = feat(y, i) / Gl[i * WKP + i];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int r = lane + 64 * s;
                if (r == i) y[s] = yi;
                else if (r > i) y[s] -= Gl[i * WKP + r] * yi;
            }
        }
        for (int i = k - 1; i >= 0; --i) {
            const T xi = feat(y, i) / Gl[i * WKP + i];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int r = lane + 64 * s;
                if (r == i) y[s] = xi;
                else if (r < i) y[s] -= Gl[r * WKP + i] * xi;
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            x[s] = y[s];
            if (nonneg) x[s] = x[s] > T(0) ? x[s] : T(0);
            if (!ok[s]) x[s] = T(0);
        }
    }
    __device__ __forceinline__ void load_x(const T* __restrict__ X, int64_t j, T (&x)[2]) const {
        x[0] = ok[0] ? X[j * (int64_t)k + lane] : T(0);
        x[1] = ok[1] ? X[j * (int64_t)k + lane + 64] : T(0);
    }
    __device__ __forceinline__ void store_x(T* __restrict__ X, int64_t j, const T (&x)[2]) const {
        if (ok[0]) X[j * (int64_t)k + lane] = x[0];
        if (ok[1]) X[j * (int64_t)k + lane + 64] = x[1];
    }
    __device__ __forceinline__ T rel_change(const T (&x)[2], const T (&xo)[2]) const {
        const T r0 = ok[0] ? tabs(x[0] - xo[0]) / (tabs(xo[0]) + T(1e-12)) : T(0);
        const T r1 = ok[1] ? tabs(x[1] - xo[1]) / (tabs(xo[1]) + T(1e-12)) : T(0);
        return wave_max(r0 > r1 ? r0 : r1);
    }
};

// ---------------------------------------------------------------------------
// IRLS half-update (irls_nb_solve_kernel for k > 64): G_w = G + sum_nz (w - 1) f f^T (+ l2 I), b_w = sum_nz (w a) f,
// residual start b_w - G_w x_old, all cd_maxit sweeps, up to irls_max_iter passes from x = 0.
// ---------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(64) void wide_irls_solve_kernel(
    const int* __restrict__ colptr, const int* __restrict__ rowidx, const T* __restrict__ vals, int64_t ncols,
    const T* __restrict__ F, const T* __restrict__ Gbase, T* __restrict__ X, int k, T l1, T l2, int nonneg, int cd_maxit,
    int irls_max_iter, T irls_tol, const T* __restrict__ theta_row, const T* __restrict__ theta_col, int loss_type, T power,
    T robust, unsigned long long* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int64_t j = blockIdx.x;
    if (j >= ncols) return;
    WideCol<T> wc(smem_raw, k);
    const int as = colptr[j], ae = colptr[j + 1];
    const T th_col = theta_col ? theta_col[j] : T(0);
    T x[2] = {T(0), T(0)};
    int passes = 0;
    for (int irls = 0; irls < irls_max_iter; ++irls) {
        ++passes;
        wc.set_base(Gbase, T(0));
        T bw[2] = {T(0), T(0)};
        for (int t0 = as; t0 < ae; t0 += WCH) {
            const int ne = ae - t0 < WCH ? ae - t0 : WCH;
            for (int e = 0; e < ne; ++e) {
                const int row = rowidx[t0 + e];
                const T a = vals[t0 + e];
                T fr[2];
                wc.load_row(F, row, fr);
                const T recon = wc.dot(fr, x);
                const T th = theta_col ? th_col : (theta_row ? theta_row[row] : T(0));
                const T w = irls_weight_full_dev<T>(loss_type, a - recon, recon, th, power, robust);
                const T wv = w * a;
                bw[0] = tfma(fr[0], wv, bw[0]);
                bw[1] = tfma(fr[1], wv, bw[1]);
                wc.stage(e, fr, w - T(1), T(0));
            }
            wc.template flush<false>(ne);
        }
        if (l2 > T(0)) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
                if (wc.ok[s]) wc.Gl[(wc.lane + 64 * s) * WKP + wc.lane + 64 * s] += l2;
        }
        RK_WAVE_SYNC();
        const T xo[2] = {x[0], x[1]};
        T b[2] = {bw[0], bw[1]};
        for (int c = 0; c < k; ++c) {
            const T xc = wc.feat(xo, c);
            b[0] = tfma(-wc.Gl[c * WKP + wc.lane], xc, b[0]);
            b[1] = tfma(-wc.Gl[c * WKP + wc.lane + 64], xc, b[1]);
        }
        wc.cd(b, x, l1, nonneg, cd_maxit, T(0));
        RK_WAVE_SYNC();
        if (wc.rel_change(x, xo) < irls_tol) break;
    }
    wc.store_x(X, j, x);
    if (stats && wc.lane == 0) { atomicAdd(stats, (unsigned long long)passes); atomicAdd(stats + 1, (unsigned long long)passes * (unsigned long long)(ae - as)); }
}

// ---------------------------------------------------------------------------
// MSE cross-validation half-update (cv_solve_kernel for k > 64): G_local = G - sum over held-out rows f f^T, b over the
// training nonzeros, solve from the current column without a warm-start correction.
// ---------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(64) void wide_cv_solve_kernel(
    const int* __restrict__ colptr, const int* __restrict__ rowidx, const T* __restrict__ vals, int64_t ncols, int nrows,
    const T* __restrict__ F, const T* __restrict__ Gfull, T* __restrict__ X, int k, unsigned long long seed,
    unsigned long long threshold, int mask_zeros, int transposed, T l1, int nonneg, int maxit, int solver_mode,
    const int* __restrict__ mp = nullptr, const int* __restrict__ mi = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int64_t j = blockIdx.x;
    if (j >= ncols) return;
    WideCol<T> wc(smem_raw, k);
    const int lane = wc.lane;
    const unsigned col = (unsigned)j;
    auto held_row = [&](int row) {
        return (transposed ? cv_hash_dev(seed, col, (unsigned)row) : cv_hash_dev(seed, (unsigned)row, col)) < threshold;
    };
    wc.set_base(Gfull, T(0));
    T b[2] = {T(0), T(0)};
    int ne = 0;
    auto held_entry = [&](int row) {
        T fr[2];
        wc.load_row(F, row, fr);
        wc.stage(ne, fr, T(-1), T(0));
        if (++ne == WCH) { wc.template flush<false>(ne); ne = 0; }
    };
    for (int t = colptr[j]; t < colptr[j + 1]; ++t) {
        const int row = rowidx[t];
        if (!held_row(row)) {
            if (cv_user_masked(mp, mi, j, row)) continue;
            T fr[2];
            wc.load_row(F, row, fr);
            b[0] = tfma(vals[t], fr[0], b[0]);
            b[1] = tfma(vals[t], fr[1], b[1]);
        } else if (mask_zeros) {
            held_entry(row);
        }
    }
    if (!mask_zeros) {
        for (int r0 = 0; r0 < nrows; r0 += 64) {
            const int r = r0 + lane;
            unsigned long long m = __ballot(r < nrows && held_row(r));
            while (m) {
                const int bit = __builtin_ctzll(m);
                m &= m - 1;
                held_entry(r0 + bit);
            }
        }
    }
    if (mp) {               // user-masked rows that are not test rows (cv_detail.hpp:433-468): out of the Gram as well
        const int ts = colptr[j], te = colptr[j + 1];
        for (int t = mp[j]; t < mp[j + 1]; ++t) {
            const int row = mi[t];
            if (held_row(row) && (!mask_zeros || cv_row_stored(rowidx, ts, te, row))) continue;
            held_entry(row);
        }
    }
    if (ne) wc.template flush<false>(ne);
    RK_WAVE_SYNC();
    T x[2];
    wc.load_x(X, j, x);
    if (solver_mode == 1) {
        if (l1 > T(0)) { if (wc.ok[0]) b[0] -= l1; if (wc.ok[1]) b[1] -= l1; }
        wc.chol(b, x, nonneg);
    } else {
        wc.cd(b, x, l1, nonneg, maxit, T(0));
    }
    wc.store_x(X, j, x);
}

// ---------------------------------------------------------------------------
// Cross-validation half-update with an IRLS loss (cv_irls_solve_kernel for k > 64): G_w = sum over TRAIN entries w f f^T + G_add
// + 1e-15 I from zero every pass, b_w = sum (w a) f, no residual correction.
// ---------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(64) void wide_cv_irls_solve_kernel(
    const int* __restrict__ colptr, const int* __restrict__ rowidx, const T* __restrict__ vals, int64_t ncols, int nrows,
    const T* __restrict__ F, const T* __restrict__ Gadd, T* __restrict__ X, int k, unsigned long long seed,
    unsigned long long threshold, int mask_zeros, int transposed, T l1, int nonneg, int maxit, int solver_mode, int loss_type,
    int irls_max_iter, T irls_tol, T power, T robust, const int* __restrict__ mp = nullptr, const int* __restrict__ mi = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int64_t j = blockIdx.x;
    if (j >= ncols) return;
    WideCol<T> wc(smem_raw, k);
    const int lane = wc.lane;
    const unsigned col = (unsigned)j;
    const int ts = colptr[j], te = colptr[j + 1];
    auto held_row = [&](int row) {          // excluded rows: held out, or user-masked (fit_cv.hpp:491-501 -> cv_detail.hpp:116-117)
        if (cv_user_masked(mp, mi, j, row)) return true;
        return (transposed ? cv_hash_dev(seed, col, (unsigned)row) : cv_hash_dev(seed, (unsigned)row, col)) < threshold;
    };
    T x[2];
    wc.load_x(X, j, x);
    for (int it = 0; it < irls_max_iter; ++it) {
        wc.set_base(nullptr, T(0));
        T b[2] = {T(0), T(0)};
        int ne = 0;
        auto entry = [&](int row, T a) {
            T fr[2];
            wc.load_row(F, row, fr);
            const T pred = wc.dot(fr, x);
            const T w = cv_irls_weight_dev<T>(loss_type, a - pred, pred, power, robust);
            b[0] += fr[0] * (w * a);
            b[1] += fr[1] * (w * a);
            wc.stage(ne, fr, w, T(0));
            if (++ne == WCH) { wc.template flush<true>(ne); ne = 0; }
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
        if (ne) wc.template flush<true>(ne);
        wc.add_base(Gadd, T(1e-15));
        RK_WAVE_SYNC();
        const T xo[2] = {x[0], x[1]};
        if (solver_mode == 1) {
            if (l1 > T(0)) { if (wc.ok[0]) b[0] -= l1; if (wc.ok[1]) b[1] -= l1; }
            wc.chol(b, x, nonneg);
        } else {
            wc.cd(b, x, l1, nonneg, maxit, T(0));
        }
        RK_WAVE_SYNC();
        if (wc.rel_change(x, xo) < irls_tol) break;
    }
    wc.store_x(X, j, x);
}

// ---------------------------------------------------------------------------
// Explicit-mask half-update (masked_solve_kernel for k > 64): b over the unmasked nonzeros, G_loc = G - sum over ALL masked rows
// f f^T, b -= l1, diagonal += l2, warm or zero start, CD with the relative-change stop or Cholesky + clip.
// ---------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(64) void wide_masked_solve_kernel(
    const int* __restrict__ colptr, const int* __restrict__ rowidx, const T* __restrict__ vals,
    const int* __restrict__ mask_p, const int* __restrict__ mask_i, int64_t ncols, const T* __restrict__ F,
    const T* __restrict__ Gfull, T* __restrict__ X, int k, T l1, T l2, int nonneg, int maxit, T tol, int solver_mode, int warm) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int64_t j = blockIdx.x;
    if (j >= ncols) return;
    WideCol<T> wc(smem_raw, k);
    wc.set_base(Gfull, T(0));
    T b[2] = {T(0), T(0)};
    const int as = colptr[j], ae = colptr[j + 1];
    int ms = mask_p[j];
    const int me = mask_p[j + 1];
    for (int t = as; t < ae; ++t) {
        const int row = rowidx[t];
        while (ms < me && mask_i[ms] < row) ++ms;
        const bool masked = ms < me && mask_i[ms] == row;
        if (!masked) {
            T fr[2];
            wc.load_row(F, row, fr);
            b[0] = tfma(vals[t], fr[0], b[0]);
            b[1] = tfma(vals[t], fr[1], b[1]);
        }
    }
    for (int t0 = mask_p[j]; t0 < me; t0 += WCH) {
        const int ne = me - t0 < WCH ? me - t0 : WCH;
        for (int e = 0; e < ne; ++e) {
            T fr[2];
            wc.load_row(F, mask_i[t0 + e], fr);
            wc.stage(e, fr, T(-1), T(0));
        }
        wc.template flush<false>(ne);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
        if (wc.ok[s]) { b[s] -= l1; wc.Gl[(wc.lane + 64 * s) * WKP + wc.lane + 64 * s] += l2; }
    RK_WAVE_SYNC();
    T x[2] = {T(0), T(0)};
    if (warm) wc.load_x(X, j, x);
    if (solver_mode == 1) wc.chol(b, x, nonneg);
    else wc.cd(b, x, T(0), nonneg, maxit, tol > T(0) ? tol : T(0));
    wc.store_x(X, j, x);
}

}  // namespace rk
