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


// value of lane I of the caller's 16-lane DPP row, in every lane of that row (row_newbcast, gfx90a+; folds into a VOP2 consumer)
template <int I> __device__ __forceinline__ float row_bcast(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + I, 0xf, 0xf, true));
}
// d0 += g0 * s, d1 += g1 * s with s = lane I of the row's `src`: two v_fmac_f32 with the DPP broadcast folded into their first
// operand.  (hipcc emits v_mov_b32_dpp + two fmacs for the builtin, and defers / spills the broadcasts of a half sweep.)  The
// s_nop covers the two wait states a DPP read needs after the VALU write of `src`.
template <int I> __device__ __forceinline__ void row_fmac2(float& d0, float& d1, float src, float g0, float g1) {
    asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %2, %3 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %1, %2, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf"
                 : "+v"(d0), "+v"(d1) : "v"(src), "v"(g0), "v"(g1), "n"(I));
}

template <int I> __device__ __forceinline__ int row_bcast_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + I, 0xf, 0xf, true); }
// sum over the eight lanes 8g .. 8g+7 of v, left in all eight (quad_perm xor 1, xor 2, then row_half_mirror: the other quad's sum)
__device__ __forceinline__ float sum8(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));     // quad_perm:[1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));     // quad_perm:[2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));    // row_half_mirror
    return v;
}

// One coordinate of a half sweep of the four-columns-per-wave CD solve, as ONE instruction group (see irls_nb_mfma32q_kernel):
//   step = med3(dc, -x, inf);  do += go_prev * bcast(step_prev);  own = turn ? step : own;  dc += gc * bcast(step)
// dc is the residual this half's steps are read from, `do` the other one (its update for the PREVIOUS coordinate, deferred by
// one so that two independent operations separate the med3 from the DPP read of its result: no s_nop).
template <int I> __device__ __forceinline__ float sweep_step(float& dc, float& dother, float& own, float step_prev, float nxe, float inf,
                                                             unsigned long long turn, float gc, float go_prev) {
    float step;
    asm volatile("v_med3_f32 %0, %1, %4, %5\n\t"
                 "v_fmac_f32_dpp %2, %6, %8 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
                 "v_cndmask_b32_e64 %3, %3, %0, %9\n\t"
                 "v_fmac_f32_dpp %1, %0, %7 row_newbcast:%11 row_mask:0xf bank_mask:0xf"
                 : "=&v"(step), "+v"(dc), "+v"(dother), "+v"(own)
                 : "v"(nxe), "v"(inf), "v"(step_prev), "v"(gc), "v"(go_prev), "s"(turn), "n"(I - 1), "n"(I));
    return step;
}
// the first coordinate of a half: nothing deferred yet, one wait state filled by an s_nop
template <int I> __device__ __forceinline__ float sweep_first(float& dc, float& own, float nxe, float inf, unsigned long long turn, float gc) {
    float step;
    asm volatile("v_med3_f32 %0, %1, %3, %4\n\t"
                 "v_cndmask_b32_e64 %2, %2, %0, %6\n\t"
                 "s_nop 0\n\t"
                 "v_fmac_f32_dpp %1, %0, %5 row_newbcast:%7 row_mask:0xf bank_mask:0xf"
                 : "=&v"(step), "+v"(dc), "+v"(own)
                 : "v"(nxe), "v"(inf), "v"(gc), "s"(turn), "n"(I));
    return step;
}
template <int I> __device__ __forceinline__ void row_fmac1(float& d, float src, float g) {
    asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(src), "v"(g), "n"(I));
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

// distribution weight of the two implemented IRLS losses (nnls_batch_irls.hpp:57-83): 5 = NB; 4 = GP, whose W/H
// updates use the KL weight 1 / max(mu, 1e-4) in Scalar arithmetic (fit_cpu.hpp:568-574, math/loss.hpp:176-179)
// 6 / 7 / 8 = Gamma / inverse Gaussian / Tweedie: power-variance weight min(1/mu^p, 1e6) in fp64 (math/loss.hpp:270-278)
template <class T> __device__ __forceinline__ T irls_weight_dev(int loss_type, T predicted, T theta, T power) {
    if (loss_type == 4) return T(1) / (predicted > T(1e-4) ? predicted : T(1e-4));
    if (loss_type >= 6) {
        double mu = static_cast<double>(predicted);
        const double eps = static_cast<double>(static_cast<T>(1e-15));
        mu = mu > eps ? mu : eps;
        const double p = loss_type == 6 ? 2.0 : (loss_type == 7 ? 3.0 : static_cast<double>(power));
        double w = 1.0 / pow(mu, p);
        w = w < 1e6 ? w : 1e6;
        return static_cast<T>(w);
    }
    return irls_weight_nb_dev<T>(predicted, theta);
}

// nnls_batch_irls.hpp:95-120  compute_irls_weight: distribution weight (1 for loss_type 0 = MSE) x Huber modifier of
// the Pearson residual when robust_delta > 0 (math/loss.hpp:294-303)
template <class T> __device__ __forceinline__ T irls_weight_full_dev(int loss_type, T residual, T predicted, T theta, T power,
                                                                     T robust) {
    const T w_dist = loss_type == 0 ? T(1) : irls_weight_dev<T>(loss_type, predicted, theta, power);
    if (robust > T(0)) {
        const T wd = w_dist > T(1e-15) ? w_dist : T(1e-15);
        const T abs_r = tabs(residual * sqrt(wd));
        return abs_r <= robust ? w_dist : w_dist * (robust / (abs_r + T(1e-15)));
    }
    return w_dist;
}

template <class T, int KP>   // KP in {32, 64}: features padded to KP (k <= KP), lane r = feature r
__global__ __launch_bounds__(256) void irls_nb_solve_kernel(
    const int* __restrict__ colptr, const int* __restrict__ rowidx, const T* __restrict__ vals, int64_t ncols,
    const T* __restrict__ F, const T* __restrict__ Gbase, T* __restrict__ X, int k, T l1, T l2, int nonneg,
    int cd_maxit, int irls_max_iter, T irls_tol, const T* __restrict__ theta_row, const T* __restrict__ theta_col,
    int loss_type, T power, T robust,
    unsigned long long* __restrict__ stats) {
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
    int passes = 0, nsw = 0;      // IRLS passes; CD sweeps executed over all passes (work counters)
    for (int irls = 0; irls < irls_max_iter; ++irls) {
        ++passes;
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
            const T w = irls_weight_full_dev<T>(loss_type, a - recon, recon, th, power, robust);
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
        // static coordinate sweeps (cd_static_sweeps, kernels.hip.h): the lane's Gram column read from the wave's LDS tile at
        // compile-time offsets
        nsw += cd_static_sweeps<T, KP>(b, x, gd, fok, l1, nonneg, cd_maxit, [&](auto IC) { return Gl[decltype(IC)::value * KP + ll]; });
        // IRLS convergence: max_i |x_i - x_old_i| / (|x_old_i| + 1e-12) < irls_tol
        T rel = fok ? tabs(x - x_old) / (tabs(x_old) + T(1e-12)) : T(0);
        rel = wave_max(rel);
        if (rel < irls_tol) break;
    }
    if (fok) X[j * (int64_t)k + lane] = x;
    // work counters (RCPPML_OPT_CD_COUNT_NOOP only): IRLS passes and nonzero-passes = weighted-Gram rank-1 updates
    if (stats && lane == 0) { atomicAdd(stats, (unsigned long long)passes); atomicAdd(stats + 1, (unsigned long long)passes * (unsigned long long)(ae - as)); atomicAdd(stats + 4, (unsigned long long)nsw); }
}

// ---------------------------------------------------------------------------
// fp32, k <= 32 (k % 4 == 0): the same IRLS half-update with the weighted Gram on the MATRIX cores.
//   G_w = G + F_nz diag(w - 1) F_nz^T  is a rank-nnz_j update of a 32 x 32 tile: two nonzeros fill the two K-slots of one
//   v_mfma_f32_32x32x2_f32 (A operand = (w_t - 1) f_t, B operand = f_t), against 32 shuffles + 32 fmas per nonzero in the
//   row-in-registers form above.  The column is processed in chunks of 32 nonzeros:
//   phase A (lane = nonzero t, half h of its features): coalesced (row, value) load, 16-byte gathers of F(row, :), the
//     reconstruction f_t . x as 16 in-lane fmas + one half swap, the NB weight in fp64 -- all 32 nonzeros IN PARALLEL
//     (the wave-per-nonzero form evaluates one weight per 64 lanes) -- and f_t, (w_t - 1), w_t a_t parked in LDS;
//   phase B (lane = feature r, K-slot kk): one conflict-free ds_read_b32 of f_t[r] per lane feeds both MFMA operands and
//     the weighted right-hand side b_w[r] += w_t a_t f_t[r].
//   The CD solve is unchanged (exact sequential sweep with ballot skipping); G_w is written from the accumulator tile
//   straight into its LDS slab, which aliases the staging buffer.
// ---------------------------------------------------------------------------
// LT >= 0: the loss is known at compile time and there is no robust modifier (LT = 5: negative binomial, what C5 runs) -- the
// weight function then holds neither the fp64 pow() of the power-variance losses nor the Huber branch, whose registers the
// generic instantiation (LT = -1) pays for in every lane whatever loss it runs (64-VGPR budget at eight waves per SIMD).
template <int LT>
static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void irls_nb_mfma32_kernel(
    const int* __restrict__ colptr, const int* __restrict__ rowidx, const float* __restrict__ vals, int64_t ncols,
    const float* __restrict__ F, const float* __restrict__ Gbase, float* __restrict__ X, int k, float l1, float l2,
    int nonneg, int cd_maxit, int irls_max_iter, float irls_tol, const float* __restrict__ theta_row,
    const float* __restrict__ theta_col, int loss_type, float power, float robust,
    unsigned long long* __restrict__ stats) {
    constexpr int KP = 32, CH = 32, FS = 36;          // FS: padded row stride of the staged F rows (bank spread)
    constexpr int WAVE_FLOATS = CH * FS + 2 * CH + KP;  // staged rows | (w-1, w a) pairs | x
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* Fst = reinterpret_cast<float*>(smem_raw) + (size_t)wave * WAVE_FLOATS;
    float2* sc = reinterpret_cast<float2*>(Fst + CH * FS);
    float* xs = Fst + CH * FS + 2 * CH;
    float* Gl = Fst;                                    // [c][r], KP*KP <= CH*FS: reused once the Gram is complete
    const int64_t j = (int64_t)blockIdx.x * 4 + wave;
    if (j >= ncols) return;
    const int r = lane & 31, hh = lane >> 5;            // phase A: nonzero r, feature half hh; phase B: feature r, K-slot hh
    const int g8 = lane >> 3, p8 = lane & 7;            // gather: group of eight lanes = one row of F, lane = 16-byte piece
    const bool pok = 4 * p8 < k;
    const char* Fp = reinterpret_cast<const char*>(F) + 16 * p8;    // the lane's piece of row 0
    const bool fok = lane < k;
    const bool lin = lane < KP;
    const int ll = lin ? lane : 0;
    const int as = colptr[j], ae = colptr[j + 1];
    const float th_col = theta_col ? theta_col[j] : 0.f;
    float x = 0.f;                                      // nnls_batch_irls.hpp:482-483  H.setZero(): no warm start
    int passes = 0, nsw = 0;      // IRLS passes; CD sweeps executed over all passes (work counters)
    for (int irls = 0; irls < irls_max_iter; ++irls) {
        ++passes;
        // accumulator tile <- base Gram (identity padding), C/D map: col = lane&31, row = (v&3) + 8(v>>2) + 4(lane>>5)
        // (four 16-byte loads off one per-lane pointer: k % 4 == 0 keeps a group of four rows on one side of k)
        f32x16 acc;
        {
            const float* gb = Gbase + (int64_t)r * k + 4 * hh;
            asm volatile("" : "+v"(gb));                 // per pass, not 16 hoisted addresses
#pragma unroll
            for (int v4 = 0; v4 < 4; ++v4) {
                const int gi0 = 8 * v4 + 4 * hh;
                const float4 g = (gi0 < k && r < k) ? *reinterpret_cast<const float4*>(gb + 8 * v4) : make_float4(0.f, 0.f, 0.f, 0.f);
                const bool pad = r >= k;
                acc[4 * v4 + 0] = pad && gi0 + 0 == r ? 1.f : g.x;
                acc[4 * v4 + 1] = pad && gi0 + 1 == r ? 1.f : g.y;
                acc[4 * v4 + 2] = pad && gi0 + 2 == r ? 1.f : g.z;
                acc[4 * v4 + 3] = pad && gi0 + 3 == r ? 1.f : g.w;
            }
        }
        if (lin) xs[lane] = x;
        float bw = 0.f;
        __builtin_amdgcn_wave_barrier();
        const float4 xq = *reinterpret_cast<const float4*>(xs + 4 * p8);    // the lane's piece of x
        for (int t0 = as; t0 < ae; t0 += CH) {
            // ---- phase A.  Gather: the eight lanes 8g .. 8g+7 read the eight 16-byte pieces of ONE row of F (one 128-byte line per
            // group and instruction, eight lines per instruction against 32 for a lane-per-row layout); instruction i of group g takes
            // chunk row 4g + i.  The reconstruction f . x is four in-lane fmas against the lane's piece of x and a sum over the group.
            // (a lane past the column's end takes the position of the column's LAST nonzero: a valid row of F, no branch around
            //  the loads; its weight pair is (0, 0), so it adds exact zeros to G_w and b_w all the same)
            const int tt = t0 + r;
            const bool ok = tt < ae;
            const int tc = ok ? tt : ae - 1;
            const int row = rowidx[tc];
            const float a = ok ? vals[tc] : 0.f;
            float mine = 0.f;
            float4 fv4[4];
            if (k == KP) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned rowg = (unsigned)__shfl(row, 4 * g8 + i, 64);  // row of the chunk: 4g + i
                    fv4[i] = *reinterpret_cast<const float4*>(Fp + ((unsigned long long)rowg << 7));
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned rowg = (unsigned)__shfl(row, 4 * g8 + i, 64);
                    fv4[i] = pok ? *reinterpret_cast<const float4*>(Fp + (unsigned long long)rowg * (unsigned)(4 * k)) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float part = fv4[i].x * xq.x;
                part = tfma(fv4[i].y, xq.y, part);
                part = tfma(fv4[i].z, xq.z, part);
                part = tfma(fv4[i].w, xq.w, part);
                part = sum8(part);                                               // W_T.col(row).dot(x), in the group's eight lanes
                mine = (p8 & 3) == i ? part : mine;
                *reinterpret_cast<float4*>(Fst + (4 * g8 + i) * FS + 4 * p8) = fv4[i];
            }
            const float recon = __shfl(mine, 8 * (r >> 2) + (r & 3), 64);        // row r of the chunk: group r / 4, instruction r % 4
            const float th = theta_col ? th_col : (theta_row ? theta_row[row] : 0.f);
            const float w = irls_weight_full_dev<float>(LT >= 0 ? LT : loss_type, a - recon, recon, th, power, LT >= 0 ? 0.f : robust);
            if (hh == 0) sc[r] = make_float2(ok ? w - 1.f : 0.f, ok ? w * a : 0.f);
            __builtin_amdgcn_wave_barrier();
            // ---- phase B: nonzeros (2s, 2s+1) of the chunk per MFMA
            const int cnt = ae - t0 < CH ? ae - t0 : CH;
            const int nst = (cnt + 1) >> 1;
            // (groups of four: the eight LDS reads of a group are issued ahead of its MFMAs; the nonzeros past cnt are staged as zero
            //  rows with zero weights, so rounding the count up adds exact zeros)
            for (int s2 = 0; s2 < nst; s2 += 4) {
                float fv[4];
                float2 ws[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = 2 * (s2 + u) + hh;
                    fv[u] = Fst[t * FS + r];
                    ws[u] = sc[t];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    bw = tfma(ws[u].y, fv[u], bw);                                    // b_w += f * (w a)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ws[u].x * fv[u], fv[u], acc, 0, 0, 0);   // G_w += (f (w-1)) f^T
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        bw += __shfl_xor(bw, 32, 64);
        // park G_w in LDS ([c][r]; symmetric, so accumulator row i is written as slab row i)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int gi = (v & 3) + 8 * (v >> 2) + 4 * hh;
            float val = acc[v];
            if (l2 > 0.f && gi == r && gi < k) val += l2;
            Gl[gi * KP + r] = val;
        }
        __builtin_amdgcn_wave_barrier();
        // residual b_c = b_w - G_w x_old
        // (the lane keeps ITS column of G_w -- G_w(c, ll), c = 0..31 -- in registers: the sweep below picks row i of it with a
        //  wave-uniform index, i.e. a register-indexed move instead of an LDS read on the dependent chain of every step)
        const float x_old = x;
        float b = bw;
        typedef float f32x32 __attribute__((ext_vector_type(32)));
        f32x32 gcol;
#pragma unroll
        for (int c = 0; c < KP; ++c) {
            const float xc = __shfl(x_old, c, 64);
            const float gv = Gl[c * KP + ll];
            gcol[c] = gv;
            b = tfma(-gv, xc, b);
        }
        const float gd = Gl[ll * KP + ll];
        // cd_nnls_col_fixed(G_w, b_c, x, L1 inside, L2 = 0, nonneg, cd_maxit, ub = 0, tol = 0): all sweeps, static form (kernels.hip.h)
        {
            float gg[KP];
#pragma unroll
            for (int c = 0; c < KP; ++c) gg[c] = gcol[c];
            nsw += cd_static_sweeps_scaled_f32<KP>(b, x, gd, fok, l1, nonneg, cd_maxit, gg);
        }
        float rel = fok ? tabs(x - x_old) / (tabs(x_old) + 1e-12f) : 0.f;
        rel = wave_max(rel);
        __builtin_amdgcn_wave_barrier();
        if (rel < irls_tol) break;
    }
    if (fok) X[j * (int64_t)k + lane] = x;
    // work counters (RCPPML_OPT_CD_COUNT_NOOP only): IRLS passes and nonzero-passes = weighted-Gram rank-1 updates
    if (stats && lane == 0) { atomicAdd(stats, (unsigned long long)passes); atomicAdd(stats + 1, (unsigned long long)passes * (unsigned long long)(ae - as)); atomicAdd(stats + 4, (unsigned long long)nsw); }
}

// ---------------------------------------------------------------------------
// fp32, k <= 32 (k % 4 == 0), MANY columns (H side of C5: 200 000 columns of ~190 nonzeros): FOUR columns per wavefront.
// The CD solve of the kernel above spends 4 VALU operations per coordinate and sweep with half of the wave's lanes idle
// (k = 32) and one column in flight; it is VALU-issue bound there (the solves are half of the H side's time).  Here the solve
// puts one column on each 16-lane DPP row, a lane holding TWO coordinates (l and l + 16) and the two matching rows of its
// column's scaled Gram (64 registers): coordinate i's step is broadcast inside every row at once by DPP row_newbcast folded
// into the consuming v_fmac (no v_readlane / SGPR round trip), so one coordinate costs med3 + 2 fmac_dpp + 1 select for
// FOUR columns instead of four operations for one.  The weighted Grams of the four columns are built one after the other by
// the whole wave exactly as above (MFMA, chunks of 64 nonzeros = two phase-A halves so that every lane evaluates ONE NB
// weight -- in the kernel above both halves of the wave evaluate the same 32), parked in the wave's LDS slab and picked up by
// the column's row (columns l and l + 16 of the slab: 64 four-byte reads, conflict-free).
// Steps, their order and the arithmetic of a column are those of irls_nb_mfma32_kernel: results are bit-identical to it
// (tests/test_gpu_nb.py::test_quad_equals_single).  A column that converged keeps its iterate: its row holds a zero Gram and a
// zero residual from then on, so the sweeps the other columns still need do not move it.
// 128 VGPRs (four waves per SIMD = 16 columns per SIMD against 8), 9.75 KiB of LDS per wave.
// ---------------------------------------------------------------------------
template <int LT>
static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void irls_nb_mfma32q_kernel(
    const int* __restrict__ colptr, const int* __restrict__ rowidx, const float* __restrict__ vals, int64_t ncols,
    const float* __restrict__ F, const float* __restrict__ Gbase, float* __restrict__ X, int k, float l1, float l2,
    int nonneg, int cd_maxit, int irls_max_iter, float irls_tol, const float* __restrict__ theta_row,
    const float* __restrict__ theta_col, int loss_type, float power, float robust,
    unsigned long long* __restrict__ stats) {
    constexpr int KP = 32, CH = 64, FS = 36;          // FS: padded row stride of the staged F rows and of the G_w slab
    constexpr int WAVE_FLOATS = CH * FS + 2 * CH + 2 * KP;  // staged rows | (w-1, w a) pairs | x | b_w
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* Fst = reinterpret_cast<float*>(smem_raw) + (size_t)wave * WAVE_FLOATS;
    float2* sc = reinterpret_cast<float2*>(Fst + CH * FS);
    float* xs = Fst + CH * FS + 2 * CH;
    float* bws = xs + KP;
    float* Gl = Fst;                                    // [col][row], stride FS: reused once a column's Gram is complete
    const int64_t jw = ((int64_t)blockIdx.x * 4 + wave) * 4;      // first of this wave's four columns
    if (jw >= ncols) return;
    const int r = lane & 31, hh = lane >> 5;            // Gram phases: nonzero / feature r, half hh
    const int qme = lane >> 4, l = lane & 15;           // solve: column jw + qme, coordinates l and l + 16
    const int g8 = lane >> 3, p8 = lane & 7;            // gather: group of eight lanes = one row of F, lane = 16-byte piece
    const bool pok = 4 * p8 < k;
    const int64_t jme = jw + qme;
    const bool fok0 = l < k, fok1 = l + 16 < k;
    bool active = jme < ncols;                          // uniform over a row
    float x0 = 0.f, x1 = 0.f;                           // nnls_batch_irls.hpp:482-483  H.setZero(): no warm start
    int passes = 0, nsw = 0;      // IRLS passes; CD sweeps executed over all passes (work counters)
    float ng0[KP], ng1[KP];                             // rows l and l + 16 of the column's Gram, then of -G_w / G_ii
#pragma unroll
    for (int c = 0; c < KP; ++c) { ng0[c] = 0.f; ng1[c] = 0.f; }
    const float pinf = __builtin_inff();
    const float inf_rt = cd_maxit >= 0 ? pinf : 0.f;
    for (int irls = 0; irls < irls_max_iter; ++irls) {
        const unsigned long long act = __ballot(active);
        if (!act) break;
        if (active) ++passes;
        float b0 = 0.f, b1 = 0.f, gd0 = 0.f, gd1 = 0.f;
        for (int q = 0; q < 4; ++q) {
            if (!((act >> (16 * q)) & 1ull)) continue;                           // wave-uniform
            const int64_t j = jw + q;
            const int as = colptr[j], ae = colptr[j + 1];
            const float th_col = theta_col ? theta_col[j] : 0.f;
            if (qme == q) { xs[l] = x0; xs[l + 16] = x1; }
            // accumulator tile <- base Gram (identity padding), C/D map: col = lane&31, row = (v&3) + 8(v>>2) + 4(lane>>5)
            // (four 16-byte loads off one per-lane pointer: k % 4 == 0 keeps a group of four rows on one side of k)
            f32x16 acc;
            const float* gb = Gbase + (int64_t)r * k + 4 * hh;
            asm volatile("" : "+v"(gb));                 // per pass, not 16 hoisted addresses
#pragma unroll
            for (int v4 = 0; v4 < 4; ++v4) {
                const int gi0 = 8 * v4 + 4 * hh;
                const float4 g = (gi0 < k && r < k) ? *reinterpret_cast<const float4*>(gb + 8 * v4) : make_float4(0.f, 0.f, 0.f, 0.f);
                const bool pad = r >= k;
                acc[4 * v4 + 0] = pad && gi0 + 0 == r ? 1.f : g.x;
                acc[4 * v4 + 1] = pad && gi0 + 1 == r ? 1.f : g.y;
                acc[4 * v4 + 2] = pad && gi0 + 2 == r ? 1.f : g.z;
                acc[4 * v4 + 3] = pad && gi0 + 3 == r ? 1.f : g.w;
            }
            float bw = 0.f;
            __builtin_amdgcn_wave_barrier();
            const float4 xq = *reinterpret_cast<const float4*>(xs + 4 * p8);    // the lane's piece of x
            for (int t0 = as; t0 < ae; t0 += CH) {
                // ---- phase A: lane t owns nonzero t of the chunk (its value, its weight); the gather is the group-of-eight form of
                // the kernel above with eight instructions, instruction i of group g taking chunk row 8g + i -- so lane t finds the
                // reconstruction of ITS nonzero in its own group at instruction t % 8, no shuffle
                const int tt = t0 + lane;
                const bool ok_me = tt < ae;
                const int row_me = ok_me ? rowidx[tt] : 0;
                const float a_me = ok_me ? vals[tt] : 0.f;
                float recon_me = 0.f;
                auto four_rows = [&](auto HC) {
                    constexpr int h4 = decltype(HC)::value;
                    float4 fv4[4];
                    cd_static_for<0, 4>([&](auto IC) {
                        constexpr int i = 4 * h4 + decltype(IC)::value;
                        const int rowg = (lane & 8) ? row_bcast_i<8 + i>(row_me) : row_bcast_i<i>(row_me);      // lane 8g + i of the wave
                        fv4[i - 4 * h4] = (t0 + 8 * g8 + i < ae && pok) ? *reinterpret_cast<const float4*>(F + (int64_t)rowg * k + 4 * p8)
                                                                       : make_float4(0.f, 0.f, 0.f, 0.f);
                    });
                    cd_static_for<0, 4>([&](auto IC) {
                        constexpr int u = decltype(IC)::value, i = 4 * h4 + u;
                        float part = fv4[u].x * xq.x;
                        part = tfma(fv4[u].y, xq.y, part);
                        part = tfma(fv4[u].z, xq.z, part);
                        part = tfma(fv4[u].w, xq.w, part);
                        part = sum8(part);                                           // W_T.col(row).dot(x), in the group's eight lanes
                        recon_me = p8 == i ? part : recon_me;
                        *reinterpret_cast<float4*>(Fst + (8 * g8 + i) * FS + 4 * p8) = fv4[u];
                    });
                };
                four_rows(std::integral_constant<int, 0>{});
                __builtin_amdgcn_sched_barrier(0);                                   // two batches of four loads: 16 registers, not 32
                four_rows(std::integral_constant<int, 1>{});
                const float th = theta_col ? th_col : (theta_row ? theta_row[row_me] : 0.f);
                const float w = irls_weight_full_dev<float>(LT >= 0 ? LT : loss_type, a_me - recon_me, recon_me, th, power, LT >= 0 ? 0.f : robust);
                sc[lane] = make_float2(ok_me ? w - 1.f : 0.f, ok_me ? w * a_me : 0.f);
                __builtin_amdgcn_wave_barrier();
                // ---- phase B: nonzeros (2s, 2s+1) of the chunk per MFMA
                const int cnt = ae - t0 < CH ? ae - t0 : CH;
                const int nst = (cnt + 1) >> 1;
                // (groups of four: the eight LDS reads of a group are issued ahead of its MFMAs; the nonzeros past cnt are staged as
                //  zero rows with zero weights, so rounding the count up adds exact zeros)
                for (int s2 = 0; s2 < nst; s2 += 4) {
                    float fv[4];
                    float2 ws[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int t = 2 * (s2 + u) + hh;
                        fv[u] = Fst[t * FS + r];
                        ws[u] = sc[t];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        bw = tfma(ws[u].y, fv[u], bw);                                // b_w += f * (w a)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ws[u].x * fv[u], fv[u], acc, 0, 0, 0);   // G_w += (f (w-1)) f^T
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            bw += __shfl_xor(bw, 32, 64);
            // park G_w in the slab column by column ([col r][row gi], as the kernel above) and b_w next to it
#pragma unroll
            for (int v4 = 0; v4 < 4; ++v4) {
                const int gi0 = 8 * v4 + 4 * hh;
                float4 val = make_float4(acc[4 * v4], acc[4 * v4 + 1], acc[4 * v4 + 2], acc[4 * v4 + 3]);
                if (l2 > 0.f && r < k) {
                    if (gi0 + 0 == r) val.x += l2;
                    if (gi0 + 1 == r) val.y += l2;
                    if (gi0 + 2 == r) val.z += l2;
                    if (gi0 + 3 == r) val.w += l2;
                }
                *reinterpret_cast<float4*>(Gl + r * FS + gi0) = val;
            }
            if (lane < KP) bws[lane] = bw;
            __builtin_amdgcn_wave_barrier();
            if (qme == q) {                              // the column's row picks up COLUMNS l and l + 16 of G_w
                // (columns, as the one-column kernel reads them: the MFMA's G_w(i, j) = sum ((w-1) f_i) f_j is not bitwise symmetric)
#pragma unroll
                for (int c4 = 0; c4 < KP / 4; ++c4) {
                    const float4 u = *reinterpret_cast<const float4*>(Gl + l * FS + 4 * c4);
                    const float4 v = *reinterpret_cast<const float4*>(Gl + (l + 16) * FS + 4 * c4);
                    ng0[4 * c4] = u.x; ng0[4 * c4 + 1] = u.y; ng0[4 * c4 + 2] = u.z; ng0[4 * c4 + 3] = u.w;
                    ng1[4 * c4] = v.x; ng1[4 * c4 + 1] = v.y; ng1[4 * c4 + 2] = v.z; ng1[4 * c4 + 3] = v.w;
                }
                gd0 = Gl[l * FS + l];
                gd1 = Gl[(l + 16) * FS + l + 16];
                b0 = bws[l];
                b1 = bws[l + 16];
            }
            __builtin_amdgcn_wave_barrier();
        }
        // ---- all (still iterating) columns of the wave at once: residual b_c = b_w - G_w x_old, coordinates in order
        const float xo0 = x0, xo1 = x1;
        {
            const float nx0 = -xo0, nx1 = -xo1;          // fma(-g, x, b) = fma(g, -x, b)
            cd_static_for<0, 16>([&](auto IC) { constexpr int i = decltype(IC)::value; row_fmac2<i>(b0, b1, nx0, ng0[i], ng1[i]); });
            cd_static_for<0, 16>([&](auto IC) { constexpr int i = decltype(IC)::value; row_fmac2<i>(b0, b1, nx1, ng0[16 + i], ng1[16 + i]); });
        }
        // cd_nnls_col_fixed(G_w, b_c, x, L1 inside, L2 = 0, nonneg, cd_maxit, ub = 0, tol = 0), static scaled form (kernels.hip.h):
        // d = b / G_ii - L1, rows scaled by -1 / G_ii once; a row that is done (or dead) holds zeros and never moves
        const bool alive0 = active && fok0 && gd0 > 0.f, alive1 = active && fok1 && gd1 > 0.f;
        const float ginv0 = alive0 ? 1.f / gd0 : 0.f, ginv1 = alive1 ? 1.f / gd1 : 0.f;
#pragma unroll
        for (int c = 0; c < KP; ++c) { ng0[c] *= -ginv0; ng1[c] *= -ginv1; }
        float d0 = __builtin_fmaf(active ? b0 : 0.f, ginv0, alive0 ? -l1 : 0.f);
        float d1 = __builtin_fmaf(active ? b1 : 0.f, ginv1, alive1 ? -l1 : 0.f);
        unsigned long long lane0_of_rows = 0x0001000100010001ull;
        asm volatile("" : "+s"(lane0_of_rows));        // opaque: shifted at run time, not sixteen 64-bit literals
        bool rowdone = !active;                          // (work counter: a row's sweeps until its own fixed point)
        for (int it = 0; it < cd_maxit; ++it) {
            nsw += rowdone ? 0 : 1;
            const float xe0 = nonneg ? x0 : pinf, xe1 = nonneg ? x1 : pinf;
            float aown0 = 0.f, aown1 = 0.f;
            const float nxe0 = -xe0, nxe1 = -xe1;
            // the lane whose turn it is keeps its step: lane mask of (l == i) as ONE scalar pair shifted by a SALU operation per
            // coordinate (sixteen compare results held in SGPRs cost 32 of them and spill)
            // Four VALU operations per coordinate and no wait state: the fmac of the OTHER residual is the previous coordinate's
            // (it is not on this half's chain) and sits, with the select, between the med3 that writes the step and the DPP read of it.
            unsigned long long turn = lane0_of_rows;
            float adp = sweep_first<0>(d0, aown0, nxe0, inf_rt, turn, ng0[0]);
            cd_static_for<1, 16>([&](auto IC) {
                constexpr int i = decltype(IC)::value;
                turn <<= 1;
                adp = sweep_step<i>(d0, d1, aown0, adp, nxe0, inf_rt, turn, ng0[i], ng1[i - 1]);
            });
            row_fmac1<15>(d1, adp, ng1[15]);
            turn = lane0_of_rows;
            adp = sweep_first<0>(d1, aown1, nxe1, inf_rt, turn, ng1[16]);
            cd_static_for<1, 16>([&](auto IC) {
                constexpr int i = decltype(IC)::value;
                turn <<= 1;
                adp = sweep_step<i>(d1, d0, aown1, adp, nxe1, inf_rt, turn, ng1[16 + i], ng0[16 + i - 1]);
            });
            row_fmac1<15>(d0, adp, ng0[31]);
            const float xn0 = x0 + aown0, xn1 = x1 + aown1;
            const bool moved = xn0 != x0 || xn1 != x1;
            x0 = xn0; x1 = xn1;
            const unsigned long long mv = __ballot(moved);
            if (!mv) break;
            // a column none of whose coordinates moved is at the fixed point the one-column kernel stops at: zero residuals make
            // every later step of its row exactly zero (the columns sharing the wave may need more sweeps)
            if (!((mv >> (lane & 48)) & 0xffffull)) { d0 = 0.f; d1 = 0.f; rowdone = true; }
        }
        float rel = fok0 ? tabs(x0 - xo0) / (tabs(xo0) + 1e-12f) : 0.f;
        const float rel1 = fok1 ? tabs(x1 - xo1) / (tabs(xo1) + 1e-12f) : 0.f;
        rel = rel1 > rel ? rel1 : rel;
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            const float o = __shfl_xor(rel, off, 64);
            rel = o > rel ? o : rel;
        }
        if (rel < irls_tol) active = false;
        __builtin_amdgcn_wave_barrier();
    }
    if (jme < ncols) {
        if (fok0) X[jme * (int64_t)k + l] = x0;
        if (fok1) X[jme * (int64_t)k + l + 16] = x1;
        // work counters (RCPPML_OPT_CD_COUNT_NOOP only): IRLS passes and nonzero-passes = weighted-Gram rank-1 updates
        if (stats && l == 0) {
            const int nzc = colptr[jme + 1] - colptr[jme];
            atomicAdd(stats, (unsigned long long)passes);
            atomicAdd(stats + 1, (unsigned long long)passes * (unsigned long long)nzc);
            atomicAdd(stats + 4, (unsigned long long)nsw);
        }
    }
}

// ---------------------------------------------------------------------------
// fp32, 32 < k <= 64 (k % 4 == 0): the same on a 2 x 2 grid of 32 x 32 accumulator tiles (the lower-left tile is the
// transpose of the upper-right one and is never computed: three MFMAs per pair of nonzeros).  Lane = feature in the sweep
// (all 64 lanes busy); in phase A a lane takes nonzero r and the 32 features of half hh (eight 16-byte gathers, the
// reconstruction f . x as 32 in-lane fmas + one half swap).  16 KiB of LDS per wave for G_w (the staged rows alias it):
// two blocks per CU -- an order of magnitude above the row-in-registers form this rank range used before (64 shuffles +
// 64 fmas per nonzero and lane), not the occupancy of the k <= 32 kernel.
// ---------------------------------------------------------------------------
static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void irls_nb_mfma32x2_kernel(
    const int* __restrict__ colptr, const int* __restrict__ rowidx, const float* __restrict__ vals, int64_t ncols,
    const float* __restrict__ F, const float* __restrict__ Gbase, float* __restrict__ X, int k, float l1, float l2,
    int nonneg, int cd_maxit, int irls_max_iter, float irls_tol, const float* __restrict__ theta_row,
    const float* __restrict__ theta_col, int loss_type, float power, float robust,
    unsigned long long* __restrict__ stats) {
    constexpr int KP = 64, CH = 32, FS = 68;          // FS: padded stride of a staged F row (64 features + bank spread)
    constexpr int GW = KP * KP;                        // G_w slab; the staged rows (CH * FS = 2176 floats) alias its head
    constexpr int WAVE_FLOATS = GW + 2 * CH + KP;      // G_w | (w-1, w a) pairs | x
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* Gl = reinterpret_cast<float*>(smem_raw) + (size_t)wave * WAVE_FLOATS;   // [c][r]
    float* Fst = Gl;
    float2* sc = reinterpret_cast<float2*>(Gl + GW);
    float* xs = Gl + GW + 2 * CH;
    const int64_t j = (int64_t)blockIdx.x * 4 + wave;
    if (j >= ncols) return;
    const int r = lane & 31, hh = lane >> 5;            // phase A: nonzero r, feature half hh; phase B: feature r (+32), K-slot hh
    const bool fok = lane < k;
    const int as = colptr[j], ae = colptr[j + 1];
    const float th_col = theta_col ? theta_col[j] : 0.f;
    float x = 0.f;                                      // nnls_batch_irls.hpp:482-483  H.setZero(): no warm start
    int passes = 0, nsw = 0;      // IRLS passes; CD sweeps executed over all passes (work counters)
    for (int irls = 0; irls < irls_max_iter; ++irls) {
        ++passes;
        // accumulator tiles <- base Gram (identity padding); C/D map of a tile: col = lane&31, row = (v&3) + 8(v>>2) + 4(lane>>5)
        f32x16 a00, a01, a11;                           // rows 0-31 x cols 0-31 | rows 0-31 x cols 32-63 | rows 32-63 x cols 32-63
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int gi = (v & 3) + 8 * (v >> 2) + 4 * hh, gj = r;
            auto base = [&](int i2, int j2) { return (i2 < k && j2 < k) ? Gbase[(int64_t)j2 * k + i2] : (i2 == j2 ? 1.f : 0.f); };
            a00[v] = base(gi, gj); a01[v] = base(gi, gj + 32); a11[v] = base(gi + 32, gj + 32);
        }
        xs[lane] = x;
        float bw0 = 0.f, bw1 = 0.f;                     // b_w of feature r and of feature 32 + r (this lane's K-slot share)
        __builtin_amdgcn_wave_barrier();
        for (int t0 = as; t0 < ae; t0 += CH) {
            // ---- phase A
            const int tt = t0 + r;
            const bool ok = tt < ae;
            const int row = ok ? rowidx[tt] : 0;
            const float a = ok ? vals[tt] : 0.f;
            const float* fsrc = F + (int64_t)row * k + 32 * hh;
            float4 fv4[8];
            float part = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int c0 = 32 * hh + 4 * q;
                fv4[q] = (ok && c0 < k) ? *reinterpret_cast<const float4*>(fsrc + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 xv = *reinterpret_cast<const float4*>(xs + c0);
                part = tfma(fv4[q].x, xv.x, part);
                part = tfma(fv4[q].y, xv.y, part);
                part = tfma(fv4[q].z, xv.z, part);
                part = tfma(fv4[q].w, xv.w, part);
            }
            const float recon = part + __shfl_xor(part, 32, 64);                 // W_T.col(row).dot(x)
            const float th = theta_col ? th_col : (theta_row ? theta_row[row] : 0.f);
            const float w = irls_weight_full_dev<float>(loss_type, a - recon, recon, th, power, robust);
#pragma unroll
            for (int q = 0; q < 8; ++q) *reinterpret_cast<float4*>(Fst + r * FS + 32 * hh + 4 * q) = fv4[q];
            if (hh == 0) sc[r] = make_float2(ok ? w - 1.f : 0.f, ok ? w * a : 0.f);
            __builtin_amdgcn_wave_barrier();
            // ---- phase B: nonzeros (2s, 2s+1) of the chunk per MFMA triple
            const int cnt = ae - t0 < CH ? ae - t0 : CH;
            const int nst = (cnt + 1) >> 1;
#pragma unroll 2
            for (int s2 = 0; s2 < nst; ++s2) {
                const int t = 2 * s2 + hh;
                const float f0 = Fst[t * FS + r], f1 = Fst[t * FS + 32 + r];
                const float2 ws = sc[t];
                bw0 = tfma(ws.y, f0, bw0);                                        // b_w += f * (w a)
                bw1 = tfma(ws.y, f1, bw1);
                const float s0 = ws.x * f0, s1 = ws.x * f1;
                a00 = __builtin_amdgcn_mfma_f32_32x32x2f32(s0, f0, a00, 0, 0, 0);   // G_w += (f (w-1)) f^T, tile by tile
                a01 = __builtin_amdgcn_mfma_f32_32x32x2f32(s0, f1, a01, 0, 0, 0);
                a11 = __builtin_amdgcn_mfma_f32_32x32x2f32(s1, f1, a11, 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
        }
        bw0 += __shfl_xor(bw0, 32, 64);
        bw1 += __shfl_xor(bw1, 32, 64);
        const float bw = hh ? bw1 : bw0;                 // lane = feature from here on
        // park G_w in LDS ([c][r]); the lower-left tile is the transpose of a01
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int gi = (v & 3) + 8 * (v >> 2) + 4 * hh;      // row inside the tile; column inside the tile = r
            float d00 = a00[v], d11 = a11[v];
            if (l2 > 0.f && gi == r) { if (gi < k) d00 += l2; if (gi + 32 < k) d11 += l2; }
            Gl[gi * KP + r] = d00;                               // symmetric tile: stored as computed
            Gl[(gi + 32) * KP + 32 + r] = d11;
            Gl[(32 + r) * KP + gi] = a01[v];                     // G_w(row gi, col 32 + r) at [c = 32 + r][r' = gi]
            Gl[gi * KP + 32 + r] = a01[v];                       // and its mirror G_w(row 32 + r, col gi) at [c = gi][r' = 32 + r]
        }
        __builtin_amdgcn_wave_barrier();
        // residual b_c = b_w - G_w x_old; the lane keeps its column of G_w in registers for the sweep
        const float x_old = x;
        float b = bw;
        // (two 32-element vectors: hipcc indexes a 32-element vector with s_set_gpr_idx, a 64-element one through scratch)
        typedef float f32x32 __attribute__((ext_vector_type(32)));
        f32x32 gcol0, gcol1;
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            const float xc = __shfl(x_old, c, 64);
            const float gv = Gl[c * KP + lane];
            gcol0[c] = gv;
            b = tfma(-gv, xc, b);
        }
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            const float xc = __shfl(x_old, 32 + c, 64);
            const float gv = Gl[(32 + c) * KP + lane];
            gcol1[c] = gv;
            b = tfma(-gv, xc, b);
        }
        const float gd = Gl[lane * KP + lane];
        // cd_nnls_col_fixed(G_w, b_c, x, L1 inside, L2 = 0, nonneg, cd_maxit, ub = 0, tol = 0): all sweeps, static form (kernels.hip.h)
        {
            float gg[KP];
#pragma unroll
            for (int c = 0; c < 32; ++c) { gg[c] = gcol0[c]; gg[32 + c] = gcol1[c]; }
            nsw += cd_static_sweeps_scaled_f32<KP>(b, x, gd, fok, l1, nonneg, cd_maxit, gg);
        }
        float rel = fok ? tabs(x - x_old) / (tabs(x_old) + 1e-12f) : 0.f;
        rel = wave_max(rel);
        __builtin_amdgcn_wave_barrier();
        if (rel < irls_tol) break;
    }
    if (fok) X[j * (int64_t)k + lane] = x;
    // work counters (RCPPML_OPT_CD_COUNT_NOOP only): IRLS passes and nonzero-passes = weighted-Gram rank-1 updates
    if (stats && lane == 0) { atomicAdd(stats, (unsigned long long)passes); atomicAdd(stats + 1, (unsigned long long)passes * (unsigned long long)(ae - as)); atomicAdd(stats + 4, (unsigned long long)nsw); }
}

// ---------------------------------------------------------------------------
// fp64, k <= 32 (k % 2 == 0): the same kernel on v_mfma_f64_16x16x4_f64 -- the 32 x 32 weighted Gram is 2 x 2 tiles of
// 16 x 16, FOUR nonzeros fill the four K-slots of an instruction (A operand = (w_t - 1) f_t[16 ti + r], B operand =
// f_t[16 tj + r], lane (r = lane&15, kk = lane>>4) serves nonzero 4s + kk), C/D map col = lane&15, row = (lane>>4) + 4v.
// Phase A / CD solve as in irls_nb_mfma32_kernel (16-byte gathers = 2 doubles).
// ---------------------------------------------------------------------------
static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void irls_nb_mfma64_kernel(
    const int* __restrict__ colptr, const int* __restrict__ rowidx, const double* __restrict__ vals, int64_t ncols,
    const double* __restrict__ F, const double* __restrict__ Gbase, double* __restrict__ X, int k, double l1, double l2,
    int nonneg, int cd_maxit, int irls_max_iter, double irls_tol, const double* __restrict__ theta_row,
    const double* __restrict__ theta_col, int loss_type, double power, double robust,
    unsigned long long* __restrict__ stats) {
    constexpr int KP = 32, CH = 32, FS = 34;          // FS: padded row stride (doubles) of the staged F rows
    constexpr int WAVE_DOUBLES = CH * FS + 2 * CH + KP;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* Fst = reinterpret_cast<double*>(smem_raw) + (size_t)wave * WAVE_DOUBLES;
    double2* sc = reinterpret_cast<double2*>(Fst + CH * FS);
    double* xs = Fst + CH * FS + 2 * CH;
    double* Gl = Fst;                                   // [c][r], KP*KP <= CH*FS
    const int64_t j = (int64_t)blockIdx.x * 4 + wave;
    if (j >= ncols) return;
    const int r = lane & 31, hh = lane >> 5;            // phase A: nonzero r, feature half hh
    const int r16 = lane & 15, kk = lane >> 4;          // phase B: feature slot r16, K-slot kk
    const bool fok = lane < k;
    const bool lin = lane < KP;
    const int ll = lin ? lane : 0;
    const int as = colptr[j], ae = colptr[j + 1];
    const double th_col = theta_col ? theta_col[j] : 0.0;
    double x = 0.0;
    int passes = 0, nsw = 0;      // IRLS passes; CD sweeps executed over all passes (work counters)
    for (int irls = 0; irls < irls_max_iter; ++irls) {
        ++passes;
        f64x4 acc[2][2];
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int gi = 16 * ti + kk + 4 * v, gj = 16 * tj + r16;
                    acc[ti][tj][v] = (gi < k && gj < k) ? Gbase[(int64_t)gj * k + gi] : (gi == gj ? 1.0 : 0.0);
                }
        if (lin) xs[lane] = x;
        double bw[2] = {0.0, 0.0};
        for (int t0 = as; t0 < ae; t0 += CH) {
            // ---- phase A
            const int tt = t0 + r;
            const bool ok = tt < ae;
            const int row = ok ? rowidx[tt] : 0;
            const double a = ok ? vals[tt] : 0.0;
            const double* fsrc = F + (int64_t)row * k + 16 * hh;
            double2 fv2[8];
            double part = 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int c0 = 16 * hh + 2 * q;
                fv2[q] = (ok && c0 < k) ? *reinterpret_cast<const double2*>(fsrc + 2 * q) : make_double2(0.0, 0.0);
                const double2 xv = *reinterpret_cast<const double2*>(xs + c0);
                part = tfma(fv2[q].x, xv.x, part);
                part = tfma(fv2[q].y, xv.y, part);
            }
            const double recon = part + __shfl_xor(part, 32, 64);
            const double th = theta_col ? th_col : (theta_row ? theta_row[row] : 0.0);
            const double w = irls_weight_full_dev<double>(loss_type, a - recon, recon, th, power, robust);
#pragma unroll
            for (int q = 0; q < 8; ++q) *reinterpret_cast<double2*>(Fst + r * FS + 16 * hh + 2 * q) = fv2[q];
            if (hh == 0) sc[r] = make_double2(ok ? w - 1.0 : 0.0, ok ? w * a : 0.0);
            __builtin_amdgcn_wave_barrier();
            // ---- phase B: nonzeros 4s .. 4s+3 of the chunk per MFMA step
            const int cnt = ae - t0 < CH ? ae - t0 : CH;
            const int nst = (cnt + 3) >> 2;
#pragma unroll 2
            for (int s4 = 0; s4 < nst; ++s4) {
                const int t = 4 * s4 + kk;
                const double f0 = Fst[t * FS + r16], f1 = Fst[t * FS + 16 + r16];
                const double2 ws = sc[t];
                bw[0] = tfma(ws.y, f0, bw[0]);
                bw[1] = tfma(ws.y, f1, bw[1]);
                const double a0 = ws.x * f0, a1 = ws.x * f1;
                acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, f0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, f1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, f0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, f1, acc[1][1], 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
        }
        // b_w[16 ti + r16]: sum over the four K-slot groups, then gather both halves into lane = feature order
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            bw[ti] += __shfl_xor(bw[ti], 16, 64);
            bw[ti] += __shfl_xor(bw[ti], 32, 64);
        }
        const double bwt = (lane & 16) ? bw[1] : bw[0];     // own registers: lane l < 32 has r16 = l & 15 and wants feature l
        // park G_w in LDS ([c][r]; symmetric)
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int gi = 16 * ti + kk + 4 * v, gj = 16 * tj + r16;
                    double val = acc[ti][tj][v];
                    if (l2 > 0.0 && gi == gj && gi < k) val += l2;
                    Gl[gi * KP + gj] = val;
                }
        __builtin_amdgcn_wave_barrier();
        const double x_old = x;
        double b = bwt;
#pragma unroll 8
        for (int c = 0; c < KP; ++c) {
            const double xc = __shfl(x_old, c, 64);
            b = tfma(-Gl[c * KP + ll], xc, b);
        }
        const double gd = Gl[ll * KP + ll];
        nsw += cd_static_sweeps<double, KP>(b, x, gd, fok, l1, nonneg, cd_maxit, [&](auto IC) { return Gl[decltype(IC)::value * KP + ll]; });
        double rel = fok ? tabs(x - x_old) / (tabs(x_old) + 1e-12) : 0.0;
        rel = wave_max(rel);
        __builtin_amdgcn_wave_barrier();
        if (rel < irls_tol) break;
    }
    if (fok) X[j * (int64_t)k + lane] = x;
    // work counters (RCPPML_OPT_CD_COUNT_NOOP only): IRLS passes and nonzero-passes = weighted-Gram rank-1 updates
    if (stats && lane == 0) { atomicAdd(stats, (unsigned long long)passes); atomicAdd(stats + 1, (unsigned long long)passes * (unsigned long long)(ae - as)); atomicAdd(stats + 4, (unsigned long long)nsw); }
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
    const bool fok = lane < k, fok2 = lane + 64 < k;                     // lane holds features lane and lane + 64 (k <= 128)
    const T wd = fok ? W_T[i * (int64_t)k + lane] * d[lane] : T(0);     // apply_scaling(W_Td, d)
    const T wd2 = fok2 ? W_T[i * (int64_t)k + lane + 64] * d[lane + 64] : T(0);
    // one lane per nonzero: the row of H is gathered with 16-byte loads and dotted in-lane against Wd_i (LDS broadcast)
    __shared__ T wds[4][128];
    wds[wave][lane] = wd;
    wds[wave][lane + 64] = wd2;
    __builtin_amdgcn_wave_barrier();
    constexpr int VEC = 16 / sizeof(T);
    typedef typename VecT<T, VEC>::type V;
    const bool vec_ok = (k % VEC == 0) && (reinterpret_cast<uintptr_t>(H) % 16 == 0);
    double s_mu2 = 0.0, s_res2 = 0.0;
    for (int t = tp[i] + lane; t < tp[i + 1]; t += 64) {
        const int col = ti[t];
        const T* hr = H + (int64_t)col * k;
        T dot = T(0);
        if (vec_ok) {
            for (int c = 0; c < k; c += VEC) {
                const V v = *reinterpret_cast<const V*>(hr + c);
#pragma unroll
                for (int e = 0; e < VEC; ++e) dot = tfma(wds[wave][c + e], v[e], dot);
            }
        } else {
            for (int c = 0; c < k; ++c) dot = tfma(wds[wave][c], hr[c], dot);
        }
        const double y = static_cast<double>(tx[t]);
        double mu = static_cast<double>(dot);
        mu = mu > 1e-10 ? mu : 1e-10;
        const double resid = y - mu;
        s_mu2 += mu * mu;
        s_res2 += resid * resid;
    }
    s_mu2 = wave_sum(s_mu2);
    s_res2 = wave_sum(s_res2);
    const T tm = wave_sum((fok ? wd * h_rs[lane] : T(0)) + (fok2 ? wd2 * h_rs[lane + 64] : T(0)));
    const double total_mu = static_cast<double>(tm);
    // total_mu_sq = sum_ab Wd_a G_H(a,b) Wd_b  (fp64 accumulation as the reference)
    double acc = 0.0;
    for (int b2 = 0; b2 < k; ++b2) {
        const double wb = static_cast<double>(wds[wave][b2]);
        if (fok) acc += static_cast<double>(wd) * static_cast<double>(G_H[(int64_t)b2 * k + lane]) * wb;
        if (fok2) acc += static_cast<double>(wd2) * static_cast<double>(G_H[(int64_t)b2 * k + lane + 64]) * wb;
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

// NB size update AND the NB negative log-likelihood in one kernel (PER_ROW dispersion, no robust modifier): fit_cpu.hpp:1094-1265
// followed by explicit_loss.hpp:53-77 + math/loss.hpp:415-426 with the UPDATED sizes, as the fit loop orders them.  One wavefront
// per row i of A: the first pass over the row's nonzeros is nb_size_rows_kernel's (prediction = Wd_i . H_col, in-lane) and parks
// every prediction in mu_cache (each lane re-reads only what it wrote); the row's new size r_i is then known to the whole wave,
// and the second pass evaluates the likelihood terms from the parked predictions -- no second gather of factor rows (the
// separate loss kernel re-gathers 128 B per nonzero), and lgamma(r_i) once per row instead of once per nonzero.  The terms are
// those of nb_loss_lane_kernel (same prediction arithmetic: (W d) . H in feature order, same fp64 formula, each term cast to
// Scalar); only the order of the fp64 sum differs (by row instead of by column).
template <class T>
__global__ __launch_bounds__(256) void nb_size_loss_rows_kernel(
    const int* __restrict__ tp, const int* __restrict__ ti, const T* __restrict__ tx, int64_t m,
    const T* __restrict__ W_T, const T* __restrict__ d, const T* __restrict__ H, const T* __restrict__ h_rs,
    const T* __restrict__ G_H, int k, double r_min, double r_max, T* __restrict__ nb_size, T* __restrict__ mu_cache,
    double* __restrict__ partial) {
    __shared__ T wds[4][128];
    __shared__ double sh[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    double nll_sum = 0.0;
    if (i < m) {
        const bool fok = lane < k, fok2 = lane + 64 < k;                     // lane holds features lane and lane + 64 (k <= 128)
        const T wd = fok ? W_T[i * (int64_t)k + lane] * d[lane] : T(0);     // apply_scaling(W_Td, d)
        const T wd2 = fok2 ? W_T[i * (int64_t)k + lane + 64] * d[lane + 64] : T(0);
        wds[wave][lane] = wd;
        wds[wave][lane + 64] = wd2;
        __builtin_amdgcn_wave_barrier();
        constexpr int VEC = 16 / sizeof(T);
        typedef typename VecT<T, VEC>::type V;
        const bool vec_ok = (k % VEC == 0) && (reinterpret_cast<uintptr_t>(H) % 16 == 0);
        const int ts = tp[i], te = tp[i + 1];
        double s_mu2 = 0.0, s_res2 = 0.0;
        for (int t = ts + lane; t < te; t += 64) {
            const int col = ti[t];
            const T* hr = H + (int64_t)col * k;
            T dot = T(0);
            if (vec_ok) {
                for (int c = 0; c < k; c += VEC) {
                    const V v = *reinterpret_cast<const V*>(hr + c);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) dot = tfma(wds[wave][c + e], v[e], dot);
                }
            } else {
                for (int c = 0; c < k; ++c) dot = tfma(wds[wave][c], hr[c], dot);
            }
            mu_cache[t] = dot;
            const double y = static_cast<double>(tx[t]);
            double mu = static_cast<double>(dot);
            mu = mu > 1e-10 ? mu : 1e-10;
            const double resid = y - mu;
            s_mu2 += mu * mu;
            s_res2 += resid * resid;
        }
        s_mu2 = wave_sum(s_mu2);
        s_res2 = wave_sum(s_res2);
        const T tm = wave_sum((fok ? wd * h_rs[lane] : T(0)) + (fok2 ? wd2 * h_rs[lane + 64] : T(0)));
        const double total_mu = static_cast<double>(tm);
        double acc = 0.0;                                                    // total_mu_sq = Wd^T G_H Wd, fp64 as the reference
        for (int b2 = 0; b2 < k; ++b2) {
            const double wb = static_cast<double>(wds[wave][b2]);
            if (fok) acc += static_cast<double>(wd) * static_cast<double>(G_H[(int64_t)b2 * k + lane]) * wb;
            if (fok2) acc += static_cast<double>(wd2) * static_cast<double>(G_H[(int64_t)b2 * k + lane + 64]) * wb;
        }
        const double total_mu_sq = wave_sum(acc);
        // every lane takes the decision lane 0 takes in nb_size_rows_kernel (wave_sum leaves the same value in all lanes)
        T size_new = nb_size[i];
        {
            const double total_resid_sq = s_res2 + (total_mu_sq - s_mu2);
            const double excess = total_resid_sq - total_mu;
            if (excess > 1e-10 && total_mu_sq > 1e-10) {
                double r_new = total_mu_sq / excess;
                r_new = r_new < r_max ? r_new : r_max;
                r_new = r_new > r_min ? r_new : r_min;
                if (isfinite(r_new)) size_new = static_cast<T>(r_new);
            } else {
                size_new = static_cast<T>(r_max);
            }
        }
        if (lane == 0) nb_size[i] = size_new;
        // ---- likelihood terms of the row with the updated size (math/loss.hpp:415-426)
        const double th = static_cast<double>(size_new);
        const double r = th > 1e-10 ? th : 1e-10;
        const double lg_r = lgamma(r);
        for (int t = ts + lane; t < te; t += 64) {
            const double y = static_cast<double>(tx[t]);
            double mu = static_cast<double>(mu_cache[t]);
            mu = mu > 1e-10 ? mu : 1e-10;
            const double nll = -lgamma(y + r) + lg_r - r * log(r / (r + mu)) - y * log(mu / (r + mu));
            nll_sum += static_cast<double>(static_cast<T>(nll));             // the reference casts each term to Scalar
        }
    }
    nll_sum = wave_sum(nll_sum);
    if (lane == 0) sh[wave] = nll_sum;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// Dispersion estimators of the other IRLS losses, one wavefront per ROW i of A (= column i of A^T), one lane per nonzero
// (the gather / in-lane dot of nb_size_rows_kernel):
//   loss_type 4      GP theta, auxiliary-function (MM) update, nmf/fit_cpu.hpp:914-1001: the Scalar predictions s of the
//                    row's nonzeros are parked in s_cache (each lane re-reads what it wrote), then five inner passes
//                    accumulate alpha / gamma in fp64 with theta rounded through Scalar between passes, as the reference;
//   loss_type 6/7/8  Pearson phi over the positive nonzeros, nmf/fit_cpu.hpp:1561-1661, clamped to [lo, hi].
template <class T>
__global__ __launch_bounds__(256) void dispersion_rows_kernel(
    const int* __restrict__ tp, const int* __restrict__ ti, const T* __restrict__ tx, int64_t m,
    const T* __restrict__ W_T, const T* __restrict__ d, const T* __restrict__ H, const T* __restrict__ h_rs, int k,
    int loss_type, double power, double lo, double hi, T* __restrict__ s_cache, T* __restrict__ theta) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    if (i >= m) return;
    const bool fok = lane < k, fok2 = lane + 64 < k;                     // lane holds features lane and lane + 64 (k <= 128)
    const T wd = fok ? W_T[i * (int64_t)k + lane] * d[lane] : T(0);     // apply_scaling(W_Td, d)
    const T wd2 = fok2 ? W_T[i * (int64_t)k + lane + 64] * d[lane + 64] : T(0);
    __shared__ T wds[4][128];
    wds[wave][lane] = wd;
    wds[wave][lane + 64] = wd2;
    __builtin_amdgcn_wave_barrier();
    constexpr int VEC = 16 / sizeof(T);
    typedef typename VecT<T, VEC>::type V;
    const bool vec_ok = (k % VEC == 0) && (reinterpret_cast<uintptr_t>(H) % 16 == 0);
    const int ts = tp[i], te = tp[i + 1];
    const bool is_gp = loss_type == 4;
    const double var_power = loss_type == 8 ? power : (loss_type == 6 ? 2.0 : 3.0);
    double acc0 = 0.0, acc1 = 0.0;      // GP: sum_y, n_nz      phi: sum of Pearson terms, count
    for (int t = ts + lane; t < te; t += 64) {
        const int col = ti[t];
        const T* hr = H + (int64_t)col * k;
        T dot = T(0);
        if (vec_ok) {
            for (int c = 0; c < k; c += VEC) {
                const V v = *reinterpret_cast<const V*>(hr + c);
#pragma unroll
                for (int e = 0; e < VEC; ++e) dot = tfma(wds[wave][c + e], v[e], dot);
            }
        } else {
            for (int c = 0; c < k; ++c) dot = tfma(wds[wave][c], hr[c], dot);
        }
        const double y = static_cast<double>(tx[t]);
        if (is_gp) {
            s_cache[t] = dot;
            acc0 += y;
            if (y >= 1.0) acc1 += 1.0;
        } else if (y > 0.0) {
            double mu = static_cast<double>(dot);
            mu = mu > 1e-10 ? mu : 1e-10;
            const double resid = y - mu;
            double v_mu = pow(mu, var_power);
            v_mu = v_mu > 1e-20 ? v_mu : 1e-20;
            acc0 += (resid * resid) / v_mu;
            acc1 += 1.0;
        }
    }
    acc0 = wave_sum(acc0);
    acc1 = wave_sum(acc1);
    if (!is_gp) {
        if (lane == 0 && acc1 > 0.0) {
            double pn = acc0 / acc1;
            pn = pn < hi ? pn : hi;
            pn = pn > lo ? pn : lo;
            if (isfinite(pn)) theta[i] = static_cast<T>(pn);
        }
        return;
    }
    const double sum_s = static_cast<double>(wave_sum((fok ? wd * h_rs[lane] : T(0)) + (fok2 ? wd2 * h_rs[lane + 64] : T(0))));   // Gram trick, :935-938
    T th_s = theta[i];
    for (int mm = 0; mm < 5; ++mm) {                                                          // THETA_INNER_ITERS
        const double th = static_cast<double>(th_s);
        double alpha = 0.0, gamma = 0.0;
        for (int t = ts + lane; t < te; t += 64) {
            const double y = static_cast<double>(tx[t]);
            if (y >= 1.0) {
                double sv = static_cast<double>(s_cache[t]);
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
        const double a = alpha + acc1;
        const double b = (acc0 - sum_s) - gamma + a;
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

// GLOBAL dispersion: every entry <- the mean (stat 0; GP, fit_cpu.hpp:1005-1008) or <- src[m/2] of the SORTED copy
// (stat 1: nth_element at m/2; NB :1257-1262, phi :1664-1669).  Single block.
// The median is found by an 8-bit radix SELECT on order-preserving integer keys (no sort: the reference only needs the element
// nth_element would put at m / 2): per digit, a 256-bin histogram of the candidates that share the prefix found so far.
template <class T> struct OrdKey;
template <> struct OrdKey<float> {
    typedef unsigned int U;
    static __device__ __forceinline__ U enc(float v) { const U b = __float_as_uint(v); return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u); }
    static __device__ __forceinline__ float dec(U k) { return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu)); }
};
template <> struct OrdKey<double> {
    typedef unsigned long long U;
    static __device__ __forceinline__ U enc(double v) { const U b = (U)__double_as_longlong(v); return b ^ ((b >> 63) ? ~0ull : 0x8000000000000000ull); }
    static __device__ __forceinline__ double dec(U k) { return __longlong_as_double((long long)(k ^ ((k >> 63) ? 0x8000000000000000ull : ~0ull))); }
};
template <class T>
__global__ __launch_bounds__(256) void vec_global_fill_kernel(T* __restrict__ x, int64_t m, int stat) {
    __shared__ double red[256];
    __shared__ unsigned int hist[256];
    __shared__ unsigned long long sel[2];              // [0] = digit chosen, [1] = rank left inside it
    T val;
    if (stat == 1) {
        typedef typename OrdKey<T>::U U;
        U prefix = 0, mask = 0;
        unsigned long long kth = (unsigned long long)(m / 2);      // 0-based rank of the wanted element
        for (int shift = (int)sizeof(U) * 8 - 8; shift >= 0; shift -= 8) {
            hist[threadIdx.x] = 0;
            __syncthreads();
            for (int64_t i = threadIdx.x; i < m; i += 256) {
                const U key = OrdKey<T>::enc(x[i]);
                if ((key & mask) == prefix) atomicAdd(&hist[(unsigned)((key >> shift) & 0xff)], 1u);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                unsigned long long run = 0;
                int dgt = 0;
                for (; dgt < 255; ++dgt) {
                    if (run + hist[dgt] > kth) break;
                    run += hist[dgt];
                }
                sel[0] = (unsigned long long)dgt;
                sel[1] = kth - run;
            }
            __syncthreads();
            prefix |= (U)sel[0] << shift;
            mask |= (U)0xff << shift;
            kth = sel[1];
            __syncthreads();
        }
        val = OrdKey<T>::dec(prefix);
    } else {
        double acc = 0.0;
        for (int64_t i = threadIdx.x; i < m; i += 256) acc += static_cast<double>(x[i]);
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
        val = static_cast<T>(red[0] / static_cast<double>(m));
    }
    __syncthreads();
    for (int64_t i = threadIdx.x; i < m; i += 256) x[i] = val;
}

// NB negative log-likelihood over the NONZEROS of A (explicit_loss.hpp:53-77), per-row theta, fp64 partials.
// One wavefront per column, ONE LANE PER NONZERO: each lane gathers its row of W_T with 16-byte loads, forms the
// prediction as k in-lane fmas against the column of H (broadcast from LDS) and evaluates the two lgamma / two log terms
// of its own nonzero -- 64 likelihood terms per wave instruction stream instead of one.
template <class T>
__global__ __launch_bounds__(256) void nb_loss_lane_kernel(
    const int* __restrict__ colptr, const int* __restrict__ rowidx, const T* __restrict__ vals, int64_t ncols,
    const T* __restrict__ W_T, const T* __restrict__ d, const T* __restrict__ H, const T* __restrict__ theta_row, int k,
    int vec_ok, int loss_type, double power, double robust, double* __restrict__ partial) {
    constexpr int VEC = 16 / sizeof(T);
    typedef typename VecT<T, VEC>::type V;
    __shared__ double sh[4];
    __shared__ T hs[4][128];
    __shared__ T dsh[128];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * 4 + wave;
    if (threadIdx.x < 128) dsh[threadIdx.x] = (int)threadIdx.x < k ? d[threadIdx.x] : T(0);
    hs[wave][lane] = (j < ncols && lane < k) ? H[j * (int64_t)k + lane] : T(0);
    hs[wave][lane + 64] = (j < ncols && lane + 64 < k) ? H[j * (int64_t)k + lane + 64] : T(0);
    __syncthreads();
    double acc = 0.0;
    if (j < ncols) {
        const int as = colptr[j], ae = colptr[j + 1];
        for (int t = as + lane; t < ae; t += 64) {
            const int row = rowidx[t];
            const T* wr = W_T + (int64_t)row * k;
            T pred = T(0);
            if (vec_ok) {
                for (int c = 0; c < k; c += VEC) {
                    const V v = *reinterpret_cast<const V*>(wr + c);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) pred = tfma(v[e] * dsh[c + e], hs[wave][c + e], pred);
                }
            } else {
                for (int c = 0; c < k; ++c) pred = tfma(wr[c] * dsh[c], hs[wave][c], pred);
            }
            const double y = static_cast<double>(vals[t]);
            double mu = static_cast<double>(pred);
            mu = mu > 1e-10 ? mu : 1e-10;
            const double th = static_cast<double>(theta_row ? theta_row[row] : T(0));
            double nll;
            if (robust > 0.0) {                    // math/loss.hpp:549-607  compute_robust_loss, in Scalar arithmetic
                const T muS = pred > T(1e-10) ? pred : T(1e-10);
                const T resid = vals[t] - muS;
                const T thS = theta_row ? theta_row[row] : T(0);
                T var_mu;
                if (loss_type == 4 || loss_type == 3) var_mu = muS;
                else if (loss_type == 5) { const T r = thS > T(1e-10) ? thS : T(1e-10); var_mu = muS + muS * muS / r; }
                else if (loss_type == 6) var_mu = muS * muS;
                else if (loss_type == 7) var_mu = muS * muS * muS;
                else if (loss_type == 8) var_mu = static_cast<T>(pow(static_cast<double>(muS), power));
                else var_mu = T(1);
                const T sd = sqrt(var_mu > T(1e-20) ? var_mu : T(1e-20));
                const T pr = resid / sd;
                const T apr = tabs(pr);
                const T dl = static_cast<T>(robust);
                const T rho = apr <= dl ? T(0.5) * pr * pr : dl * apr - T(0.5) * dl * dl;
                nll = static_cast<double>(rho);
            } else if (loss_type == 4) {           // math/loss.hpp:382-398  loss_contribution_gp
                const double opt = 1.0 + th;
                nll = -log(mu / opt);
                if (y >= 1.0) {
                    double inner = (mu + th * y) / opt;
                    inner = inner > 1e-10 ? inner : 1e-10;
                    nll -= (y - 1.0) * log(inner);
                }
                nll += (mu + th * y) / opt;
            } else if (loss_type >= 6) {           // math/loss.hpp:439-505  Gamma / inverse Gaussian / Tweedie deviance terms
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
            } else {                               // math/loss.hpp:415-426  loss_contribution_nb
                const double r = th > 1e-10 ? th : 1e-10;
                nll = -lgamma(y + r) + lgamma(r) - r * log(r / (r + mu)) - y * log(mu / (r + mu));
            }
            acc += static_cast<double>(static_cast<T>(nll));      // the reference casts each term to Scalar
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) sh[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// Wave-per-nonzero form (any k <= 64 layout; kept for reference and as the fallback).
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
        const bool fok2 = lane + 64 < k;
        const T hv = fok ? H[j * (int64_t)k + lane] : T(0);
        const T hv2 = fok2 ? H[j * (int64_t)k + lane + 64] : T(0);
        for (int t = colptr[j]; t < colptr[j + 1]; ++t) {
            const int row = rowidx[t];
            const T wd = fok ? W_T[(int64_t)row * k + lane] * d[lane] : T(0);
            const T wd2 = fok2 ? W_T[(int64_t)row * k + lane + 64] * d[lane + 64] : T(0);
            const T pred = wave_sum(wd * hv + wd2 * hv2);
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
