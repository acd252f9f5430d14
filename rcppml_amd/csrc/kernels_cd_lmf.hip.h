// ============================================================================
// kernels_cd_lmf.hip.h -- coordinate-descent NNLS, "lane = column" layout with the rank-1 residual updates on the matrix
// cores as 4x4 outer-product blocks (v_mfma_f32_4x4x1_16B_f32), persistent waves with column refill.  fp32, k <= 64,
// non-negativity only (what every NMF half-update uses).
//
// Reference routine: primitives/cpu/nnls_batch.hpp:70-132 (cd_nnls_col_fixed), prologue fused_nnls.hpp:116-123.
//
// Layout.  v_mfma_f32_4x4x1_16B computes 16 independent 4x4 outer products: block = lane>>2, A[block][r] and B[block][c]
// come from lane 4*block + (r|c), D[block][r][c] lives in register r of lane 4*block + c (probed: tools/probe/
// mfma4x4_probe.hip).  With the A-block broadcast (CBSZ/ABID: ONE block of A serves a whole group of blocks) the
// instruction becomes  acc[t][r](lane) += A[lane 4t + r] * B[lane]  for r = 0..3: a 4-row slice of a rank-1 update in which
// every lane keeps ITS OWN B value.  So:
//   * a lane owns a column (LG = 1) or 1/LG of one (LG lane groups of 64/LG lanes; group g holds the coordinates
//     i = g (mod LG)), its residuals in NTL = KP/(4 LG) accumulator tiles and its iterate in as many registers;
//   * G(:, i) enters as ONE register, lane-distributed (lane <-> row), read from LDS together with 1/G_ii;
//   * the step of coordinate i is computed where its residual lives and IS the B operand -- no cross-lane traffic at all
//     for LG = 1; for LG > 1 the B lane-group broadcast (BLGP) hands group g's steps to the other groups inside the MFMA.
// Per coordinate: ~4 VALU + 16/LG MFMAs for 64/LG columns.  The 32x32x2 kernel (kernels_cd_mfma.hip.h) spends ~15 VALU + a
// permlane swap per coordinate PAIR and 32 columns and evaluates every step twice.
// Arithmetic per residual element: the reference's single-rounded fma chain in coordinate order (f32 MFMA == fmaf chain);
// the tolerance sum runs in coordinate order per lane group (LG = 1: exactly the reference's order) with v_rcp_f32.
//
// Persistent waves.  The launch holds as many waves as the device keeps resident and every wave owns a FIXED, interleaved share
// of the work order: band b (= NW consecutive positions of `order`, NW = waves in the launch) gives wave w its position
// b NW + w (odd bands reversed, so that no wave collects the long end of every band).  Lane c starts with the wave's column of
// band c; a lane (group) that finishes its column stores it and takes the wave's column of the next unused band.  `order`
// lists the columns longest first, so the early-free lanes get the longest remaining columns (LPT inside a wave) and all waves
// hold statistically identical job sets -- no global ticket counter, no atomics, the same schedule on every run.  The column
// ids of a wave's next 64 bands are fetched once (one lane each) and handed out with a lane permute.
// A new column enters in CORRECTION mode for one sweep: its "step" is its warm iterate, so the regular MFMA stream of that
// sweep computes b - G x (fused_nnls.hpp:121-123) for it while the other lanes do CD steps.
// ============================================================================
#pragma once
#include <type_traits>
#include "kernels.hip.h"

namespace rk {

typedef float lmf_f32x4 __attribute__((ext_vector_type(4)));

template <int I, int N, class F> __device__ __forceinline__ void lmf_static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); lmf_static_for<I + 1, N>(f); }
}

template <int KP, int LG> struct LmfGeom {
    static constexpr int CW = 64 / LG;          // columns per wave = lanes per lane group
    static constexpr int RPG = KP / LG;         // coordinates held by one lane group
    static constexpr int NTL = RPG / 4;         // accumulator tiles per lane = MFMAs per coordinate
    static constexpr int CBSZ = LG == 1 ? 4 : (LG == 2 ? 3 : 2);   // A-block broadcast over 16 / 8 / 4 blocks = one lane group
    static_assert(KP == 32 || KP == 64, "KP");
    static_assert(LG == 1 || LG == 2 || LG == 4, "LG");
    // coordinate i sits in lane group i % LG, slot s = i / LG of that group: tile s / 4, register s % 4; the A operand lane
    // that carries row i is CW * (i % LG) + i / LG
    __host__ __device__ static constexpr int row_of_lane(int lane) {          // logical row carried by an A-operand lane
        return (lane % CW) < RPG ? (lane % CW) * LG + lane / CW : -1;
    }
    // B lane-group pattern that hands the steps of lane group g to all groups (probe: 1 = lanes 0-31 -> 32-63, 2 = the
    // reverse, 4 + g = 16-lane group g to all)
    __host__ __device__ static constexpr int blgp(int g) { return LG == 1 ? 0 : (LG == 2 ? (g == 0 ? 1 : 2) : 4 + g); }
};

// img[i * 64 + lane] = { -G(row_of_lane(lane), i), 1 / G(i, i) (0 if G(i,i) <= 0) }, identity padding beyond k.
template <int KP, int LG>
static __global__ void cd_lmf_prep_kernel(const float* __restrict__ G, int k, float2* __restrict__ img) {
    typedef LmfGeom<KP, LG> Ge;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= KP * 64) return;
    const int i = e >> 6, lane = e & 63;
    auto gp = [&](int col, int row) { return (row < k && col < k) ? G[(int64_t)col * k + row] : (row == col ? 1.f : 0.f); };
    const int row = Ge::row_of_lane(lane);
    const float gd = gp(i, i);
    float2 v;
    v.x = row >= 0 ? -gp(i, row) : 0.f;
    v.y = gd > 0.f ? 1.f / gd : 0.f;
    img[e] = v;
}

template <int CBSZ, int ABID, int BLGP>
__device__ __forceinline__ lmf_f32x4 lmf_mfma(float a, float b, lmf_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, CBSZ, ABID, BLGP);
}

template <int KP, int LG, bool COUNT>
__global__ __launch_bounds__(256) void cd_lmf_kernel(const float2* __restrict__ img_g, const float* __restrict__ B,
                                                      float* __restrict__ X, int k, int64_t ncols, float l1_pre, int warm,
                                                      int zero_init, int maxit, float tol, float ub_post,
                                                      int* __restrict__ sweeps, const int* __restrict__ order,
                                                      unsigned long long* __restrict__ stats) {
    typedef LmfGeom<KP, LG> Ge;
    constexpr int CW = Ge::CW, NTL = Ge::NTL, CBSZ = Ge::CBSZ;
    constexpr int PD = LG == 1 ? 2 : 4;                     // LDS operands are requested PD coordinates ahead
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2* img = reinterpret_cast<float2*>(smem_raw);      // KP x 64 x {-G, 1/G_ii}
    for (int e = threadIdx.x; e < KP * 64; e += blockDim.x) img[e] = img_g[e];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane / CW, cl = lane % CW;
    const int64_t NW = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t wid = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    // position of this wave's column of band b in the work order; -1 behind the end
    // When the launch holds a slot for every column (no refill can happen) a wave takes CW CONSECUTIVE positions of the work
    // order instead -- columns of similar sweep counts, as the tile kernels do: with the interleaved bands every wave would hold
    // one of the longest columns and run to its end with most lanes idle.
    const bool tile_mode = NW * CW >= ncols;
    auto column_of_band = [&](int64_t band) -> int64_t {
        const int64_t t = tile_mode ? (band < CW ? wid * CW + band : ncols) : band * NW + ((band & 1) ? NW - 1 - wid : wid);
        return t < ncols ? (order ? (int64_t)order[t] : t) : -1;
    };
    const bool vec_ok = LG == 1 && (k & 3) == 0 && ((reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(X)) & 15) == 0;

    lmf_f32x4 acc[NTL];
    float xn[NTL][4];                     // the iterate, NEGATED: the step is max(diff, -x) and v_max takes no free negation
    int64_t j = -1;
    int nsweep = 0;
    bool cm = false;                      // correction mode: this sweep applies b -= G x for a freshly loaded column

    // (re)load: column -> residual start b - l1, iterate.  Every lane group of a column executes it with the same column id
    // all loads of a refill are issued before any of them is consumed (the empty asm statements pin the loaded values, or LLVM
    // sinks every load into the select that uses it and a divergent refill becomes 2 KP/LG serial round trips)
    auto load_column = [&](int64_t col) {
        const bool has = col >= 0;
        const int64_t jj = has ? col : 0;
        j = col;
        nsweep = 0;
        cm = has & (warm != 0) & (zero_init == 0);
        const float* bj = B + jj * (int64_t)k;
        const float* xj = X + jj * (int64_t)k;
        float bv[NTL][4], xv[NTL][4];
        if (vec_ok) {
#pragma unroll
            for (int t = 0; t < NTL; ++t) {
                const int off = 4 * t < k ? 4 * t : 0;
                const float4 b4 = *reinterpret_cast<const float4*>(bj + off);
                const float4 x4 = *reinterpret_cast<const float4*>(xj + off);
                bv[t][0] = b4.x; bv[t][1] = b4.y; bv[t][2] = b4.z; bv[t][3] = b4.w;
                xv[t][0] = x4.x; xv[t][1] = x4.y; xv[t][2] = x4.z; xv[t][3] = x4.w;
            }
        } else {
#pragma unroll
            for (int t = 0; t < NTL; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = (4 * t + r) * LG + g;
                    const int rr = row < k ? row : 0;
                    bv[t][r] = bj[rr];
                    xv[t][r] = xj[rr];
                }
        }
#pragma unroll
        for (int t = 0; t < NTL; ++t)
            asm volatile("" : "+v"(bv[t][0]), "+v"(bv[t][1]), "+v"(bv[t][2]), "+v"(bv[t][3]),
                              "+v"(xv[t][0]), "+v"(xv[t][1]), "+v"(xv[t][2]), "+v"(xv[t][3]));
#pragma unroll
        for (int t = 0; t < NTL; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = (4 * t + r) * LG + g;
                const bool ok = has & (row < k);
                acc[t][r] = ok ? bv[t][r] - l1_pre : 0.f;           // b - 0 is exact
                xn[t][r] = (ok & (zero_init == 0)) ? -xv[t][r] : 0.f;
            }
    };
    // column ids of this wave's bands CW + lane (the next 64 refills), fetched once; re-fetched if a wave ever needs more
    int64_t pref_base = CW;
    int64_t pref_col = column_of_band(pref_base + lane);
    int64_t next_band = CW;
    load_column(column_of_band(cl));

    const bool check = tol > 0.f;
    const float inv_k = 1.f / static_cast<float>(k);
    unsigned long long my_colsweeps = 0, my_cols = 0, wave_sweeps = 0, noop_steps = 0;
    [[maybe_unused]] unsigned long long col_noops = 0, col_steps = 0;
    bool ing[LG];
#pragma unroll
    for (int q = 0; q < LG; ++q) ing[q] = g == q;

    // LDS operands of the first PD coordinates; coordinate i then requests coordinate i + PD (the sweep's last PD
    // coordinates request the next sweep's first ones)
    float2 ring[PD];
#pragma unroll
    for (int p = 0; p < PD; ++p) ring[p] = img[p * 64 + lane];

    while (__any(j >= 0)) {
        float tsum = 0.f;
        float areg = 0.f;
        const float keep = cm ? 0.f : 1.f;
        lmf_static_for<0, KP>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int gi = i % LG, s = i / LG, t = s / 4, r = s % 4;
            constexpr int tn = (((i + 1) % KP) / LG) / 4;         // tile that holds the next coordinate's residual
            const float2 gq = ring[i % PD];
            ring[i % PD] = img[((i + PD) % KP) * 64 + lane];
            const float xo = xn[t][r];                 // = -x_i
            const float diff = acc[t][r] * gq.y;
            // reference: nv = x + diff; nv < 0 ? (a = -x, x = 0) : (a = diff, x = nv)  ==  a = max(diff, -x), x = x + a
            // (see cd_scalar_step in kernels_cd_mfma.hip.h); a dead diagonal arrives as 1/G_ii = 0: a = 0
            float a = __builtin_fmaxf(diff, xo);
            a = cm ? -xo : a;
            if constexpr (LG == 1) areg = a;
            else areg = ing[gi] ? a : areg;            // lanes of the other groups hold other coordinates' steps
            if constexpr (COUNT) {
                const bool mine = (LG == 1 || ing[gi]) && j >= 0;
                if (!__any(mine && a != 0.f)) noop_steps += 1;
                if (!cm && i < k) {      // per (column, coordinate): the steps the reference skips (nnls_batch.hpp:102,106,109 `continue`)
                    col_steps += __popcll(__ballot(mine));
                    col_noops += __popcll(__ballot(mine && a == 0.f));
                }
            }
            lmf_static_for<0, NTL>([&](auto uc) {
                constexpr int tt = (tn + decltype(uc)::value) % NTL;
                acc[tt] = lmf_mfma<CBSZ, tt, Ge::blgp(gi)>(gq.x, areg, acc[tt]);
            });
            if constexpr (gi == LG - 1) {
                // all LG coordinates that share register (t, r) are done: iterate update and tolerance terms for all of them
                const float xnew = __builtin_fmaf(-areg, keep, xo);          // -(x + a)
                xn[t][r] = xnew;
                tsum = __builtin_fmaf(tabs(areg), __builtin_amdgcn_rcpf(tabs(xnew) + 1e-15f), tsum);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        wave_sweeps += 1;
        float tot = tsum;
        if constexpr (LG == 2) tot += __shfl_xor(tot, 32, 64);
        if constexpr (LG == 4) { tot += __shfl_xor(tot, 16, 64); tot += __shfl_xor(tot, 32, 64); }
        const bool has = j >= 0;
        const bool was_cm = cm;
        cm = false;
        nsweep += (has & !was_cm) ? 1 : 0;
        // bitwise on purpose: behind a short-circuit branch LLVM sinks all KP tolerance terms of the sweep to here and keeps
        // every step of the sweep alive for them (+64 registers)
        const bool fin = has & !was_cm & ((check & (tot * inv_k < tol)) | (nsweep >= maxit));
        if (__any(fin)) {
            if (fin) {
                float* xj = X + j * (int64_t)k;
                if (vec_ok) {
#pragma unroll
                    for (int t = 0; t < NTL; ++t)
                        if (4 * t < k) {
                            float4 v = make_float4(0.f - xn[t][0], 0.f - xn[t][1], 0.f - xn[t][2], 0.f - xn[t][3]);   // 0 - (+0) = +0: zeros leave as +0
                            if (ub_post > 0.f) {
                                v.x = v.x < ub_post ? v.x : ub_post; v.y = v.y < ub_post ? v.y : ub_post;
                                v.z = v.z < ub_post ? v.z : ub_post; v.w = v.w < ub_post ? v.w : ub_post;
                            }
                            *reinterpret_cast<float4*>(xj + 4 * t) = v;
                        }
                } else {
#pragma unroll
                    for (int t = 0; t < NTL; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = (4 * t + r) * LG + g;
                            if (row < k) {
                                float val = 0.f - xn[t][r];
                                if (ub_post > 0.f) val = val < ub_post ? val : ub_post;
                                xj[row] = val;
                            }
                        }
                }
                if (g == 0) {
                    if (sweeps) sweeps[j] = nsweep;
                    my_colsweeps += (unsigned long long)nsweep;
                    my_cols += 1;
                }
            }
            // lane group 0 numbers its finished columns, column q of the event takes the wave's band next_band + q; the other
            // lane groups read the id of their column from group 0
            const unsigned long long m = __ballot(fin && g == 0);
            const int n = __popcll(m);
            if (next_band + n > pref_base + 64) {              // (uniform) more than 64 refills in this wave: next batch of ids
                pref_base = next_band;
                pref_col = column_of_band(pref_base + lane);
            }
            const int idx = (int)(next_band - pref_base) + __popcll(m & ((1ull << lane) - 1ull));
            int64_t col = __shfl(pref_col, idx & 63, 64);
            if constexpr (LG > 1) col = __shfl(col, cl, 64);
            next_band += n;
            if (fin) load_column(col);
        }
    }
    if (stats) {
        unsigned long long v = my_colsweeps, c = my_cols;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { v += __shfl_xor(v, off, 64); c += __shfl_xor(c, off, 64); }
        if (lane == 0) {
            if (c > 0) { atomicAdd(stats, v); atomicAdd(stats + 1, c); }
            atomicAdd(stats + 2, wave_sweeps * (unsigned long long)CW);     // slot-sweeps executed (idle + correction included)
            if constexpr (COUNT) { atomicAdd(stats + 3, noop_steps); atomicAdd(stats + 6, col_noops); atomicAdd(stats + 7, col_steps); }
        }
    }
}

}  // namespace rk
