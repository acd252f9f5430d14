// ============================================================================
// kernels_cd_mfma.hip.h -- coordinate-descent NNLS with the rank-1 residual updates on the MATRIX cores (fp32).
//
// Reference routine: primitives/cpu/nnls_batch.hpp:70-132 (cd_nnls_col_fixed), prologue fused_nnls.hpp:116-123.
//
// Idea.  A wave solves 64 columns.  Their residuals B (KP rows x 64 columns) live in MFMA accumulator tiles
// (v_mfma_f32_32x32x2_f32: RT row tiles x 2 column tiles of 32x32, 16 VGPRs each).  One CD coordinate i changes all
// residuals by the outer product  B -= G(:,i) a_i^T  (a_i = the 64 columns' steps): exactly the K-slot of an MFMA.
// Two consecutive coordinates fill the two K-slots of 32x32x2, so ONE instruction per tile applies coordinates i and
// i+1 (accumulation order inside the instruction is slot 0 then slot 1 = the reference's order, each a single-rounded
// fma, so results equal the VALU kernels').  What this buys on MI355X:
//   * G enters LANE-DISTRIBUTED (lane l supplies G[row l&31][coordinate i + (l>>5)]): one conflict-free ds_read_b32 per
//     row tile and coordinate pair, instead of broadcasting every G element to all lanes (which saturates the LDS);
//   * the KP fmas per column and coordinate leave the VALU, which keeps only the ~12-op scalar step of the reference;
//     matrix and vector pipes run concurrently (two waves per SIMD keep the matrix pipe busy).
// Data placement trick: the logical row <-> accumulator position map is chosen so that logical rows 2q and 2q+1 of a
// tile sit in the SAME accumulator register q, in the low (lanes 0-31) and high (lanes 32-63) half respectively
// (C/D map of the 32x32 MFMA: row = (v&3) + 8(v>>2) + 4(lane>>5); we store logical row 32rt + 2v + (lane>>5) there
// and permute the rows of G to match).  Hence for the coordinate pair (i, i+1) = (32rt+2q, 32rt+2q+1):
//   low half  holds b_i   for its two columns (column tiles 0/1) in acc[rt][ct][q]  -> computes a_i,
//   high half holds b_i+1 for the same columns in the same register               -> computes a_i+1 after the lazy
//   Gauss-Seidel correction  b_i+1 -= G(i+1,i) a_i  (a_i fetched from the low half with one v_permlane32_swap),
// and the resulting register {a_i | a_i+1} IS the MFMA B operand (B[kk = lane>>5][col = lane&31]) with no data movement.
// The iterate x uses the same register layout.  Finished columns are frozen (step forced to 0).
// The per-column tolerance sum is accumulated per half (even / odd coordinates) and added at the end of the sweep --
// the only reassociation relative to the sequential reference.
// ============================================================================
#pragma once
#include "kernels.hip.h"

namespace rk {

// Gq[((i/2)*RT + rt)*64 + (i&1)*32 + r'] = -G(lrow, i), lrow = 32*rt + 2*v + h, h = (r'>>2)&1, v = (r'&3) + 4*(r'>>3).
// tab[c] (c = coordinate; pair i = c & ~1 is served by lane half h = c & 1) = { 1/G(c,c) (0 if G(c,c) <= 0),
//   coupling inside the pair: h ? G(i+1, i) : 0 }.
// Formed by every workgroup of cd_mfma_kernel from the k x k Gram while it fills its LDS (identity padding to KP and 1/G_ii on the
// fly): pair-major layout -- 64 consecutive floats = one MFMA A operand (both halves of a wave) of pair i/2 and row tile rt, so
// every LDS read of the sweep is  <one base register> + <compile-time multiple of 256 bytes>; the high half (odd coordinate)
// reads its Gauss-Seidel coupling from tab, the low half reads 0, so ONE evaluation of the second step serves both halves.

struct CdStepOut { float a, nx; };

// max(a, b) as v_med3_f32(a, b, +inf): llvm.maxnum on a loop-carried operand costs a canonicalising v_max(b, b) in front of
// the real one (IEEE mode: the instruction must not see a signalling NaN); the target intrinsic does not.  Same value for
// every non-NaN input.  `inf` must be a RUN-TIME +infinity (the kernels' `hi` when there is no upper bound): with a literal
// LLVM folds the median back into maxnum.
__device__ __forceinline__ float cd_max_nc(float a, float b, float inf) { return __builtin_amdgcn_fmed3f(a, b, inf); }

// The reference's scalar step.  SIMPLE = non-negativity only (no upper bound, no in-CD L1/L2): what every NMF
// half-update uses; the general form serves nnls()/predict() (L1 inside CD, box constraints, nonneg = FALSE).
template <bool SIMPLE>
__device__ __forceinline__ CdStepOut cd_scalar_step(float b, float xo, float ginv, bool active, float l1_cd, float l2_cd,
                                                    float lo, float hi) {
    CdStepOut o;
    if constexpr (SIMPLE) {
        // reference: nv = xo + diff; if (nv < 0) { a = -xo; x = 0 } else { a = diff; x = nv }.  nv < 0 <=> diff < -xo (the
        // sum of two floats keeps its sign through rounding: tiny sums are exact), so a = max(diff, -xo) and
        // x = max(nv, 0) -- the same values with a two-instruction dependent chain (v_mul, v_max) in front of the MFMA
        // instead of four (v_mul, v_add, v_cmp, v_cndmask): the sweep is a serial chain of these steps.
        // `active` and the reference's `if (g_diag <= 0) continue;` arrive folded into ginv (= 0): then diff = 0,
        // a = max(0, -xo) = 0 and nx = xo for the non-negative iterates of this mode
        const float diff = b * ginv;
        o.a = __builtin_fmaxf(diff, -xo);
        o.nx = __builtin_fmaxf(xo + diff, 0.f);
    } else {
        float diff = b * ginv;
        diff -= l1_cd;                        // reference: `if (L1 != 0) diff -= L1` (subtracting 0 is exact)
        diff = __builtin_fmaf(l2_cd, xo, diff);
        const float nv = xo + diff;
        const bool neg = nv < lo, up = nv > hi;
        float nx = neg ? lo : (up ? hi : nv);
        float a = neg ? lo - xo : (up ? hi - xo : diff);
        const bool on = active && (ginv > 0.f);
        a = on ? a : 0.f;
        nx = on ? nx : xo;
        o.a = a;
        o.nx = nx;
    }
    return o;
}

// LDS bytes of the operand image (launch code: solve_cd_impl.hip.h).  RT <= 2 (k <= 64): PACKED -- one float4 per lane and
// coordinate pair {A operand of row tile 0, of row tile 1, 1/G_cc, pair coupling}, i.e. ONE ds_read_b128 per pair instead of a
// b64 (table) + a b32 pair (operands); RT > 2: the pair-major float image + the float2 table described above.
__host__ __device__ constexpr size_t cd_mfma_lds_bytes(int RT) {
    return RT <= 2 ? (size_t)(16 * RT) * 64 * sizeof(float4) : ((size_t)(32 * RT) * (32 * RT) + 2 * (32 * RT)) * sizeof(float);
}

template <int RT, int CT, bool SIMPLE>   // KP = 32*RT rows (k <= KP), 32*CT columns per wave, 4 waves per block share G
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((CT == 1 && RT <= 2) ? 4 : 2, 8))) void cd_mfma_kernel(const float* __restrict__ G /* k x k, as rcppml_hip_gram wrote it */,
                                                       const float* __restrict__ B,
                                                       float* __restrict__ X, int k, int64_t ncols, float l1_pre,
                                                       int warm, int zero_init, float l1_cd, float l2_cd, int nonneg,
                                                       int maxit, float tol, float ub_cd, float ub_post,
                                                       int* __restrict__ sweeps, const int* __restrict__ order,
                                                       unsigned long long* __restrict__ stats) {
    constexpr int KP = 32 * RT;
    constexpr bool PACK = RT <= 2;
    // SIMPLE keeps the iterate NEGATED (xs = -x): the step is max(diff, -x) and v_max takes no free negation, the update is
    // min(fma(-b, 1/G, -x), 0) with the negation folded into the fma's modifiers -- same magnitudes bit for bit, one VALU less
    // per pair
    constexpr float XSIGN = SIMPLE ? -1.f : 1.f;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* Gs = reinterpret_cast<float*>(smem_raw);      // !PACK: KP*KP, pair-major (layout comment above)
    float2* tab_s = reinterpret_cast<float2*>(Gs + KP * KP);   // !PACK: KP x {1/G_cc, pair coupling}
    float4* img = reinterpret_cast<float4*>(smem_raw);   // PACK: (KP/2) x 64
    // The permuted operand image and the per-coordinate table are formed HERE from the k x k Gram (until round 3 a launch of its own: every workgroup reads the same 4 k^2 bytes either way, and the solve is two launches shorter per
    // iteration).  LDS element e <-> (pair, row tile, half, r): the layout comment above.
    {
        auto gp = [&](int col, int row) { return (row < k && col < k) ? G[(int64_t)col * k + row] : (row == col ? 1.f : 0.f); };
        if constexpr (PACK) {
            for (int e = threadIdx.x; e < (KP / 2) * 64; e += blockDim.x) {
                const int pr = e >> 6, w6 = e & 63;
                const int i = 2 * pr + (w6 >> 5), r = w6 & 31;
                const int h = (r >> 2) & 1, v = (r & 3) + 4 * (r >> 3);
                const float gd = gp(i, i);
                float4 t;
                t.x = -gp(i, 2 * v + h);
                t.y = RT == 2 ? -gp(i, 32 + 2 * v + h) : 0.f;
                t.z = gd > 0.f ? 1.f / gd : 0.f;
                t.w = (i & 1) ? gp(i - 1, i) : 0.f;
                img[e] = t;
            }
        } else {
            for (int e = threadIdx.x; e < KP * KP; e += blockDim.x) {
                const int pr = e >> 6, w6 = e & 63;
                const int i = 2 * (pr / RT) + (w6 >> 5), rt = pr % RT, r = w6 & 31;
                const int h = (r >> 2) & 1, v = (r & 3) + 4 * (r >> 3);
                Gs[e] = -gp(i, 32 * rt + 2 * v + h);
            }
            for (int i = threadIdx.x; i < KP; i += blockDim.x) {
                float2 t;
                const float gd = gp(i, i);
                t.x = gd > 0.f ? 1.f / gd : 0.f;
                t.y = (i & 1) ? gp(i - 1, i) : 0.f;
                tab_s[i] = t;
            }
        }
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, cl = lane & 31;
    const int64_t base = ((int64_t)blockIdx.x * (blockDim.x >> 6) + wave) * (32 * CT);
    if (base >= ncols) return;
    int64_t j[CT];
    bool inb[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int64_t slot = base + ct * 32 + cl;
        inb[ct] = slot < ncols;
        j[ct] = (inb[ct] && order) ? order[slot] : slot;
    }
    f32x16 acc[RT][CT];
    float xr[RT][CT][16];          // XSIGN * x
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const float* bj = B + j[ct] * (int64_t)k;
            const float* xj = X + j[ct] * (int64_t)k;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = 32 * rt + 2 * v + half;
                const bool ok = inb[ct] && row < k;
                float bv = ok ? bj[row] : 0.f;
                if (ok && l1_pre != 0.f) bv -= l1_pre;
                acc[rt][ct][v] = bv;
                xr[rt][ct][v] = (ok && !zero_init) ? XSIGN * xj[row] : 0.f;
            }
        }
    // operands of pair p for this lane: A values of the RT row tiles, {1/G_cc, coupling} of the lane half's coordinate
    struct PairOps { float av[RT]; float ginv, coup; };
    auto load_pair = [&](int p) {
        PairOps o;
        if constexpr (PACK) {
            const float4 t = img[p * 64 + lane];
            o.av[0] = t.x;
            if constexpr (RT == 2) o.av[1] = t.y;
            o.ginv = t.z; o.coup = t.w;
        } else {
#pragma unroll
            for (int rt2 = 0; rt2 < RT; ++rt2) o.av[rt2] = Gs[(p * RT + rt2) * 64 + lane];
            const float2 t = tab_s[2 * p + half];
            o.ginv = t.x; o.coup = t.y;
        }
        return o;
    };
    if (warm) {   // B -= G X (fused_nnls.hpp:121-123): the same MFMA stream with x in place of the steps
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int i = 32 * rt + 2 * q;
                const PairOps po = load_pair(i >> 1);
#pragma unroll
                for (int rt2 = 0; rt2 < RT; ++rt2) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        acc[rt2][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(po.av[rt2], XSIGN * xr[rt][ct][q], acc[rt2][ct], 0, 0, 0);
                }
            }
    }
    const float lo = nonneg ? 0.f : -INFINITY;
    const float hi = ub_cd > 0.f ? ub_cd : INFINITY;
    const bool check = tol > 0.f;
    const float inv_k = 1.f / static_cast<float>(k);
    bool active[CT];
    int nsweep[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) { active[ct] = inb[ct]; nsweep[ct] = 0; }
    // LDS operands of the first pair; every pair then requests the NEXT pair's operands before it starts computing,
    // so their latency hides behind the current pair (the last pair of a sweep prefetches pair 0 again)
    PairOps cur = load_pair(0);
    for (int it = 0; it < maxit; ++it) {
        bool any_active = false;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) any_active |= active[ct];
        if (!__any(any_active)) break;
        float tsum[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) { tsum[ct] = 0.f; nsweep[ct] += active[ct] ? 1 : 0; }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                // low half: coordinate i = 32rt + 2q, high half: coordinate i + 1
                constexpr int KPc = KP;
                const int inext = (32 * rt + 2 * q + 2) % KPc;
                const int rtn = (q == 15 ? rt + 1 : rt) % RT;   // row tile of the next pair
                const PairOps nxt = load_pair(inext >> 1);
                const float g_oe = cur.coup;
                float aval[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const float b = acc[rt][ct][q];
                    const float xo = xr[rt][ct][q];
                    if constexpr (SIMPLE) {
                        // reference: nv = x + diff; if (nv < 0) { a = -x; x = 0 } else { a = diff; x = nv }.  nv < 0 <=> diff < -x
                        // (the sum of two floats keeps its sign through rounding), so a = max(diff, -x), x = max(nv, 0): a
                        // two-instruction dependent chain in front of the MFMA.  Frozen columns hold b = 0 (zeroed when they
                        // converge, below): diff = 0, a = max(0, -x) = 0, x unchanged -- no per-pair select of 1/G_cc; a dead
                        // diagonal arrives as 1/G_cc = 0.  xo = -x here.
                        const float ginv = cur.ginv;
                        float ae_lo = cd_max_nc(b * ginv, xo, hi);
                        // its step, seen from the high half (same column, lane - 32).  v_permlane32_swap with vdst == src0
                        // exchanges the two halves of one register in place; the low half then holds a don't-care that
                        // meets g_oe = 0
                        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %0" : "+v"(ae_lo));
                        // high half: odd coordinate after the lazy Gauss-Seidel correction b -= G(i+1,i) a_i.  Low half:
                        // g_oe = 0, so this re-evaluates the even step on identical inputs -- the result register is
                        // {a_i | a_i+1} = the MFMA B operand, and no half-select is needed.
                        const float bo = __builtin_fmaf(-g_oe, ae_lo, b);
                        const float diff = bo * ginv;
                        const float a = cd_max_nc(diff, xo, hi);
                        const float nxn = __builtin_fminf(xo - diff, 0.f);      // -(max(x + diff, 0))
                        aval[ct] = a;
                        xr[rt][ct][q] = nxn;
                        // |a| / (|x_new| + 1e-15): one evaluation per pair and half (v_rcp_f32), nnls_batch.hpp:117-120
                        tsum[ct] = __builtin_fmaf(tabs(a), __builtin_amdgcn_rcpf(tabs(nxn) + 1e-15f), tsum[ct]);
                    } else {
                        const float ginv = cur.ginv;
                        const CdStepOut e = cd_scalar_step<SIMPLE>(b, xo, ginv, active[ct], l1_cd, l2_cd, lo, hi);
                        float ae_lo = e.a;
                        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %0" : "+v"(ae_lo));
                        const float bo = __builtin_fmaf(-g_oe, ae_lo, b);
                        const CdStepOut o = cd_scalar_step<SIMPLE>(bo, xo, ginv, active[ct], l1_cd, l2_cd, lo, hi);
                        aval[ct] = o.a;
                        xr[rt][ct][q] = o.nx;
                        tsum[ct] = __builtin_fmaf(tabs(o.a), __builtin_amdgcn_rcpf(tabs(o.nx) + 1e-15f), tsum[ct]);
                    }
                }
                // the row tile that holds the NEXT pair's residuals goes first, so its results are back first
#pragma unroll
                for (int s2 = 0; s2 < RT; ++s2) {
                    const int rt2 = (rtn + s2) % RT;
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        acc[rt2][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.av[rt2], aval[ct], acc[rt2][ct], 0, 0, 0);
                }
                cur = nxt;
                // one scheduling region per coordinate pair: without it hipcc hoists the LDS reads of many pairs and
                // spends > 380 registers on this fully unrolled sweep
                __builtin_amdgcn_sched_barrier(0);
            }
        // branch-free on purpose: behind an `if (check)` LLVM sinks all 64 tolerance terms of the sweep into the
        // branch and keeps every step and iterate of the sweep alive for it (+128 registers)
        bool froze = false;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const float tot = tsum[ct] + __shfl_xor(tsum[ct], 32, 64);       // even + odd coordinates
            const bool now = active[ct] && !(check && tot * inv_k < tol);
            froze |= active[ct] && !now;
            active[ct] = now;
        }
        if constexpr (SIMPLE) {
            // columns that converged in this sweep: their residuals are no longer needed -- zero them, and every later step of
            // the column is exactly 0 (see the step above)
            if (__any(froze)) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                        for (int v = 0; v < 16; ++v) acc[rt][ct][v] = active[ct] ? acc[rt][ct][v] : 0.f;
            }
        }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        if (!inb[ct]) continue;
        float* xj = X + j[ct] * (int64_t)k;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = 32 * rt + 2 * v + half;
                if (row < k) {
                    float val = SIMPLE ? 0.f - xr[rt][ct][v] : xr[rt][ct][v];      // 0 - (+-0) = +0: zeros leave as +0
                    if (ub_post > 0.f) val = val < ub_post ? val : ub_post;
                    xj[row] = val;
                }
            }
        if (sweeps && half == 0) sweeps[j[ct]] = nsweep[ct];
    }
    int nsw = 0, ncol = 0;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
        if (inb[ct] && half == 0) { nsw += nsweep[ct]; ncol += 1; }
    cd_stats_add(stats, nsw, ncol);
}

}  // namespace rk
