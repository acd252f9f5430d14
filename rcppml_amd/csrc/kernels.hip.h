// ============================================================================
// kernels.hip.h -- hand-written CDNA4 (gfx950) kernels for RcppML's ALS-NNLS NMF hot path.
// Written for 64-wide wavefronts, MI355X only (no CUDA/portable paths).
//
// Layout conventions: dense matrices column-major with rank k as leading dimension
// (W_T k x m, H k x n, G k x k, B k x c); CSC with int32 indices.
// Each kernel cites the reference routine it implements
// (paths relative to /root/reference/inst/include/FactorNet).
// ============================================================================
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rk {

// Order LDS traffic between lanes of ONE wavefront (no block barrier: waves diverge in trip counts).
#define RK_WAVE_SYNC()                                              \
    do {                                                            \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      \
        __builtin_amdgcn_wave_barrier();                            \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float tabs(float v) { return __builtin_fabsf(v); }    // free |x| source modifier
__device__ __forceinline__ double tabs(double v) { return __builtin_fabs(v); }
__device__ __forceinline__ float tfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double tfma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// value of lane i (wave-uniform i) through v_readlane_b32: the result lands in an SGPR, no LDS-crossbar round trip
// b / G_cc inside a coordinate-descent sweep: fp64 divides as the reference writes it (parity mode), fp32 multiplies by the
// reciprocal formed once per column (one rounding more; the same choice as EXACT = false in the MSE kernels) -- an IEEE fp32
// division is ~10 dependent instructions on the chain of every step
__device__ __forceinline__ float sweep_quotient(float b, float /*gd*/, float ginv) { return b * ginv; }
__device__ __forceinline__ double sweep_quotient(double b, double gd, double /*ginv*/) { return b / gd; }
// b / gd inside the fp64 ("parity mode") sweeps that keep 1 / G_cc per column: the quotient from that reciprocal plus ONE correction,
//   q0 = b ginv,  r = b - q0 gd (a single fma: exact),  q = q0 + r ginv
// -- with ginv = RN(1 / gd) this is RN(b / gd), the reference's `b[i] / g_diag` (nnls_batch.hpp:100), bar over/underflow of the
// intermediates; two fmas on the chain of a coordinate instead of the ~15 dependent instructions of an IEEE fp64 division (until round 4
// these sweeps took q0: one rounding more).  Checked against the division operator on 2e8 random operand pairs in
// the CPU test test_corrected_reciprocal_quotient_equals_division under tests/ (host fma = device fma: both IEEE).  gd <= 0 arrives as
// ginv = 0: q = 0.  fp32 keeps b ginv (the reference's own fp32 arithmetic is what the fp32 mode is compared with at 1e-4..1e-6).
__device__ __forceinline__ double cd_quotient(double b, double gd, double ginv) {
    const double q0 = b * ginv;
    return __builtin_fma(__builtin_fma(-q0, gd, b), ginv, q0);
}
__device__ __forceinline__ float cd_quotient(float b, float /*gd*/, float ginv) { return b * ginv; }
// the step's quotient with the L1 term: fp32 one fma (b ginv - L1), fp64 the reference's two roundings (b / gd, then - L1)
__device__ __forceinline__ float cd_static_diff(float b, float /*gd*/, float ginv, float nl1) { return __builtin_fmaf(b, ginv, nl1); }
__device__ __forceinline__ double cd_static_diff(double b, double gd, double ginv, double nl1) { return cd_quotient(b, gd, ginv) + nl1; }
__device__ __forceinline__ float lane_value(float v, int i) {
    return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), i));
}
__device__ __forceinline__ double lane_value(double v, int i) {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, i), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), i);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

// ---------------------------------------------------------------------------
// Static coordinate sweeps for the wave-per-column CD solves (IRLS, CV, explicit masks with their PER-COLUMN Gram; the small-side
// MSE solve with the shared one; lane = feature, the lane's column of the Gram in registers or an LDS tile):
// cd_nnls_col_fixed(G, b, x, L1 inside, nonneg, maxit, tol = 0) with the fixed-point exit, or with the relative-change stop.
// Every coordinate is visited in turn with wave-uniform control flow: each lane evaluates the step of ITS coordinate from its own
// residual and iterate (fma, med3), coordinate i's is read with one v_readlane at a compile-time lane and applied to all
// residuals with one fma on a compile-time register; the steps are collected per lane (v_writelane) and the iterate moves once
// per sweep (cd_static_one_sweep).  Dependent chain per coordinate: fma -> med3 -> readlane -> fma.  The form this replaces found the next coordinate that moves with a ballot and
// skipped the others: ~16 VALU + a dozen SALU operations and two branches per MOVING coordinate, all on one chain -- these
// kernels are latency-bound even at eight waves per SIMD (four waves: 1.8x slower), so the chain length is what counts (C5:
// NB iteration 32.4 -> 22.5 ms).  A dead diagonal / a lane beyond k holds 1/G_ii = 0 and no L1: its step is max(0, -x) = 0.
// gcol(std::integral_constant<int, i>) = G(i, lane).
// ---------------------------------------------------------------------------
template <int B, int E, class Fn> __device__ __forceinline__ void cd_static_for(Fn&& fn) {
    if constexpr (B < E) { fn(std::integral_constant<int, B>{}); cd_static_for<B + 1, E>(fn); }
}
// v_writelane: value (wave-uniform, an SGPR after v_readlane) into lane I of v.  Inline asm (this hipcc has no writelane builtin);
// the s_nop covers the VALU-writes-SGPR -> VALU-reads-it wait states hipcc itself places after a v_readlane.
template <int I> __device__ __forceinline__ float cd_write_lane(float v, float value) {
    asm volatile("s_nop 1\n\tv_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(value), "n"(I));
    return v;
}
template <int I> __device__ __forceinline__ double cd_write_lane(double v, double value) {
    unsigned long long u = __double_as_longlong(v);
    const unsigned long long w = __double_as_longlong(value);
    unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
    asm volatile("s_nop 1\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4" : "+v"(lo), "+v"(hi) : "s"((unsigned)w), "s"((unsigned)(w >> 32)), "n"(I));
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
// max(a, b) on the step's chain: fp32 as v_med3_f32(a, b, +inf) (no canonicalising pre-op; `inf` must be a run-time +inf), fp64 plain
__device__ __forceinline__ float cd_static_max(float a, float b, float inf) { return __builtin_amdgcn_fmed3f(a, b, inf); }
__device__ __forceinline__ double cd_static_max(double a, double b, double) { return __builtin_fmax(a, b); }
// One static sweep.  A coordinate is visited once per sweep, so the iterate a step sees is the one the sweep started with: the
// sweep only COLLECTS every lane's own step (v_writelane of the broadcast value into lane i) and the iterate moves once, after
// the last coordinate.  Per coordinate: fma, med3, v_readlane, v_writelane, fma.
// fp64: the quotient is the corrected one (cd_quotient above = b / G_ii); fp32 multiplies by 1/G_ii formed once per solve.
template <class T, int KP, class GC>
__device__ __forceinline__ T cd_static_one_sweep(T& b, T xe, T gd, T ginv, T nl1, T inf_rt, GC&& gcol) {
    T aown = T(0);
    cd_static_for<0, KP>([&](auto IC) {
        constexpr int i = decltype(IC)::value;
        const T diff = cd_static_diff(b, gd, ginv, nl1);
        const T ad = cd_static_max(diff, -xe, inf_rt);
        const T ad_i = lane_value(ad, i);
        b = tfma(-gcol(IC), ad_i, b);           // the chain goes on from here; collecting the step is off it
        aown = cd_write_lane<i>(aown, ad_i);
    });
    return aown;
}
template <class T, int KP, class GC>
__device__ __forceinline__ int cd_static_sweeps(T& b, T& x, T gd, bool fok, T l1, int nonneg, int maxit, GC&& gcol) {      // returns the sweeps executed
    const bool alive = fok && gd > T(0);
    const T ginv = alive ? T(1) / gd : T(0);          // one division per solve
    const T nl1 = alive ? -l1 : T(0);
    const T pinf = static_cast<T>(__builtin_inff());
    const T inf_rt = maxit >= 0 ? pinf : T(0);        // +inf at run time: with a literal LLVM folds the median back into maxnum
    for (int it = 0; it < maxit; ++it) {
        const T xe = !alive ? T(0) : (nonneg ? x : pinf);   // the clamp's operand: max(diff, -xe) is max(diff, -x) or diff; a dead diagonal (reference: `continue`) takes no step, whatever its warm x
        const T aown = cd_static_one_sweep<T, KP>(b, xe, gd, ginv, nl1, inf_rt, gcol);
        const T xn = x + aown;
        const bool moved = xn != x;
        x = xn;
        if (!__any(moved)) return it + 1;      // no effective step, or the iterate is at its floating-point fixed point
    }
    return maxit > 0 ? maxit : 0;
}
// The same sweeps with the reference's relative-change stop (explicit-mask solver, small-side MSE solver: cd_nnls with cd_tol, no
// L1 inside the step): the sum of |a_i| / (|x_i| + 1e-15) is one division per lane and a wave reduction per sweep (xor tree: the
// reference adds the terms in coordinate order).  Returns the number of sweeps.
template <class T, int KP, class GC>
__device__ __forceinline__ int cd_static_sweeps_tol(T& b, T& x, T gd, bool fok, int nonneg, int maxit, T tol, int k, GC&& gcol) {
    const bool alive = fok && gd > T(0);
    const T ginv = alive ? T(1) / gd : T(0);
    const T pinf = static_cast<T>(__builtin_inff());
    const T inf_rt = maxit >= 0 ? pinf : T(0);
    const bool check = tol > T(0);
    const T inv_k = T(1) / static_cast<T>(k);
    for (int it = 0; it < maxit; ++it) {
        const T xe = !alive ? T(0) : (nonneg ? x : pinf);
        const T aown = cd_static_one_sweep<T, KP>(b, xe, gd, ginv, T(0), inf_rt, gcol);
        const T xn = x + aown;
        const bool moved = xn != x;
        x = xn;
        if (check) {
            T term = fok ? (aown < T(0) ? -aown : aown) / ((x < T(0) ? -x : x) + T(1e-15)) : T(0);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) term += __shfl_xor(term, off, 64);
            if (term * inv_k < tol) return it + 1;
        } else if (!__any(moved)) return it + 1;
    }
    return maxit;
}
// fp32, Gram column in REGISTERS: the sweep carries the SCALED residual d = b / G_ii - L1 (the quotient the step starts from) instead
// of b, with the lane's Gram column scaled once per solve (gg_i = G(i, lane) / G(lane, lane)): d -= gg_i a_i is the same update in
// exact arithmetic, and the quotient's fma leaves the chain -- med3 -> readlane -> fma per coordinate.  Rounding differs from
// b * (1/G_ii) by an ulp here and there (fp32 fast path only; the fp64 kernels keep b).  gcol[] is overwritten.
template <int KP>
__device__ __forceinline__ int cd_static_sweeps_scaled_f32(float b, float& x, float gd, bool fok, float l1, int nonneg, int maxit,
                                                           float (&gcol)[KP]) {      // returns the sweeps executed
    const bool alive = fok && gd > 0.f;
    const float ginv = alive ? 1.f / gd : 0.f;
    const float pinf = __builtin_inff();
    const float inf_rt = maxit >= 0 ? pinf : 0.f;
#pragma unroll
    for (int i = 0; i < KP; ++i) gcol[i] *= ginv;
    float d = __builtin_fmaf(b, ginv, alive ? -l1 : 0.f);
    for (int it = 0; it < maxit; ++it) {
        const float xe = nonneg ? x : pinf;
        float aown = 0.f;
        cd_static_for<0, KP>([&](auto IC) {
            constexpr int i = decltype(IC)::value;
            const float ad = cd_static_max(d, -xe, inf_rt);
            const float ad_i = lane_value(ad, i);
            d = __builtin_fmaf(-gcol[i], ad_i, d);
            aown = cd_write_lane<i>(aown, ad_i);
        });
        const float xn = x + aown;
        const bool moved = xn != x;
        x = xn;
        if (!__any(moved)) return it + 1;
    }
    return maxit > 0 ? maxit : 0;
}
template <int KP, class GC>
__device__ __forceinline__ int cd_static_sweeps_f32(float& b, float& x, float gd, bool fok, float l1, int nonneg, int maxit, GC&& gcol) {
    return cd_static_sweeps<float, KP>(b, x, gd, fok, l1, nonneg, maxit, gcol);
}

__device__ __forceinline__ float shfl_xor_t(float v, int m) { return __shfl_xor(v, m, 64); }
__device__ __forceinline__ double shfl_xor_t(double v, int m) { return __shfl_xor(v, m, 64); }

// 8/16-byte vector types for whole-vector loads and stores
template <class T, int VEC> struct VecT;
template <> struct VecT<float, 1> { typedef float type; };
template <> struct VecT<float, 2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct VecT<float, 4> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct VecT<double, 1> { typedef double type; };
template <> struct VecT<double, 2> { typedef double type __attribute__((ext_vector_type(2))); };

// ---------------------------------------------------------------------------
// Gram  G = F F^T  (reference primitives/cpu/gram.hpp:37-67)
//
// MFMA kernels.  F (k x r) is streamed straight from HBM into the MFMA A/B operand registers:
// for v_mfma_f32_32x32x2_f32 lane l supplies A[i = l&31][kk = l>>5] = F[32*t + (l&31), c + (l>>5)],
// which is two contiguous 128-B runs per wave -- no LDS staging needed.  B operand of tile (ti,tj)
// is the A operand of row-tile tj, so a wave loads T operand registers per K-step and issues T MFMAs
// for the tile row blockIdx.y = ti.  The r dimension is split over waves; partial tiles are reduced
// in fixed order (in-block through LDS, then across blocks by gram_finalize) so the result is
// deterministic and bitwise symmetric.
// ---------------------------------------------------------------------------
// VL (vector loads): with T_TILES in {2, 4} and k % T_TILES == 0 the wave reads each column of F with ONE 8/16-byte load
// per lane (lane slot s takes rows T*s .. T*s+T-1) and tile t is made of the rows congruent to t mod T -- a row
// permutation the Gram is indifferent to, undone when the tile is written.  STEPS K-steps of loads are issued before
// their MFMAs: the loop is bound by HBM latency, not by the matrix pipe (64 cycles per instruction).
template <int T_TILES, bool VL, int STEPS>  // KP = 32 * T_TILES
__global__ __launch_bounds__(256) void gram_partial_f32(const float* __restrict__ F, int k, int64_t r,
                                                         float* __restrict__ partial) {
    constexpr int KP = 32 * T_TILES;
    __shared__ float red[3][T_TILES * 1024];
    const int ti = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t nw = (int64_t)gridDim.x * 4, wid = (int64_t)blockIdx.x * 4 + wave;
    const int64_t npairs = (r + 1) / 2;
    const int64_t per = (npairs + nw - 1) / nw;
    const int64_t p0 = wid * per;
    const int64_t p1 = p0 + per < npairs ? p0 + per : npairs;
    const int kk = lane >> 5, row = lane & 31;
    f32x16 acc[T_TILES];
#pragma unroll
    for (int t = 0; t < T_TILES; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
    for (int64_t pb = p0; pb < p1; pb += STEPS) {
        float a[STEPS][T_TILES];
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int64_t c = 2 * (pb + s) + kk;
            const bool cok = (pb + s) < p1 && c < r;
            const float* fc = F + c * (int64_t)k;
            if constexpr (VL) {
                typedef typename VecT<float, T_TILES>::type V;
                const int r0 = T_TILES * row;
                if (cok && r0 < k) {
                    const V v = *reinterpret_cast<const V*>(fc + r0);
#pragma unroll
                    for (int t = 0; t < T_TILES; ++t) a[s][t] = v[t];
                } else {
#pragma unroll
                    for (int t = 0; t < T_TILES; ++t) a[s][t] = 0.f;
                }
            } else {
#pragma unroll
                for (int t = 0; t < T_TILES; ++t) {
                    const int rr = 32 * t + row;
                    a[s][t] = (cok && rr < k) ? fc[rr] : 0.f;
                }
            }
        }
#pragma unroll
        for (int s = 0; s < STEPS; ++s)
#pragma unroll
            for (int t = 0; t < T_TILES; ++t) {
                float ai = a[s][0];
#pragma unroll
                for (int t2 = 1; t2 < T_TILES; ++t2) ai = (ti == t2) ? a[s][t2] : ai;
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, a[s][t], acc[t], 0, 0, 0);
            }
    }
    // in-block reduction, fixed order wave0 + wave1 + wave2 + wave3
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < T_TILES; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) red[wave - 1][t * 1024 + v * 64 + lane] = acc[t][v];
    }
    __syncthreads();
    if (wave == 0) {
        float* out = partial + ((int64_t)blockIdx.x * KP * KP);
#pragma unroll
        for (int t = 0; t < T_TILES; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                float s = acc[t][v];
                s += red[0][t * 1024 + v * 64 + lane];
                s += red[1][t * 1024 + v * 64 + lane];
                s += red[2][t * 1024 + v * 64 + lane];
                // C/D map of 32x32 MFMA: col = lane&31, row = (v&3) + 8*(v>>2) + 4*(lane>>5)
                const int is = (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5), js = lane & 31;
                const int i = VL ? T_TILES * is + ti : 32 * ti + is;
                const int j = VL ? T_TILES * js + t : 32 * t + js;
                out[(int64_t)j * KP + i] = s;
            }
    }
}

// k = 64 fp32 (two 32-row tiles, 8-byte vector loads: the headline shape): all four output tiles in ONE block, so every
// slice of F is read once instead of once per tile row.  Same waves, same column pairs, same in-block and across-block
// summation order as gram_partial_f32<2, true, STEPS> on a (nblk, 2) grid -- bitwise the same partial tiles.
// SCALE (the iteration's tail, kernels_tail.hip.h): every element of F is loaded by exactly one lane of one block, so the lane can
// divide it by its row's d (formed from the row sums exactly as scale_rows_from_sums does), store it back and feed the SCALED value to
// the matrix cores: extract_scaling's second pass and the Gram partials in one pass over F, bit-identical to the two kernels.
template <int STEPS, bool SCALE>
__device__ __forceinline__ void gram_partial_f32_k64_body(float* __restrict__ F, int k, int64_t r, float* __restrict__ partial,
                                                          const unsigned bid, const unsigned nb, const float* __restrict__ sums,
                                                          int norm_type) {
    constexpr int KP = 64;
    __shared__ float red[3][4 * 1024];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t nw = (int64_t)nb * 4, wid = (int64_t)bid * 4 + wave;
    const int64_t npairs = (r + 1) / 2;
    const int64_t per = (npairs + nw - 1) / nw;
    const int64_t p0 = wid * per;
    const int64_t p1 = p0 + per < npairs ? p0 + per : npairs;
    const int kk = lane >> 5, row = lane & 31;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;
    typedef VecT<float, 2>::type V;
    float dv0 = 1.f, dv1 = 1.f;              // SCALE: d of this lane's two rows (2 row, 2 row + 1)
    if constexpr (SCALE) {
        if (2 * row < k) {
            float s0 = sums[2 * row], s1 = sums[2 * row + 1];
            if (norm_type == 1) { s0 = sqrtf(s0); s1 = sqrtf(s1); }
            dv0 = s0 + 1e-15f; dv1 = s1 + 1e-15f;
        }
    }
    for (int64_t pb = p0; pb < p1; pb += STEPS) {
        float a[STEPS][2];
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int64_t c = 2 * (pb + s) + kk;
            const bool cok = (pb + s) < p1 && c < r;
            const int r0 = 2 * row;
            if (cok && r0 < k) {
                V v = *reinterpret_cast<const V*>(F + c * (int64_t)k + r0);
                if constexpr (SCALE) {
                    v[0] = v[0] / dv0; v[1] = v[1] / dv1;
                    *reinterpret_cast<V*>(F + c * (int64_t)k + r0) = v;
                }
                a[s][0] = v[0]; a[s][1] = v[1];
            } else { a[s][0] = 0.f; a[s][1] = 0.f; }
        }
#pragma unroll
        for (int s = 0; s < STEPS; ++s)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    acc[ti][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][ti], a[s][t], acc[ti][t], 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int v = 0; v < 16; ++v) red[wave - 1][(ti * 2 + t) * 1024 + v * 64 + lane] = acc[ti][t][v];
    }
    __syncthreads();
    if (wave == 0) {
        float* out = partial + ((int64_t)bid * KP * KP);
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    float s = acc[ti][t][v];
                    s += red[0][(ti * 2 + t) * 1024 + v * 64 + lane];
                    s += red[1][(ti * 2 + t) * 1024 + v * 64 + lane];
                    s += red[2][(ti * 2 + t) * 1024 + v * 64 + lane];
                    const int is = (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5), js = lane & 31;
                    out[(int64_t)(2 * js + t) * KP + (2 * is + ti)] = s;
                }
    }
}
template <int STEPS>
__global__ __launch_bounds__(256) void gram_partial_f32_k64(const float* __restrict__ F, int k, int64_t r,
                                                             float* __restrict__ partial) {
    gram_partial_f32_k64_body<STEPS, false>(const_cast<float*>(F), k, r, partial, blockIdx.x, gridDim.x, nullptr, 0);
}

template <int T_TILES>  // KP = 16 * T_TILES  (v_mfma_f64_16x16x4_f64)
__global__ __launch_bounds__(256) void gram_partial_f64(const double* __restrict__ F, int k, int64_t r,
                                                         double* __restrict__ partial) {
    constexpr int KP = 16 * T_TILES;
    __shared__ double red[3][T_TILES * 256];
    const int ti = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t nw = (int64_t)gridDim.x * 4, wid = (int64_t)blockIdx.x * 4 + wave;
    const int64_t nquad = (r + 3) / 4;
    const int64_t per = (nquad + nw - 1) / nw;
    const int64_t p0 = wid * per;
    const int64_t p1 = p0 + per < nquad ? p0 + per : nquad;
    const int kk = lane >> 4, row = lane & 15;
    f64x4 acc[T_TILES];
#pragma unroll
    for (int t = 0; t < T_TILES; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[t][v] = 0.0;
    const int rowi = 16 * ti + row;
#pragma unroll 2
    for (int64_t p = p0; p < p1; ++p) {
        const int64_t c = 4 * p + kk;
        const bool cok = c < r;
        const double* fc = F + c * (int64_t)k;
        const double ai = (cok && rowi < k) ? fc[rowi] : 0.0;
        double a[T_TILES];
#pragma unroll
        for (int t = 0; t < T_TILES; ++t) {
            const int rr = 16 * t + row;
            a[t] = (cok && rr < k) ? fc[rr] : 0.0;
        }
#pragma unroll
        for (int t = 0; t < T_TILES; ++t)
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, a[t], acc[t], 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < T_TILES; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) red[wave - 1][t * 256 + v * 64 + lane] = acc[t][v];
    }
    __syncthreads();
    if (wave == 0) {
        double* out = partial + ((int64_t)blockIdx.x * KP * KP);
#pragma unroll
        for (int t = 0; t < T_TILES; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                double s = acc[t][v];
                s += red[0][t * 256 + v * 64 + lane];
                s += red[1][t * 256 + v * 64 + lane];
                s += red[2][t * 256 + v * 64 + lane];
                // C/D map of the f64 16x16x4 MFMA: col = lane&15, row = (lane>>4) + 4*v
                const int i = 16 * ti + (lane >> 4) + 4 * v;
                const int j = 16 * t + (lane & 15);
                out[(int64_t)j * KP + i] = s;
            }
    }
}

// Sum the per-block partial tiles (fixed order -> deterministic), add eps then l2 on the diagonal
// (gram.hpp:51 tiny_num, fit_cpu.hpp:506/738 L2), write the k x k result.  32 lanes per output element stride over
// the blocks, then a fixed-shape xor-shuffle tree (a serial loop over up to 512 partials cost 25 us of pure latency).
template <class T>
__device__ __forceinline__ void gram_finalize_body(const T* __restrict__ partial, int nblk, int KP, int k, T eps, T l2,
                                                   T* __restrict__ G, const unsigned bid) {
    const int e = bid * 8 + (threadIdx.x >> 5);
    const int sl = threadIdx.x & 31;
    T s = 0;
    if (e < KP * KP)
        for (int b = sl; b < nblk; b += 32) s += partial[(int64_t)b * KP * KP + e];
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) s += shfl_xor_t(s, off);
    if (e >= KP * KP || sl != 0) return;
    const int i = e % KP, j = e / KP;
    if (i >= k || j >= k) return;
    if (i == j) { s += eps; s += l2; }
    G[(int64_t)j * k + i] = s;
}
template <class T>
__global__ __launch_bounds__(256) void gram_finalize(const T* __restrict__ partial, int nblk, int KP, int k, T eps, T l2,
                                                      T* __restrict__ G) {
    gram_finalize_body<T>(partial, nblk, KP, k, eps, l2, G, blockIdx.x);
}

// ---------------------------------------------------------------------------
// RHS  B(:,j) = sum_{i in nz(j)} A(i,j) F(:,i)   (reference primitives/cpu/rhs.hpp:52-70,
// fused_nnls.hpp:109-114).  The SpMM-like, HBM/L2-bound kernel of the path.
//
// One wavefront per output column.  The wave is split into NG = 64/LPN lane groups; a group of LPN
// lanes covers one k-vector of F with VEC contiguous elements per lane (16-byte loads when k allows),
// so one wave-wide load instruction gathers NG rows of F (k=64 fp32: 4 rows, 1 KiB).  Group g walks
// nonzeros start+g, start+g+NG, ...; (row, value) pairs are read with group-uniform addresses
// (consecutive int32 / T, cache-line coalesced).  U independent gathers are kept in flight per lane.
// Group partial sums are combined with xor-shuffles at the end.
// ---------------------------------------------------------------------------
template <class T, int VEC, int LPN, int U, bool NT>
__global__ __launch_bounds__(256) void rhs_kernel(const int* __restrict__ colptr,
                                                   const int* __restrict__ rowidx,
                                                   const T* __restrict__ vals, int64_t ncols,
                                                   const T* __restrict__ F, int k,
                                                   T* __restrict__ B) {
    constexpr int NG = 64 / LPN;
    const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= ncols) return;
    const int lane = threadIdx.x & 63;
    const int g = lane / LPN, li = lane % LPN;
    const int f0 = li * VEC;
    const bool fok = f0 < k;
    const int start = colptr[j], end = colptr[j + 1];
    T acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = T(0);
    const T* Fl = F + f0;
    for (int t = start + g; t < end; t += NG * U) {
        int rr[U];
        T vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int tt = t + u * NG;
            const bool ok = tt < end;
            rr[u] = ok ? rowidx[tt] : 0;
            vv[u] = ok ? vals[tt] : T(0);
        }
        T ff[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (fok) {
                const T* src = Fl + (int64_t)rr[u] * k;
                if constexpr (VEC == 1) {
                    ff[u][0] = src[0];
                } else {
                    typedef typename VecT<T, VEC>::type V;
                    V v;
                    if constexpr (NT) v = __builtin_nontemporal_load(reinterpret_cast<const V*>(src));
                    else v = *reinterpret_cast<const V*>(src);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) ff[u][e] = v[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) ff[u][e] = T(0);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = tfma(vv[u], ff[u][e], acc[e]);
    }
#pragma unroll
    for (int off = LPN; off < 64; off <<= 1)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += shfl_xor_t(acc[e], off);
    if (g == 0 && fok) {
        T* dst = B + j * (int64_t)k + f0;
        if constexpr (VEC == 1) {
            dst[0] = acc[0];
        } else {
            typedef typename VecT<T, VEC>::type V;
            V v;
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = acc[e];
            *reinterpret_cast<V*>(dst) = v;
        }
    }
}

// ---------------------------------------------------------------------------
// RHS, general-rank fallback (k > 64 with k not a multiple of the 16-byte vector): one wavefront per column, lane l owns
// features l, l+64, ... (EPL per lane), nonzeros in order -- the reference's summation order (rhs.hpp:59-63).
// ---------------------------------------------------------------------------
template <class T, int EPL>
__global__ __launch_bounds__(256) void rhs_generic_kernel(const int* __restrict__ colptr, const int* __restrict__ rowidx,
                                                           const T* __restrict__ vals, int64_t ncols,
                                                           const T* __restrict__ F, int k, T* __restrict__ B) {
    const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= ncols) return;
    const int lane = threadIdx.x & 63;
    T acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = T(0);
    const int start = colptr[j], end = colptr[j + 1];
#pragma unroll 4
    for (int t = start; t < end; ++t) {
        const int row = rowidx[t];
        const T v = vals[t];
        const T* src = F + (int64_t)row * k;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int f = lane + 64 * e;
            if (f < k) acc[e] = tfma(v, src[f], acc[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int f = lane + 64 * e;
        if (f < k) B[j * (int64_t)k + f] = acc[e];
    }
}

// ---------------------------------------------------------------------------
// RHS, staged-index variant ("rhs_stage_kernel").  Same lane-group mapping and the same summation order as rhs_kernel
// (bitwise identical results), but the (row, value) stream of the column is fetched COALESCED -- one vector load per
// NG*U nonzeros instead of one group-uniform load per nonzero -- and handed to the lane groups through the LDS crossbar
// (ds_bpermute).  rhs_kernel spends a third of its texture-addresser time on those group-uniform index/value loads;
// here the only per-nonzero vector-memory instruction left is the gather itself.
// ---------------------------------------------------------------------------
template <class T, int VEC, int LPN, int U>
__global__ __launch_bounds__(256) void rhs_stage_kernel(const int* __restrict__ colptr,
                                                         const int* __restrict__ rowidx,
                                                         const T* __restrict__ vals, int64_t ncols,
                                                         const T* __restrict__ F, int k,
                                                         T* __restrict__ B) {
    constexpr int NG = 64 / LPN;
    constexpr int CH = NG * U;              // nonzeros per staged chunk (<= 64)
    static_assert(CH <= 64, "one lane per staged nonzero");
    const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= ncols) return;
    const int lane = threadIdx.x & 63;
    const int g = lane / LPN, li = lane % LPN;
    const int f0 = li * VEC;
    const bool fok = f0 < k;
    const int start = colptr[j], end = colptr[j + 1];
    T acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = T(0);
    const T* Fl = F + (fok ? f0 : 0);
    // The NEXT chunk's (row, value) lanes are requested before the current chunk's gathers are issued.  The loads are
    // unconditional (address clamped into the column, value masked afterwards): a predicated load becomes a branch,
    // and hipcc then sinks the prefetch to the end of the iteration where its whole latency is exposed.
    int ri_n = 0;
    T vi_n = T(0);
    if (start < end) {
        const int tt = start + lane;
        const int tc = tt < end ? tt : end - 1;
        const int r0 = rowidx[tc];
        const T v0 = vals[tc];
        const bool ok = lane < CH && tt < end;
        ri_n = ok ? r0 : 0;                           // past the end: row 0 with weight 0, as rhs_kernel does
        vi_n = ok ? v0 : T(0);
    }
    for (int t0 = start; t0 < end; t0 += CH) {
        const int ri = ri_n;
        const T vi = vi_n;
        {
            const int tt = t0 + CH + lane;
            const int tc = tt < end ? tt : end - 1;
            const int r0 = rowidx[tc];
            const T v0 = vals[tc];
            __builtin_amdgcn_sched_barrier(0);
            const bool ok = lane < CH && tt < end;
            ri_n = ok ? r0 : 0;
            vi_n = ok ? v0 : T(0);
        }
        T ff[U][VEC];
        T vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int nz = u * NG + g;               // group g takes nonzeros g, g + NG, ... (rhs_kernel's order)
            const int r = __shfl(ri, nz, 64);
            vv[u] = __shfl(vi, nz, 64);
            const T* src = Fl + (int64_t)r * k;
            if constexpr (VEC == 1) {
                ff[u][0] = src[0];
            } else {
                typedef typename VecT<T, VEC>::type V;
                const V v = *reinterpret_cast<const V*>(src);
#pragma unroll
                for (int e = 0; e < VEC; ++e) ff[u][e] = v[e];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = tfma(vv[u], ff[u][e], acc[e]);
    }
#pragma unroll
    for (int off = LPN; off < 64; off <<= 1)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += shfl_xor_t(acc[e], off);
    if (g == 0 && fok) {
        T* dst = B + j * (int64_t)k + f0;
        if constexpr (VEC == 1) {
            dst[0] = acc[0];
        } else {
            typedef typename VecT<T, VEC>::type V;
            V v;
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = acc[e];
            *reinterpret_cast<V*>(dst) = v;
        }
    }
}

// ---------------------------------------------------------------------------
// RHS, scalarised-index variant for 33 <= k <= 128 ("rhs_wave_kernel").
// Measured on MI355X the lane-group kernel above is bound by the texture-addresser (TA), not by L2 or HBM: with the
// gathered factor shrunk to 64 KB (all L1 hits) it still tops out at 17 TB/s, because every group-uniform (row, value)
// fetch is a full vector-memory instruction.  Here a whole wavefront owns one nonzero at a time, so the column's
// (row, value) stream is WAVE-UNIFORM and is read with scalar loads (s_load through the scalar cache, no TA traffic);
// the only vector-memory instructions left are the gathers themselves (one k-vector of F per instruction, EPL = k/64
// elements per lane).  A side effect: each feature is accumulated by ONE lane in nonzero order -- the reference's
// summation order (rhs.hpp:59-63) up to fma contraction.
// ---------------------------------------------------------------------------
template <class T, int EPL, int U>
__global__ __launch_bounds__(256) void rhs_wave_kernel(const int* __restrict__ colptr,
                                                        const int* __restrict__ rowidx,
                                                        const T* __restrict__ vals, int64_t ncols,
                                                        const T* __restrict__ F, int k, T* __restrict__ B) {
    const int64_t j = __builtin_amdgcn_readfirstlane((int)((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (j >= ncols) return;
    const int lane = threadIdx.x & 63;
    const bool fok = lane * EPL < k;               // k is a multiple of EPL
    const int f0 = fok ? lane * EPL : 0;           // lanes beyond k re-read feature 0 and are never stored: no branches
    const int start = colptr[j], end = colptr[j + 1];
    T acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = T(0);
    const T* Fl = F + f0;
    auto gather_fma = [&](const int (&rr)[U], const T (&vv)[U]) {
        T ff[U][EPL];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const T* src = Fl + (int64_t)rr[u] * k;
            if constexpr (EPL == 1) ff[u][0] = src[0];
            else {
                typedef typename VecT<T, EPL>::type V;
                const V v = *reinterpret_cast<const V*>(src);
#pragma unroll
                for (int e = 0; e < EPL; ++e) ff[u][e] = v[e];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int e = 0; e < EPL; ++e) acc[e] = tfma(vv[u], ff[u][e], acc[e]);
    };
    // U consecutive dword-aligned elements as ONE wide scalar load (s_load_dwordx8 only needs 4-byte alignment)
    struct __attribute__((packed, aligned(4))) IdxPack { int v[U]; };
    struct __attribute__((packed, aligned(4))) ValPack { T v[U]; };
    int t = start;
    for (; t + U <= end; t += U) {               // full batches: contiguous wave-uniform reads -> wide scalar loads
        const IdxPack ip = *reinterpret_cast<const IdxPack*>(rowidx + t);
        const ValPack vp = *reinterpret_cast<const ValPack*>(vals + t);
        int rr[U];
        T vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { rr[u] = ip.v[u]; vv[u] = vp.v[u]; }
        gather_fma(rr, vv);
    }
    if (t < end) {                               // tail batch
        int rr[U];
        T vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = t + u < end;
            rr[u] = ok ? rowidx[t + u] : 0;
            vv[u] = ok ? vals[t + u] : T(0);
        }
        gather_fma(rr, vv);
    }
    if (fok) {
        T* dst = B + j * (int64_t)k + f0;
        if constexpr (EPL == 1) dst[0] = acc[0];
        else {
            typedef typename VecT<T, EPL>::type V;
            V v;
#pragma unroll
            for (int e = 0; e < EPL; ++e) v[e] = acc[e];
            *reinterpret_cast<V*>(dst) = v;
        }
    }
}

// ---------------------------------------------------------------------------
// Padding helper for the solve kernels: Gp (KP x KP) = G (k x k) with identity on the padded
// diagonal; invd[i] = 1/G_ii (0 if G_ii <= 0: the reference skips such coordinates,
// nnls_batch.hpp:88-89).
// ---------------------------------------------------------------------------
template <class T>
__global__ void pad_gram(const T* __restrict__ G, int k, int KP, T* __restrict__ Gp,
                         T* __restrict__ invd) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= KP * KP) return;
    const int i = e % KP, j = e / KP;
    T v = (i < k && j < k) ? G[(int64_t)j * k + i] : (i == j ? T(1) : T(0));
    Gp[e] = v;
    if (i == j) invd[i] = v > T(0) ? T(1) / v : T(0);
}

// ---------------------------------------------------------------------------
// CD NNLS, one LANE per column ("lane" variant).
// Reference: primitives/cpu/nnls_batch.hpp:70-132 cd_nnls_col_fixed, with the prologue of
// fused_nnls.hpp:116-123.  A wave solves 64 columns at once: the residual b (KP registers per lane)
// and x live with the lane, G(:,i) is wave-uniform and is read through the scalar cache into SGPRs,
// so the rank-1 residual update of a coordinate is KP v_fma with an SGPR operand and every scalar
// decision of the reference (clamp, skip, tolerance) is lane-parallel.  Per-column early exit
// (cd_tol) freezes the lane through the exec mask; the wave leaves when all its columns are done.
// A coordinate whose step is 0 is a no-op in the reference (`continue`); here it executes
// b -= G(:,i)*0 and tol += 0, which is the same arithmetic result.
//   EXACT = true : IEEE divisions as the reference writes them (fp64 parity mode)
//   EXACT = false: multiply by precomputed 1/G_ii, v_rcp for the tolerance term (fp32 throughput mode)
// ---------------------------------------------------------------------------
template <class T> __device__ __forceinline__ T fast_div(T a, T b);
template <> __device__ __forceinline__ float fast_div<float>(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
template <> __device__ __forceinline__ double fast_div<double>(double a, double b) { return a / b; }

template <class T, int KP, bool EXACT>
__global__ __launch_bounds__(64) void cd_lane_kernel(const T* __restrict__ Gp,
                                                      const T* __restrict__ invd,
                                                      const T* __restrict__ B, T* __restrict__ X,
                                                      int k, int64_t ncols, T l1_pre, int warm,
                                                      int zero_init, T l1_cd, T l2_cd, int nonneg,
                                                      int maxit, T tol, T ub_cd, T ub_post,
                                                      int* __restrict__ sweeps, const int* __restrict__ order) {
    const int64_t slot_j = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool inb = slot_j < ncols;
    const int64_t j = (inb && order) ? order[slot_j] : slot_j;
    T b[KP], x[KP];
    const T* bj = B + j * (int64_t)k;
    T* xj = X + j * (int64_t)k;
#pragma unroll
    for (int i = 0; i < KP; ++i) {
        b[i] = (inb && i < k) ? bj[i] : T(0);
        x[i] = (inb && i < k && !zero_init) ? xj[i] : T(0);
    }
    if (l1_pre != T(0)) {
#pragma unroll
        for (int i = 0; i < KP; ++i)
            if (i < k) b[i] -= l1_pre;
    }
    if (warm) {  // b -= G x   (fused_nnls.hpp:121-123)
#pragma unroll
        for (int c = 0; c < KP; ++c) {
            const T xc = x[c];
#pragma unroll
            for (int r = 0; r < KP; ++r) b[r] = tfma(-Gp[c * KP + r], xc, b[r]);
        }
    }
    const bool has_upper = ub_cd > T(0);
    const bool check = tol > T(0);
    const T inv_k = T(1) / static_cast<T>(k);
    bool active = inb;
    int nsweep = 0;
    for (int it = 0; it < maxit; ++it) {
        if (!__any(active)) break;
        if (active) {
            ++nsweep;
            T tol_sum = T(0);
#pragma unroll
            for (int i = 0; i < KP; ++i) {
                const T ginv = invd[i];
                if (ginv > T(0)) {   // wave-uniform: reference `if (g_diag <= 0) continue;`
                    T diff;
                    if constexpr (EXACT) diff = b[i] / Gp[i * KP + i];
                    else diff = b[i] * ginv;
                    if (l1_cd != T(0)) diff -= l1_cd;
                    if (l2_cd != T(0)) diff += l2_cd * x[i];
                    const T nv = x[i] + diff;
                    T ad = diff, nx = nv;
                    if (nonneg && nv < T(0)) { ad = -x[i]; nx = T(0); }
                    else if (has_upper && nv > ub_cd) { ad = ub_cd - x[i]; nx = ub_cd; }
                    x[i] = nx;
                    if (check) {
                        if constexpr (EXACT) tol_sum += tabs(ad) / (tabs(nx) + T(1e-15));
                        else tol_sum += fast_div<T>(tabs(ad), tabs(nx) + T(1e-15));
                    }
#pragma unroll
                    for (int r = 0; r < KP; ++r) b[r] = tfma(-Gp[i * KP + r], ad, b[r]);
                }
            }
            if (check && tol_sum * inv_k < tol) active = false;
        }
    }
    if (inb) {
#pragma unroll
        for (int i = 0; i < KP; ++i)
            if (i < k) {
                T v = x[i];
                if (ub_post > T(0)) v = v < ub_post ? v : ub_post;
                xj[i] = v;
            }
        if (sweeps) sweeps[j] = nsweep;   // what cd_nnls_col_fixed returns (nnls_batch.hpp:127-131)
    }
}

// ---------------------------------------------------------------------------
// CD NNLS, LPC adjacent lanes per column ("group" variant): the default solve kernel.
// Same reference routine (nnls_batch.hpp:70-132).  A column's residual is split over LPC = 1, 2 or 4 adjacent
// lanes (RPL = KP/LPC rows each, in registers together with the matching slice of x); a wave carries 64/LPC
// columns.  Per coordinate i the owner lane's b_i and x_i are broadcast inside the group with one DPP
// quad_perm move each (no LDS, no readlane), every lane of the group evaluates the reference's scalar step
// redundantly (so `a`, the clamp decisions and the tolerance sum are bit-identical across the group), and each
// lane applies the rank-1 residual update to its own rows with G(:,i) read from LDS (16-byte reads; lanes with
// the same sub-index share an address -> broadcast, LPC distinct addresses on distinct banks).
// Why: the rank-1 update is only KP/LPC fmas per lane, so the sequential per-coordinate dependency chain
// (the real limiter of CD: measured 460-680 cycles/coordinate for LPC = 1 on MI355X) shrinks, and the number of
// waves grows LPC-fold, which is what small column counts (the W side: m = 20000 -> 313 waves at LPC = 1)
// need to fill 1024 SIMDs.  The launcher picks LPC from the column count.
// Finished columns are frozen by forcing the step to 0.
// ---------------------------------------------------------------------------
template <int OWNER, int LPC> __device__ __forceinline__ int dpp_bcast_i32(int v) {
    if constexpr (LPC == 1) return v;
    else if constexpr (LPC == 2) {
        // quad [a b c d] -> owner 0: [a a c c], owner 1: [b b d d]
        constexpr int ctrl = OWNER == 0 ? (0 | (0 << 2) | (2 << 4) | (2 << 6)) : (1 | (1 << 2) | (3 << 4) | (3 << 6));
        return __builtin_amdgcn_mov_dpp(v, ctrl, 0xf, 0xf, true);
    } else {
        constexpr int ctrl = OWNER | (OWNER << 2) | (OWNER << 4) | (OWNER << 6);
        return __builtin_amdgcn_mov_dpp(v, ctrl, 0xf, 0xf, true);
    }
}
template <int OWNER, int LPC> __device__ __forceinline__ float dpp_bcast(float v) {
    return __int_as_float(dpp_bcast_i32<OWNER, LPC>(__float_as_int(v)));
}
template <int OWNER, int LPC> __device__ __forceinline__ double dpp_bcast(double v) {
    const long long bits = __double_as_longlong(v);
    const int lo = dpp_bcast_i32<OWNER, LPC>((int)(bits & 0xffffffffll));
    const int hi = dpp_bcast_i32<OWNER, LPC>((int)(bits >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <class T, int KP, int LPC, bool EXACT, int I>
struct CdGroupStep {
    // One coordinate (compile-time index I) of the sweep, then recurse to I+1.  `gcur` holds this lane's rows of
    // G(:,I) (fetched one coordinate ahead); the rows of G(:,I+1) are requested first thing so their LDS latency
    // hides behind this coordinate's arithmetic.  A scheduling barrier per coordinate keeps the compiler from
    // hoisting further ahead (it otherwise floods the register file: 256 VGPRs + scratch on this 5k-instruction block).
    template <class GV, int NV>
    static __device__ __forceinline__ void run(T (&b)[KP / LPC], T (&x)[KP / LPC], GV (&gcur)[NV],
                                               const T* __restrict__ Gs, const T* __restrict__ diag,
                                               const T* __restrict__ invs, int sub_off, T l1_cd, T l2_cd, T lo, T hi,
                                               bool active, bool check, T& tol_sum) {
        constexpr int RPL = KP / LPC;
        constexpr int OWNER = I / RPL, LI = I % RPL;
        constexpr int EV = 16 / sizeof(T);
        constexpr int INEXT = (I + 1) % KP;            // the last coordinate prefetches column 0 for the next sweep
        GV gnext[NV];
        const T* gcol = Gs + INEXT * KP + sub_off;
#pragma unroll
        for (int q = 0; q < NV; ++q) gnext[q] = *reinterpret_cast<const GV*>(gcol + q * EV);
        const T bi = dpp_bcast<OWNER, LPC>(b[LI]);
        const T xo = dpp_bcast<OWNER, LPC>(x[LI]);
        const T ginv = invs[I];
        T diff;
        if constexpr (EXACT) diff = bi / diag[I];
        else diff = bi * ginv;
        diff -= l1_cd;                       // reference: `if (L1 != 0) diff -= L1` -- subtracting 0 is exact
        diff = tfma(l2_cd, xo, diff);        // reference: `if (L2 != 0) diff += L2 * x[i]`
        const T nv = xo + diff;
        const bool neg = nv < lo, up = nv > hi;
        T nx = neg ? lo : (up ? hi : nv);
        T a = neg ? lo - xo : (up ? hi - xo : diff);
        const bool on = active && (ginv > T(0));       // reference: `if (g_diag <= 0) continue;`
        a = on ? a : T(0);
        nx = on ? nx : xo;
        x[LI] = (sub_off == OWNER * RPL) ? nx : x[LI];
        // accumulated unconditionally (it is only READ when cd_tol > 0): a branch here splits the sweep into
        // basic blocks and LLVM then sinks the residual updates towards their uses, keeping ~KP*RPL values live
        if constexpr (EXACT) tol_sum += tabs(a) / (tabs(nx) + T(1e-15));
        else tol_sum += fast_div<T>(tabs(a), tabs(nx) + T(1e-15));
        const T na = -a;                                 // b -= G(:,I) * a
#pragma unroll
        for (int q = 0; q < NV; ++q)
#pragma unroll
            for (int e = 0; e < EV; ++e) b[q * EV + e] = tfma(gcur[q][e], na, b[q * EV + e]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (I + 1 < KP) {
            CdGroupStep<T, KP, LPC, EXACT, I + 1>::template run<GV, NV>(b, x, gnext, Gs, diag, invs, sub_off, l1_cd, l2_cd,
                                                                        lo, hi, active, check, tol_sum);
        } else {
#pragma unroll
            for (int q = 0; q < NV; ++q) gcur[q] = gnext[q];   // hand column 0 back to the caller's buffer
        }
    }
};


// one atomic pair per wavefront: sum of the per-lane sweep counts (0 on lanes that own no column) -> ctx counters
__device__ __forceinline__ void cd_stats_add(unsigned long long* stats, int nsw, int ncol) {
    if (!stats) return;
    int v = nsw, c = ncol;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { v += __shfl_xor(v, off, 64); c += __shfl_xor(c, off, 64); }
    if ((threadIdx.x & 63) == 0 && c > 0) {
        atomicAdd(stats, (unsigned long long)v);
        atomicAdd(stats + 1, (unsigned long long)c);
    }
}

// ---------------------------------------------------------------------------
// CD NNLS for SMALL sides (fewer columns than ~1.5 waves per SIMD: C3's 610 columns, hawaiibirds): one wavefront per column,
// lane = coordinate, the lane's column of the SHARED Gram in registers, static coordinate sweeps with the relative-change stop
// (cd_static_sweeps_tol).  With so few columns every wave runs alone on its SIMD and the solve lasts as long as one column's
// dependent chain: four instructions per coordinate here against ~150 cycles per coordinate for a lone 32-column MFMA tile
// (C3, k = 32, 610 columns).  Non-negativity only (what the NMF half-updates use); reference
// nnls_batch.hpp:70-132, prologue fused_nnls.hpp:116-123.
// ---------------------------------------------------------------------------
template <class T, int KP>
__global__ __launch_bounds__(256) void cd_wave_static_kernel(const T* __restrict__ G /* k x k */, const T* __restrict__ B,
                                                              T* __restrict__ X, int k, int64_t ncols, T l1_pre, int warm,
                                                              int zero_init, int maxit, T tol, T ub_post,
                                                              int* __restrict__ sweeps, unsigned long long* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (j >= ncols) return;                      // whole waves only
    const bool fok = lane < k;
    const int ll = fok ? lane : 0;
    T gcol[KP];                                  // G(i, lane), identity padding
#pragma unroll
    for (int i = 0; i < KP; ++i) gcol[i] = (fok && i < k) ? G[(int64_t)i * k + ll] : (i == lane ? T(1) : T(0));
    const T gd = fok ? G[(int64_t)ll * k + ll] : T(1);
    T b = fok ? B[j * (int64_t)k + ll] - l1_pre : T(0);        // b - 0 is exact
    T x = (fok && !zero_init) ? X[j * (int64_t)k + ll] : T(0);
    if (warm) {                                  // b -= G x
        const T xw = x;
        cd_static_for<0, KP>([&](auto IC) {
            constexpr int i = decltype(IC)::value;
            b = tfma(-gcol[i], lane_value(xw, i), b);
        });
    }
    const int nsweep = cd_static_sweeps_tol<T, KP>(b, x, gd, fok, 1, maxit, tol, k, [&](auto IC) { return gcol[decltype(IC)::value]; });
    if (fok) {
        T val = x;
        if (ub_post > T(0)) val = val < ub_post ? val : ub_post;
        X[j * (int64_t)k + lane] = val;
    }
    if (sweeps && lane == 0) sweeps[j] = nsweep;
    cd_stats_add(stats, lane == 0 ? nsweep : 0, lane == 0 ? 1 : 0);
}

template <class T, int KP, int LPC, bool EXACT>
__global__ __launch_bounds__(256) void cd_group_kernel(const T* __restrict__ Gp, const T* __restrict__ invd,
                                                        const T* __restrict__ B, T* __restrict__ X, int k,
                                                        int64_t ncols, T l1_pre, int warm, int zero_init, T l1_cd,
                                                        T l2_cd, int nonneg, int maxit, T tol, T ub_cd, T ub_post,
                                                        int* __restrict__ sweeps, const int* __restrict__ order,
                                                        unsigned long long* __restrict__ stats) {
    constexpr int RPL = KP / LPC;               // rows per lane
    constexpr int CPW = 64 / LPC;               // columns per wave
    constexpr int EV = 16 / sizeof(T);
    constexpr int NV = RPL / EV;
    static_assert(RPL % EV == 0, "rows per lane must be a multiple of the 16-byte vector");
    typedef typename VecT<T, EV>::type GV;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* Gs = reinterpret_cast<T*>(smem_raw);     // KP*KP, column i contiguous
    T* diag = Gs + KP * KP;                     // KP
    T* invs = diag + KP;                        // KP
    for (int e = threadIdx.x; e < KP * KP; e += blockDim.x) Gs[e] = Gp[e];
    for (int e = threadIdx.x; e < KP; e += blockDim.x) { diag[e] = Gp[e * KP + e]; invs[e] = invd[e]; }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int sub = lane % LPC, slot = lane / LPC;
    const int sub_off = sub * RPL;
    const int64_t slot_j = ((int64_t)blockIdx.x * (blockDim.x >> 6) + wave) * CPW + slot;
    const bool inb = slot_j < ncols;
    // optional work order: columns sorted by the sweeps they needed last time, so a wave's columns finish together
    const int64_t j = (inb && order) ? order[slot_j] : slot_j;
    T b[RPL], x[RPL];
    const T* bj = B + j * (int64_t)k;
    T* xj = X + j * (int64_t)k;
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
        const int row = sub_off + r;
        T bv = (inb && row < k) ? bj[row] : T(0);
        if (l1_pre != T(0) && inb && row < k) bv -= l1_pre;
        b[r] = bv;
        x[r] = (inb && row < k && !zero_init) ? xj[row] : T(0);
    }
    if (warm) {   // b -= G x   (fused_nnls.hpp:121-123): x_c broadcast from its owner, own rows of G(:,c) from LDS
#pragma unroll 1
        for (int o = 0; o < LPC; ++o) {
#pragma unroll
            for (int li = 0; li < RPL; ++li) {
                const int c = o * RPL + li;
                T xc;
                if constexpr (LPC == 1) xc = x[li];
                else xc = __shfl(x[li], (lane & ~(LPC - 1)) + o, 64);
                const T nxc = -xc;
                const T* gcol = Gs + c * KP + sub_off;
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    const GV g = *reinterpret_cast<const GV*>(gcol + q * EV);
#pragma unroll
                    for (int e = 0; e < EV; ++e) b[q * EV + e] = tfma(g[e], nxc, b[q * EV + e]);
                }
            }
        }
    }
    const T lo = nonneg ? T(0) : -INFINITY;
    const T hi = ub_cd > T(0) ? ub_cd : INFINITY;
    const bool check = tol > T(0);
    const T inv_k = T(1) / static_cast<T>(k);
    bool active = inb;
    int nsweep = 0;
    GV g0[NV];                                   // this lane's rows of G(:,0), refreshed by the last coordinate
#pragma unroll
    for (int q = 0; q < NV; ++q) g0[q] = *reinterpret_cast<const GV*>(Gs + sub_off + q * EV);
    for (int it = 0; it < maxit; ++it) {
        if (!__any(active)) break;
        nsweep += active ? 1 : 0;
        T tol_sum = T(0);
        CdGroupStep<T, KP, LPC, EXACT, 0>::template run<GV, NV>(b, x, g0, Gs, diag, invs, sub_off, l1_cd, l2_cd, lo, hi,
                                                                active, check, tol_sum);
        if (check && active && tol_sum * inv_k < tol) active = false;
    }
    if (inb) {
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            const int row = sub_off + r;
            if (row < k) {
                T v = x[r];
                if (ub_post > T(0)) v = v < ub_post ? v : ub_post;
                xj[row] = v;
            }
        }
        if (sweeps && sub == 0) sweeps[j] = nsweep;
    }
    cd_stats_add(stats, (inb && sub == 0) ? nsweep : 0, (inb && sub == 0) ? 1 : 0);
}

// ---------------------------------------------------------------------------
// CD NNLS, one WAVEFRONT per column with active-coordinate ballot skipping ("wave" variant).
// Same reference routine.  Lane r owns coordinates r, r+64, ... (VPL = KP/64 of them): its residual
// b_r, iterate x_r and 1/G_rr stay in registers; G sits in LDS ([i][r], conflict-free column reads).
// The reference visits coordinates in order and most visits are no-ops (x_i = 0 and the step would
// make it negative, or a zero step).  Whether coordinate r is a no-op depends only on the current
// residual, and the residual only changes at an effective step, so all lanes evaluate their own
// coordinate in parallel, a ballot + s_ff1 finds the next coordinate >= cur that really moves, and
// everything before it is skipped exactly.  Cost is per EFFECTIVE step, which is what makes this
// variant win when the solution is sparse and for ranks the lane variant cannot hold in registers.
// Persistent blocks: each wave strides over columns, G is staged into LDS once per block.
// ---------------------------------------------------------------------------
// GLDS = false (ranks above 128): the padded Gram stays in global memory (256 KiB in fp32 at KP = 256: L2-resident) and the
// moving coordinate's column is read coalesced from there -- the correct-if-slower path for any rank the MFMA tiles do
// not cover (reference gpu/batch_nnls.cuh:325-378 hands such ranks to a parallel general kernel as well).
template <class T, int KP, bool EXACT, bool GLDS = true>
__global__ __launch_bounds__(256) void cd_wave_kernel(const T* __restrict__ Gp,
                                                       const T* __restrict__ invd,
                                                       const T* __restrict__ B, T* __restrict__ X,
                                                       int k, int64_t ncols, T l1_pre, int warm,
                                                       int zero_init, T l1_cd, T l2_cd, int nonneg,
                                                       int maxit, T tol, T ub_cd, T ub_post,
                                                      int* __restrict__ sweeps, const int* __restrict__ order) {
    constexpr int VPL = KP / 64;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const T* Gs = Gp;
    if constexpr (GLDS) {
        T* Gl = reinterpret_cast<T*>(smem_raw);
        for (int e = threadIdx.x; e < KP * KP; e += blockDim.x) Gl[e] = Gp[e];
        __syncthreads();
        Gs = Gl;
    }
    const int lane = threadIdx.x & 63;
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const bool has_upper = ub_cd > T(0);
    const bool check = tol > T(0);
    const T inv_k = T(1) / static_cast<T>(k);
    T ginv[VPL], gdiag[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) { ginv[v] = invd[lane + 64 * v]; gdiag[v] = Gs[(lane + 64 * v) * KP + lane + 64 * v]; }

    for (int64_t sj = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); sj < ncols; sj += nwaves) {
        const int64_t j = order ? order[sj] : sj;
        T b[VPL], x[VPL];
        const T* bj = B + j * (int64_t)k;
        T* xj = X + j * (int64_t)k;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const int f = lane + 64 * v;
            b[v] = f < k ? bj[f] : T(0);
            x[v] = (f < k && !zero_init) ? xj[f] : T(0);
            if (l1_pre != T(0) && f < k) b[v] -= l1_pre;
        }
        if (warm) {  // b -= G x: broadcast x_c, column c of G from LDS
            for (int c = 0; c < k; ++c) {
                const T xc = __shfl(x[c >> 6], c & 63, 64);
                if (xc != T(0)) {
#pragma unroll
                    for (int v = 0; v < VPL; ++v) b[v] = tfma(-Gs[c * KP + lane + 64 * v], xc, b[v]);
                }
            }
        }
        int nsweep = 0;
        for (int it = 0; it < maxit; ++it) {
            ++nsweep;
            T tol_sum = T(0);
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                int cur = 0;  // next coordinate (within this 64-slab) the sweep has not visited
                while (true) {
                    T diff;
                    if constexpr (EXACT) diff = b[v] / gdiag[v];
                    else diff = b[v] * ginv[v];
                    if (l1_cd != T(0)) diff -= l1_cd;
                    if (l2_cd != T(0)) diff += l2_cd * x[v];
                    const T nv = x[v] + diff;
                    T ad = diff, nx = nv;
                    if (nonneg && nv < T(0)) { ad = -x[v]; nx = T(0); }
                    else if (has_upper && nv > ub_cd) { ad = ub_cd - x[v]; nx = ub_cd; }
                    const bool moves = (ginv[v] > T(0)) && (ad != T(0)) && (lane >= cur);
                    const unsigned long long mask = __ballot(moves);
                    if (mask == 0ull) break;
                    const int i = __builtin_ctzll(mask);   // first coordinate >= cur that moves
                    const T ad_i = __shfl(ad, i, 64);
                    const T nx_i = __shfl(nx, i, 64);
                    if (lane == i) x[v] = nx_i;
                    if (check) {
                        if constexpr (EXACT) tol_sum += tabs(ad_i) / (tabs(nx_i) + T(1e-15));
                        else tol_sum += fast_div<T>(tabs(ad_i), tabs(nx_i) + T(1e-15));
                    }
                    const T* gc = Gs + (i + 64 * v) * KP;
#pragma unroll
                    for (int u = 0; u < VPL; ++u) b[u] = tfma(-gc[lane + 64 * u], ad_i, b[u]);
                    cur = i + 1;
                    if (cur >= 64) break;
                }
            }
            if (check && tol_sum * inv_k < tol) break;
        }
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const int f = lane + 64 * v;
            if (f < k) {
                T val = x[v];
                if (ub_post > T(0)) val = val < ub_post ? val : ub_post;
                xj[f] = val;
            }
        }
        if (sweeps && lane == 0) sweeps[j] = nsweep;
    }
}

// ---------------------------------------------------------------------------
// Cholesky of the (padded) Gram, one wavefront: right-looking, column by column, in LDS.
// Restates Eigen::LLT as used at reference primitives/cpu/fused_nnls.hpp:185.
// Output L (KP x KP, lower, column-major) and invl[i] = 1/L_ii.
// ---------------------------------------------------------------------------
template <class T, int KP>
__global__ __launch_bounds__(64) void chol_factor_kernel(const T* __restrict__ Gp, T* __restrict__ L,
                                                          T* __restrict__ invl) {
    __shared__ T A[KP * KP];
    const int lane = threadIdx.x;
    for (int e = lane; e < KP * KP; e += 64) A[e] = Gp[e];
    __syncthreads();
    for (int j = 0; j < KP; ++j) {
        // left-looking: column j minus contributions of previous columns, rows i >= j
        for (int i = j + lane; i < KP; i += 64) {
            T s = A[j * KP + i];
            for (int p = 0; p < j; ++p) s -= A[p * KP + i] * A[p * KP + j];
            A[j * KP + i] = s;
        }
        __syncthreads();
        T djj = A[j * KP + j];
        if (!(djj > T(0))) djj = tabs(djj) + T(1e-30);
        const T ljj = sqrt(djj);
        __syncthreads();
        for (int i = j + lane; i < KP; i += 64) A[j * KP + i] = (i == j) ? ljj : A[j * KP + i] / ljj;
        __syncthreads();
    }
    for (int e = lane; e < KP * KP; e += 64) {
        const int i = e % KP, j = e / KP;
        L[e] = i >= j ? A[e] : T(0);
    }
    for (int i = lane; i < KP; i += 64) invl[i] = T(1) / A[i * KP + i];
}

// ---------------------------------------------------------------------------
// Cholesky solve + clip, one lane per column (reference fused_nnls.hpp:200-218):
// x = L^-T L^-1 (b - l1); x = max(x, 0) if nonneg; x = min(x, ub) if ub > 0.
// L is wave-uniform (SGPR operands), b per lane.  Divisions as the reference's llt.solve.
// ---------------------------------------------------------------------------
template <class T, int KP>
__global__ __launch_bounds__(64) void chol_solve_kernel(const T* __restrict__ L,
                                                         const T* __restrict__ B, T* __restrict__ X,
                                                         int k, int64_t ncols, T l1_pre, int nonneg,
                                                         T ub_post) {
    const int64_t j = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (j >= ncols) return;
    T x[KP];
    const T* bj = B + j * (int64_t)k;
#pragma unroll
    for (int i = 0; i < KP; ++i) x[i] = i < k ? bj[i] : T(0);
    if (l1_pre != T(0)) {
#pragma unroll
        for (int i = 0; i < KP; ++i)
            if (i < k) x[i] -= l1_pre;
    }
#pragma unroll
    for (int i = 0; i < KP; ++i) {
        T t = x[i];
#pragma unroll
        for (int p = 0; p < i; ++p) t = tfma(-L[p * KP + i], x[p], t);
        x[i] = t / L[i * KP + i];
    }
#pragma unroll
    for (int i = KP - 1; i >= 0; --i) {
        T t = x[i];
#pragma unroll
        for (int p = i + 1; p < KP; ++p) t = tfma(-L[i * KP + p], x[p], t);
        x[i] = t / L[i * KP + i];
    }
    T* xj = X + j * (int64_t)k;
#pragma unroll
    for (int i = 0; i < KP; ++i)
        if (i < k) {
            T v = x[i];
            if (nonneg) v = v > T(0) ? v : T(0);
            if (ub_post > T(0)) v = v < ub_post ? v : ub_post;
            xj[i] = v;
        }
}

// ---------------------------------------------------------------------------
// Scaling (reference nmf/variant_helpers.hpp:286-305): row sums of |x| or x^2, two deterministic
// passes; then d = (sqrt) + 1e-15 and X(i,:) /= d_i.
// ---------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ void row_norm_partial_body(const T* __restrict__ X, int k, int64_t ncols, int norm_type,
                                                      T* __restrict__ partial, const unsigned bid, const unsigned nb) {
    // thread t handles feature f = t % kp2 (kp2 = pow2 >= k, <= 256) and column slot t / kp2
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* sh = reinterpret_cast<T*>(smem_raw);
    int kp2 = 1;
    while (kp2 < k) kp2 <<= 1;
    if (kp2 > 256) kp2 = 256;  // k > 256 handled by the feature loop below
    const int slots = 256 / kp2;
    const int f0 = threadIdx.x % kp2, slot = threadIdx.x / kp2;
    const int64_t per = (ncols + nb - 1) / nb;
    const int64_t c0 = (int64_t)bid * per;
    const int64_t c1 = c0 + per < ncols ? c0 + per : ncols;
    for (int f = f0; f < k; f += kp2) {
        T acc = 0;
        for (int64_t c = c0 + slot; c < c1; c += slots) {
            const T v = X[c * (int64_t)k + f];
            acc += norm_type == 0 ? tabs(v) : (norm_type == 3 ? v : v * v);   // 3 = plain sum (H.rowwise().sum())
        }
        sh[threadIdx.x] = acc;
        __syncthreads();
        if (slot == 0) {
            T s = acc;
            for (int q = 1; q < slots; ++q) s += sh[q * kp2 + f0];
            partial[(int64_t)bid * k + f] = s;
        }
        __syncthreads();
    }
}
template <class T>
__global__ __launch_bounds__(256) void row_norm_partial(const T* __restrict__ X, int k, int64_t ncols,
                                                         int norm_type, T* __restrict__ partial) {
    row_norm_partial_body<T>(X, k, ncols, norm_type, partial, blockIdx.x, gridDim.x);
}
// Vectorised form for k % VEC == 0 (16-byte loads, four independent accumulators per lane: the scalar kernel above is
// one dependent 4-byte load chain per thread and runs at ~1.4 TB/s).  Same partial layout, fixed summation order.
template <class T, int VEC>
__device__ __forceinline__ void row_norm_partial_vec_body(const T* __restrict__ X, int k, int64_t ncols, int norm_type,
                                                          T* __restrict__ partial, const unsigned bid, const unsigned nb) {
    typedef typename VecT<T, VEC>::type V;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* sh = reinterpret_cast<T*>(smem_raw);          // slots x k
    const int lpc = k / VEC;                          // lanes per column (<= 256)
    const int slots = 256 / lpc;
    const int li = threadIdx.x % lpc, slot = threadIdx.x / lpc;
    const int64_t per = (ncols + nb - 1) / nb;
    const int64_t c0 = (int64_t)bid * per;
    const int64_t c1 = c0 + per < ncols ? c0 + per : ncols;
    T acc[4][VEC];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[u][e] = T(0);
    if (slot < slots) {
        for (int64_t c = c0 + slot; c < c1; c += 4 * slots) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t cc = c + (int64_t)u * slots;
                if (cc < c1) {
                    const V v = *reinterpret_cast<const V*>(X + cc * (int64_t)k + li * VEC);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const T x = v[e];
                        acc[u][e] += norm_type == 0 ? tabs(x) : (norm_type == 3 ? x : x * x);
                    }
                }
            }
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) sh[slot * k + li * VEC + e] = (acc[0][e] + acc[1][e]) + (acc[2][e] + acc[3][e]);
    }
    __syncthreads();
    for (int f = threadIdx.x; f < k; f += 256) {
        T s = T(0);
        for (int q = 0; q < slots; ++q) s += sh[q * k + f];
        partial[(int64_t)bid * k + f] = s;
    }
}
template <class T, int VEC>
__global__ __launch_bounds__(256) void row_norm_partial_vec(const T* __restrict__ X, int k, int64_t ncols,
                                                             int norm_type, T* __restrict__ partial) {
    row_norm_partial_vec_body<T, VEC>(X, k, ncols, norm_type, partial, blockIdx.x, gridDim.x);
}
// one wavefront per feature: lane-strided sums over the blocks, then a fixed xor-shuffle tree
template <class T>
__device__ __forceinline__ T row_norm_final_row(const T* partial, int nblk, int k, int f, int lane) {
    T s = 0;
    for (int b = lane; b < nblk; b += 64) s += partial[(int64_t)b * k + f];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += shfl_xor_t(s, off);
    return s;
}
template <class T>
__global__ __launch_bounds__(64) void row_norm_final(const T* __restrict__ partial, int nblk, int k, T* __restrict__ out) {
    const T s = row_norm_final_row<T>(partial, nblk, k, blockIdx.x, threadIdx.x);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}
template <class T>
__global__ void scaling_finalize(const T* __restrict__ sums, int k, int norm_type, T* __restrict__ d) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= k) return;
    if (norm_type == 2) { d[f] = T(1); return; }
    T s = sums[f];
    if (norm_type == 1) s = sqrt(s);
    d[f] = s + T(1e-15);
}
template <class T>
__global__ __launch_bounds__(256) void scale_rows(T* __restrict__ X, int k, int64_t total,
                                                   const T* __restrict__ d) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride)
        X[e] = X[e] / d[e % k];
}
// scaling_finalize + scale_rows in one launch (norm_type 0 / 1): every thread forms d_f from the row sums exactly as
// scaling_finalize does, block 0 also stores d
// 16-byte form of the same (k a multiple of the vector width, X 16-byte aligned): one vector load / store per VEC elements
template <class T, int VEC>
__device__ __forceinline__ void scale_rows_from_sums_vec_body(T* __restrict__ X, int k, int64_t total, const T* __restrict__ sums,
                                                              int norm_type, T* __restrict__ d, const unsigned bid, const unsigned nb) {
    typedef typename VecT<T, VEC>::type V;
    const int64_t stride = (int64_t)nb * blockDim.x;
    if (bid == 0)
        for (int f = threadIdx.x; f < k; f += blockDim.x) {
            T s = sums[f];
            if (norm_type == 1) s = sqrt(s);
            d[f] = s + T(1e-15);
        }
    const int64_t nvec = total / VEC;
    for (int64_t e = (int64_t)bid * blockDim.x + threadIdx.x; e < nvec; e += stride) {
        const int f0 = (int)((e * VEC) % k);
        V x = reinterpret_cast<V*>(X)[e];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            T s = sums[f0 + v];
            if (norm_type == 1) s = sqrt(s);
            x[v] = x[v] / (s + T(1e-15));
        }
        reinterpret_cast<V*>(X)[e] = x;
    }
}
template <class T, int VEC>
__global__ __launch_bounds__(256) void scale_rows_from_sums_vec(T* __restrict__ X, int k, int64_t total, const T* __restrict__ sums,
                                                                int norm_type, T* __restrict__ d) {
    scale_rows_from_sums_vec_body<T, VEC>(X, k, total, sums, norm_type, d, blockIdx.x, gridDim.x);
}
template <class T>
__device__ __forceinline__ void scale_rows_from_sums_body(T* __restrict__ X, int k, int64_t total, const T* __restrict__ sums,
                                                          int norm_type, T* __restrict__ d, const unsigned bid, const unsigned nb) {
    const int64_t stride = (int64_t)nb * blockDim.x;
    if (bid == 0)
        for (int f = threadIdx.x; f < k; f += blockDim.x) {
            T s = sums[f];
            if (norm_type == 1) s = sqrt(s);
            d[f] = s + T(1e-15);
        }
    for (int64_t e = (int64_t)bid * blockDim.x + threadIdx.x; e < total; e += stride) {
        T s = sums[e % k];
        if (norm_type == 1) s = sqrt(s);
        X[e] = X[e] / (s + T(1e-15));
    }
}
template <class T>
__global__ __launch_bounds__(256) void scale_rows_from_sums(T* __restrict__ X, int k, int64_t total, const T* __restrict__ sums,
                                                            int norm_type, T* __restrict__ d) {
    scale_rows_from_sums_body<T>(X, k, total, sums, norm_type, d, blockIdx.x, gridDim.x);
}

// Graph regularisation (features/graph_reg.hpp:38-50):  G += lambda * (F L) F^T  with FL = F L formed by the SpMM kernel.
// cross_gram_partial: per block, P[b*k + a] = sum over its columns j of X(a, j) Y(b, j)  (k <= KMAX = 64 or 128; tiles of 32
// columns in LDS); cross_gram_axpy: G += lambda * (fixed-order sum of the block partials).
template <class T, int KMAX>
__global__ __launch_bounds__(256) void cross_gram_partial(const T* __restrict__ X, const T* __restrict__ Y, int k, int64_t ncols,
                                                           T* __restrict__ partial) {
    constexpr int NA = KMAX * KMAX / 256;          // accumulators per thread
    __shared__ T xs[32 * KMAX], ys[32 * KMAX];
    const int64_t per = (ncols + gridDim.x - 1) / gridDim.x;
    const int64_t c0 = (int64_t)blockIdx.x * per;
    const int64_t c1 = c0 + per < ncols ? c0 + per : ncols;
    T acc[NA];
#pragma unroll
    for (int u = 0; u < NA; ++u) acc[u] = T(0);
    for (int64_t cb = c0; cb < c1; cb += 32) {
        const int nc = (int)(c1 - cb < 32 ? c1 - cb : 32);
        for (int e = threadIdx.x; e < 32 * k; e += 256) {
            const int cc = e / k, f = e % k;
            xs[cc * KMAX + f] = cc < nc ? X[(cb + cc) * (int64_t)k + f] : T(0);
            ys[cc * KMAX + f] = cc < nc ? Y[(cb + cc) * (int64_t)k + f] : T(0);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int e = threadIdx.x + 256 * u;
            if (e < k * k) {
                const int a = e % k, b = e / k;
                T s = acc[u];
                for (int cc = 0; cc < 32; ++cc) s = tfma(xs[cc * KMAX + a], ys[cc * KMAX + b], s);
                acc[u] = s;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < NA; ++u) {
        const int e = threadIdx.x + 256 * u;
        if (e < k * k) partial[(int64_t)blockIdx.x * k * k + e] = acc[u];
    }
}
template <class T>
__global__ void cross_gram_axpy(const T* __restrict__ partial, int nblk, int kk, T lambda, T* __restrict__ G) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= kk) return;
    T s = T(0);
    for (int b = 0; b < nblk; ++b) s += partial[(int64_t)b * kk + e];
    G[e] += lambda * s;
}

// Y = diag(d) X (rows scaled UP; variant_helpers.hpp:265-272 apply_scaling) -- the projective H update's W_Td
template <class T>
__global__ __launch_bounds__(256) void mul_rows(const T* __restrict__ X, int k, int64_t total, const T* __restrict__ d,
                                                 T* __restrict__ Y) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) Y[e] = X[e] * d[e % k];
}

// ---------------------------------------------------------------------------
// fp64 reductions for the loss (reference primitives/primitives.hpp:100-115 trace_AtA;
// nmf/fit_cpu.hpp:1740-1753 cross term and recon norm).  Block partials then a fixed-order sum.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double block_sum_256(double v, double* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double s = 0;
    if (threadIdx.x == 0) s = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return s;
}
template <class T>
__global__ __launch_bounds__(256) void sumsq_partial(const T* __restrict__ x, int64_t len,
                                                      double* __restrict__ partial) {
    __shared__ double sh[4];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double acc = 0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += stride) {
        const double v = static_cast<double>(x[e]);
        acc += v * v;
    }
    const double s = block_sum_256(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
// cross partials: sum_e d[e%k] * W[e] * Bw[e]
template <class T>
__device__ __forceinline__ void cross_partial_body(const T* __restrict__ W, const T* __restrict__ Bw, const T* __restrict__ d, int k,
                                                   int64_t total, double* __restrict__ partial, const unsigned bid, const unsigned nb,
                                                   double* sh /* 4 doubles of LDS */) {
    const int64_t stride = (int64_t)nb * blockDim.x;
    double acc = 0;
    for (int64_t e = (int64_t)bid * blockDim.x + threadIdx.x; e < total; e += stride)
        acc += static_cast<double>(d[e % k]) * static_cast<double>(W[e]) * static_cast<double>(Bw[e]);
    const double s = block_sum_256(acc, sh);
    if (threadIdx.x == 0) partial[bid] = s;
}
template <class T>
__global__ __launch_bounds__(256) void cross_partial(const T* __restrict__ W, const T* __restrict__ Bw,
                                                      const T* __restrict__ d, int k, int64_t total,
                                                      double* __restrict__ partial) {
    __shared__ double sh[4];
    cross_partial_body<T>(W, Bw, d, k, total, partial, blockIdx.x, gridDim.x, sh);
}
static __global__ __launch_bounds__(256) void sum_partials(const double* __restrict__ partial, int n,
                                                     double* __restrict__ out) {
    __shared__ double sh[4];
    double acc = 0;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    const double s = block_sum_256(acc, sh);
    if (threadIdx.x == 0) out[0] = s;
}
// out[0] = trAtA - 2 cross + recon; out[1] = cross; out[2] = recon    (single block of 256)
template <class T>
__device__ __forceinline__ void loss_mse_final_body(const double* trAtA, const double* cross_part, int npart, const T* d, const T* Gwt,
                                                    const T* Gsaved, int k, double* out, double* sh /* 4 doubles of LDS */) {
    double acc = 0;
    for (int e = threadIdx.x; e < k * k; e += 256) {
        const int i = e % k, j = e / k;
        acc += static_cast<double>(d[i]) * static_cast<double>(d[j]) * static_cast<double>(Gwt[e]) *
               static_cast<double>(Gsaved[e]);
    }
    const double recon = block_sum_256(acc, sh);
    double cacc = 0;
    for (int i = threadIdx.x; i < npart; i += 256) cacc += cross_part[i];
    const double cross = block_sum_256(cacc, sh);
    if (threadIdx.x == 0) {
        out[0] = trAtA[0] - 2.0 * cross + recon;
        out[1] = cross;
        out[2] = recon;
    }
}
template <class T>
__global__ __launch_bounds__(256) void loss_mse_final(const double* __restrict__ trAtA,
                                                       const double* __restrict__ cross_part, int npart,
                                                       const T* __restrict__ d, const T* __restrict__ Gwt,
                                                       const T* __restrict__ Gsaved, int k,
                                                       double* __restrict__ out) {
    __shared__ double sh[4];
    loss_mse_final_body<T>(trAtA, cross_part, npart, d, Gwt, Gsaved, k, out, sh);
}

// ---------------------------------------------------------------------------
// Explicit-mask per-column NNLS (reference nmf/masked_nnls.hpp:96-154 / 177-242).
// One wavefront per column; lane r owns feature r (k <= 64).  The per-column Gram
//   G_loc = G_full - sum_{r in masked(j)} f_r f_r^T  (+ l2 on the diagonal)
// lives in LDS (one k x k tile per wave); b skips masked rows of A (two-pointer merge: both row
// lists are sorted).  Solve: CD exactly as the wave variant above (sequential visit, in-order), or
// in-LDS Cholesky + clip for solver_mode 1.
// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
// Cross-validation half-update (reference nmf/fit_cv.hpp:420-478 / :591-830, nmf/cv_detail.hpp:66-85,304-405,
// nmf/speckled_cv.hpp): speckled holdout mask defined by SplitMix64::hash(seed, i, j) < UINT64_MAX / inv_prob
// (rng/rng.hpp:129-170), no mask matrix.  One wavefront per column j of D (D = A on the H side, A^T with
// transposed = 1 on the W side; the mask is always asked in the coordinates of A):
//   b       = sum over the column's TRAIN nonzeros  a F(row, :)
//   G_local = G - sum over the column's TEST rows  F(row, :) F(row, :)^T   (mask_zeros: held-out nonzeros only;
//             otherwise every held-out row, zeros included -- the 64 lanes hash 64 rows at a time)
//   x       = cholesky_clip(G_local, b - L1) or CD(G_local, b, x; L1 inside, cd_maxit sweeps, no tolerance), started from
//             the current column without a warm-start correction of b, exactly as the reference does.
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long cv_hash_dev(unsigned long long seed, unsigned i, unsigned j) {
    unsigned long long h = seed + (unsigned long long)i * 0x9e3779b97f4a7c15ULL + (unsigned long long)j * 0x6c62272e07bb0142ULL;
    h = (h ^ (h >> 30)) * 0xbf58476d1ce4e5b9ULL;
    h = (h ^ (h >> 27)) * 0x94d049bb133111ebULL;
    return h ^ (h >> 31);
}

// user mask of a CV fit (fit_cv.hpp:327-331; cv_detail.hpp:408-415 is_user_masked_h): pattern CSC in the orientation of the CSC the
// kernel walks, rows ascending inside a column; NULL = none.  Bisection: masks are sparse and the question is asked per entry.
__device__ __forceinline__ bool cv_user_masked(const int* __restrict__ mp, const int* __restrict__ mi, int64_t j, int row) {
    if (!mp) return false;
    int lo = mp[j];
    const int end = mp[j + 1];
    int hi = end;
    while (lo < hi) { const int mid = lo + ((hi - lo) >> 1); if (mi[mid] < row) lo = mid + 1; else hi = mid; }
    return lo < end && mi[lo] == row;
}
// is `row` one of the stored rows of column [ts, te) of a CSC with ascending rows?
__device__ __forceinline__ bool cv_row_stored(const int* __restrict__ rowidx, int ts, int te, int row) {
    int lo = ts, hi = te;
    while (lo < hi) { const int mid = lo + ((hi - lo) >> 1); if (rowidx[mid] < row) lo = mid + 1; else hi = mid; }
    return lo < te && rowidx[lo] == row;
}

// (mp, mi: optional user mask, cv_detail.hpp:433-505 -- a masked row that is not already a test row leaves b and joins the rows of
// the Gram correction, whether its entry is a nonzero or not)
template <class T, int KP>   // KP in {32, 64}
__global__ __launch_bounds__(256) void cv_solve_kernel(
    const int* __restrict__ colptr, const int* __restrict__ rowidx, const T* __restrict__ vals, int64_t ncols, int nrows,
    const T* __restrict__ F, const T* __restrict__ Gfull, T* __restrict__ X, int k, unsigned long long seed,
    unsigned long long threshold, int mask_zeros, int transposed, T l1, int nonneg, int maxit, int solver_mode,
    const int* __restrict__ mp = nullptr, const int* __restrict__ mi = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    T* Gl = reinterpret_cast<T*>(smem_raw) + (size_t)wave * KP * KP;   // [c][r]
    const int64_t j = (int64_t)blockIdx.x * 4 + wave;
    if (j >= ncols) return;
    const bool fok = lane < k;
    const bool lin = lane < KP;
    const int ll = lin ? lane : 0;
    for (int c = 0; c < KP; ++c) {
        T v = (fok && c < k) ? Gfull[(int64_t)c * k + lane] : (c == lane ? T(1) : T(0));
        if (lin) Gl[c * KP + lane] = v;
    }
    const unsigned col = (unsigned)j;
    // train right-hand side (and, with mask_zeros, the Gram correction of the held-out nonzeros)
    T b = T(0);
    for (int t = colptr[j]; t < colptr[j + 1]; ++t) {
        const int row = rowidx[t];
        const bool held = (transposed ? cv_hash_dev(seed, col, (unsigned)row) : cv_hash_dev(seed, (unsigned)row, col)) < threshold;
        const T fr = fok ? F[(int64_t)row * k + lane] : T(0);
        if (!held) {
            if (!cv_user_masked(mp, mi, j, row)) b = tfma(vals[t], fr, b);
        } else if (mask_zeros) {
            for (int c = 0; c < k; ++c) {
                const T fc = __shfl(fr, c, 64);
                if (lin) Gl[c * KP + lane] -= fr * fc;
            }
        }
    }
    if (!mask_zeros) {      // every held-out row of this column, zeros included
        for (int r0 = 0; r0 < nrows; r0 += 64) {
            const int r = r0 + lane;
            const bool held = r < nrows &&
                (transposed ? cv_hash_dev(seed, col, (unsigned)r) : cv_hash_dev(seed, (unsigned)r, col)) < threshold;
            unsigned long long m = __ballot(held);
            while (m) {
                const int bit = __builtin_ctzll(m);
                m &= m - 1;
                const int row = r0 + bit;
                const T fr = fok ? F[(int64_t)row * k + lane] : T(0);
                for (int c = 0; c < k; ++c) {
                    const T fc = __shfl(fr, c, 64);
                    if (lin) Gl[c * KP + lane] -= fr * fc;
                }
            }
        }
    }
    if (mp) {               // user-masked rows that are not test rows: out of the Gram as well
        const int ts = colptr[j], te = colptr[j + 1];
        for (int t = mp[j]; t < mp[j + 1]; ++t) {
            const int row = mi[t];
            const bool held = (transposed ? cv_hash_dev(seed, col, (unsigned)row) : cv_hash_dev(seed, (unsigned)row, col)) < threshold;
            if (held && (!mask_zeros || cv_row_stored(rowidx, ts, te, row))) continue;      // already corrected above
            const T fr = fok ? F[(int64_t)row * k + lane] : T(0);
            for (int c = 0; c < k; ++c) {
                const T fc = __shfl(fr, c, 64);
                if (lin) Gl[c * KP + lane] -= fr * fc;
            }
        }
    }
    RK_WAVE_SYNC();
    T x = fok ? X[j * (int64_t)k + lane] : T(0);
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
        T y = b;
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
    } else {
        const T gd = Gl[ll * KP + ll];
        // static coordinate sweeps (cd_static_sweeps, kernels.hip.h): the lane's Gram column read from the wave's LDS tile at
        // compile-time offsets
        cd_static_sweeps<T, KP>(b, x, gd, fok, l1, nonneg, maxit, [&](auto IC) { return Gl[decltype(IC)::value * KP + ll]; });
    }
    if (fok) X[j * (int64_t)k + lane] = x;
}

// fp32, k <= 32 (k % 4 == 0): the same half-update with the Gram correction on the MATRIX cores.  Held-out rows are
// collected in a small LDS queue (ballot + prefix popcount); every 32 of them are gathered with 16-byte loads (lane =
// row t, feature half hh), parked in LDS and applied as 16 rank-2 updates  G_local -= f f^T  with
// v_mfma_f32_32x32x2_f32 (A operand = -f, B operand = f; accumulator tile initialised with G) -- against 32 shuffles
// and 32 LDS read-modify-writes per held-out row in cv_solve_kernel.  With zeros held out a column of a 20 000-row
// matrix has ~2 000 such rows.
static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void cv_solve_mfma32_kernel(
    const int* __restrict__ colptr, const int* __restrict__ rowidx, const float* __restrict__ vals, int64_t ncols, int nrows,
    const float* __restrict__ F, const float* __restrict__ Gfull, float* __restrict__ X, int k, unsigned long long seed,
    unsigned long long threshold, int mask_zeros, int transposed, float l1, int nonneg, int maxit, int solver_mode) {
    constexpr int KP = 32, FS = 36, QCAP = 96;
    constexpr int WAVE_FLOATS = 32 * FS + QCAP;       // staged rows (aliased by G_local afterwards) | row queue
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* Fst = reinterpret_cast<float*>(smem_raw) + (size_t)wave * WAVE_FLOATS;
    int* hq = reinterpret_cast<int*>(Fst + 32 * FS);
    float* Gl = Fst;                                  // [c][r]
    const int64_t j = (int64_t)blockIdx.x * 4 + wave;
    if (j >= ncols) return;
    const int r = lane & 31, hh = lane >> 5;
    const bool fok = lane < k;
    const bool lin = lane < KP;
    const int ll = lin ? lane : 0;
    const unsigned col = (unsigned)j;
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int gi = (v & 3) + 8 * (v >> 2) + 4 * hh, gj = r;
        acc[v] = (gi < k && gj < k) ? Gfull[(int64_t)gj * k + gi] : (gi == gj ? 1.f : 0.f);
    }
    int qn = 0;                                       // rows waiting in hq (wave-uniform)
    // apply the first `cnt` (<= 32) queued rows to the accumulator tile
    auto flush = [&](int cnt) {
        const bool ok = r < cnt;
        const int row = ok ? hq[r] : 0;
        const float* fsrc = F + (int64_t)row * k + 16 * hh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c0 = 16 * hh + 4 * q;
            const float4 v = (ok && c0 < k) ? *reinterpret_cast<const float4*>(fsrc + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(Fst + r * FS + c0) = v;
        }
        __builtin_amdgcn_wave_barrier();
        const int nst = (cnt + 1) >> 1;
#pragma unroll 4
        for (int s2 = 0; s2 < nst; ++s2) {
            const float fv = Fst[(2 * s2 + hh) * FS + r];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(-fv, fv, acc, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    };
    // append the rows flagged in `held` (one per lane, row index `row`) to the queue; flush while >= 32 are waiting
    auto push = [&](bool held, int row) {
        const unsigned long long m = __ballot(held);
        if (m == 0ull) return;
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        if (held) hq[qn + rank] = row;
        qn += __popcll(m);
        __builtin_amdgcn_wave_barrier();
        while (qn >= 32) {
            flush(32);
            const int rest = qn - 32;
            const int moved = lane < rest ? hq[32 + lane] : 0;      // rest <= 63
            __builtin_amdgcn_wave_barrier();
            if (lane < rest) hq[lane] = moved;
            __builtin_amdgcn_wave_barrier();
            qn = rest;
        }
    };
    float b = 0.f;
    for (int t0 = colptr[j]; t0 < colptr[j + 1]; t0 += 64) {       // 64 nonzeros per step: lane-parallel hashing
        const int t = t0 + lane;
        const bool valid = t < colptr[j + 1];
        const int row = valid ? rowidx[t] : 0;
        const float a = valid ? vals[t] : 0.f;
        const bool held = valid && (transposed ? cv_hash_dev(seed, col, (unsigned)row) : cv_hash_dev(seed, (unsigned)row, col)) < threshold;
        // train right-hand side: lane = feature again, one nonzero at a time
        unsigned long long tm = __ballot(valid && !held);
        while (tm) {
            const int bit = __builtin_ctzll(tm);
            tm &= tm - 1;
            const int rw = __builtin_amdgcn_readlane(row, bit);
            const float av = lane_value(a, bit);
            if (fok) b = tfma(av, F[(int64_t)rw * k + lane], b);
        }
        if (mask_zeros) push(held, row);
    }
    if (!mask_zeros) {
        for (int r0 = 0; r0 < nrows; r0 += 64) {
            const int rw = r0 + lane;
            const bool held = rw < nrows &&
                (transposed ? cv_hash_dev(seed, col, (unsigned)rw) : cv_hash_dev(seed, (unsigned)rw, col)) < threshold;
            push(held, rw);
        }
    }
    if (qn > 0) flush(qn);
    // park G_local in LDS ([c][r]; symmetric)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int gi = (v & 3) + 8 * (v >> 2) + 4 * hh;
        Gl[gi * KP + r] = acc[v];
    }
    __builtin_amdgcn_wave_barrier();
    float x = fok ? X[j * (int64_t)k + lane] : 0.f;
    if (solver_mode == 1) {
        if (l1 > 0.f && fok) b -= l1;
        for (int c = 0; c < KP; ++c) {
            float s = Gl[c * KP + ll];
            for (int p = 0; p < c; ++p) s -= Gl[p * KP + ll] * Gl[p * KP + c];
            float dcc = __shfl(s, c, 64);
            if (!(dcc > 0.f)) dcc = tabs(dcc) + 1e-30f;
            const float lcc = sqrt(dcc);
            if (lin) Gl[c * KP + lane] = lane == c ? lcc : (lane > c ? s / lcc : 0.f);
            RK_WAVE_SYNC();
        }
        float y = b;
        for (int i = 0; i < k; ++i) {
            const float yi = __shfl(y, i, 64) / Gl[i * KP + i];
            if (lane == i) y = yi;
            else if (lane > i) y -= Gl[i * KP + ll] * yi;
        }
        for (int i = k - 1; i >= 0; --i) {
            const float xi = __shfl(y, i, 64) / Gl[i * KP + i];
            if (lane == i) y = xi;
            else if (lane < i) y -= Gl[ll * KP + i] * xi;
        }
        x = y;
        if (nonneg) x = x > 0.f ? x : 0.f;
    } else {
        const float gd = Gl[ll * KP + ll];
        // the lane's column of the corrected Gram in registers: the sweep picks row i with a wave-uniform register-indexed
        // move instead of an LDS read on the dependent chain of every step (as irls_nb_mfma32_kernel does)
        typedef float f32x32 __attribute__((ext_vector_type(32)));
        f32x32 gcol;
#pragma unroll
        for (int c = 0; c < KP; ++c) gcol[c] = Gl[c * KP + ll];
        {
            float gg[KP];
#pragma unroll
            for (int c = 0; c < KP; ++c) gg[c] = gcol[c];
            cd_static_sweeps_scaled_f32<KP>(b, x, gd, fok, l1, nonneg, maxit, gg);
        }
    }
    if (fok) X[j * (int64_t)k + lane] = x;
}

// fp32, 32 < k <= 64 (k % 4 == 0): the same on a 2 x 2 grid of 32 x 32 accumulator tiles (three MFMAs per pair of held-out
// rows), lane = feature over the whole wave in the solve; 16 KiB of LDS per wave for G_local (two blocks per CU).
static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void cv_solve_mfma32x2_kernel(
    const int* __restrict__ colptr, const int* __restrict__ rowidx, const float* __restrict__ vals, int64_t ncols, int nrows,
    const float* __restrict__ F, const float* __restrict__ Gfull, float* __restrict__ X, int k, unsigned long long seed,
    unsigned long long threshold, int mask_zeros, int transposed, float l1, int nonneg, int maxit, int solver_mode) {
    constexpr int KP = 64, FS = 68, QCAP = 96;
    constexpr int WAVE_FLOATS = KP * KP + QCAP;       // G_local (its head doubles as the 32 x FS staging area) | row queue
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* Fst = reinterpret_cast<float*>(smem_raw) + (size_t)wave * WAVE_FLOATS;
    int* hq = reinterpret_cast<int*>(Fst + KP * KP);
    float* Gl = Fst;                                  // [c][r]
    const int64_t j = (int64_t)blockIdx.x * 4 + wave;
    if (j >= ncols) return;
    const int r = lane & 31, hh = lane >> 5;
    const bool fok = lane < k;
    const bool lin = lane < KP;
    const int ll = lin ? lane : 0;
    const unsigned col = (unsigned)j;
    // 2 x 2 tiles of 32 x 32; the lower-left one is the transpose of the upper-right one and is never computed
    f32x16 a00, a01, a11;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int gi = (v & 3) + 8 * (v >> 2) + 4 * hh, gj = r;
        auto base = [&](int i2, int j2) { return (i2 < k && j2 < k) ? Gfull[(int64_t)j2 * k + i2] : (i2 == j2 ? 1.f : 0.f); };
        a00[v] = base(gi, gj); a01[v] = base(gi, gj + 32); a11[v] = base(gi + 32, gj + 32);
    }
    int qn = 0;                                       // rows waiting in hq (wave-uniform)
    // apply the first `cnt` (<= 32) queued rows to the accumulator tile
    auto flush = [&](int cnt) {
        const bool ok = r < cnt;
        const int row = ok ? hq[r] : 0;
        const float* fsrc = F + (int64_t)row * k + 32 * hh;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int c0 = 32 * hh + 4 * q;
            const float4 v = (ok && c0 < k) ? *reinterpret_cast<const float4*>(fsrc + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(Fst + r * FS + c0) = v;
        }
        __builtin_amdgcn_wave_barrier();
        const int nst = (cnt + 1) >> 1;
#pragma unroll 2
        for (int s2 = 0; s2 < nst; ++s2) {
            const float f0 = Fst[(2 * s2 + hh) * FS + r], f1 = Fst[(2 * s2 + hh) * FS + 32 + r];
            a00 = __builtin_amdgcn_mfma_f32_32x32x2f32(-f0, f0, a00, 0, 0, 0);
            a01 = __builtin_amdgcn_mfma_f32_32x32x2f32(-f0, f1, a01, 0, 0, 0);
            a11 = __builtin_amdgcn_mfma_f32_32x32x2f32(-f1, f1, a11, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    };
    // append the rows flagged in `held` (one per lane, row index `row`) to the queue; flush while >= 32 are waiting
    auto push = [&](bool held, int row) {
        const unsigned long long m = __ballot(held);
        if (m == 0ull) return;
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        if (held) hq[qn + rank] = row;
        qn += __popcll(m);
        __builtin_amdgcn_wave_barrier();
        while (qn >= 32) {
            flush(32);
            const int rest = qn - 32;
            const int moved = lane < rest ? hq[32 + lane] : 0;      // rest <= 63
            __builtin_amdgcn_wave_barrier();
            if (lane < rest) hq[lane] = moved;
            __builtin_amdgcn_wave_barrier();
            qn = rest;
        }
    };
    float b = 0.f;
    for (int t0 = colptr[j]; t0 < colptr[j + 1]; t0 += 64) {       // 64 nonzeros per step: lane-parallel hashing
        const int t = t0 + lane;
        const bool valid = t < colptr[j + 1];
        const int row = valid ? rowidx[t] : 0;
        const float a = valid ? vals[t] : 0.f;
        const bool held = valid && (transposed ? cv_hash_dev(seed, col, (unsigned)row) : cv_hash_dev(seed, (unsigned)row, col)) < threshold;
        // train right-hand side: lane = feature again, one nonzero at a time
        unsigned long long tm = __ballot(valid && !held);
        while (tm) {
            const int bit = __builtin_ctzll(tm);
            tm &= tm - 1;
            const int rw = __builtin_amdgcn_readlane(row, bit);
            const float av = lane_value(a, bit);
            if (fok) b = tfma(av, F[(int64_t)rw * k + lane], b);
        }
        if (mask_zeros) push(held, row);
    }
    if (!mask_zeros) {
        for (int r0 = 0; r0 < nrows; r0 += 64) {
            const int rw = r0 + lane;
            const bool held = rw < nrows &&
                (transposed ? cv_hash_dev(seed, col, (unsigned)rw) : cv_hash_dev(seed, (unsigned)rw, col)) < threshold;
            push(held, rw);
        }
    }
    if (qn > 0) flush(qn);
    // park G_local in LDS ([c][r]; symmetric)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int gi = (v & 3) + 8 * (v >> 2) + 4 * hh;
        Gl[gi * KP + r] = a00[v];
        Gl[(gi + 32) * KP + 32 + r] = a11[v];
        Gl[(32 + r) * KP + gi] = a01[v];                  // G(row gi, col 32 + r) and its mirror
        Gl[gi * KP + 32 + r] = a01[v];
    }
    __builtin_amdgcn_wave_barrier();
    float x = fok ? X[j * (int64_t)k + lane] : 0.f;
    if (solver_mode == 1) {
        if (l1 > 0.f && fok) b -= l1;
        for (int c = 0; c < KP; ++c) {
            float s = Gl[c * KP + ll];
            for (int p = 0; p < c; ++p) s -= Gl[p * KP + ll] * Gl[p * KP + c];
            float dcc = __shfl(s, c, 64);
            if (!(dcc > 0.f)) dcc = tabs(dcc) + 1e-30f;
            const float lcc = sqrt(dcc);
            if (lin) Gl[c * KP + lane] = lane == c ? lcc : (lane > c ? s / lcc : 0.f);
            RK_WAVE_SYNC();
        }
        float y = b;
        for (int i = 0; i < k; ++i) {
            const float yi = __shfl(y, i, 64) / Gl[i * KP + i];
            if (lane == i) y = yi;
            else if (lane > i) y -= Gl[i * KP + ll] * yi;
        }
        for (int i = k - 1; i >= 0; --i) {
            const float xi = __shfl(y, i, 64) / Gl[i * KP + i];
            if (lane == i) y = xi;
            else if (lane < i) y -= Gl[ll * KP + i] * xi;
        }
        x = y;
        if (nonneg) x = x > 0.f ? x : 0.f;
    } else {
        const float gd = Gl[ll * KP + ll];
        // the lane's column of the corrected Gram in registers: the sweep picks row i with a wave-uniform register-indexed
        // move instead of an LDS read on the dependent chain of every step (as irls_nb_mfma32_kernel does)
        typedef float f32x32 __attribute__((ext_vector_type(32)));      // (two halves: hipcc indexes 32-element vectors with
        f32x32 gcol0, gcol1;                                            //  s_set_gpr_idx, 64-element ones through scratch)
#pragma unroll
        for (int c = 0; c < 32; ++c) { gcol0[c] = Gl[c * KP + ll]; gcol1[c] = Gl[(32 + c) * KP + ll]; }
        {
            float gg[KP];
#pragma unroll
            for (int c = 0; c < 32; ++c) { gg[c] = gcol0[c]; gg[32 + c] = gcol1[c]; }
            cd_static_sweeps_scaled_f32<KP>(b, x, gd, fok, l1, nonneg, maxit, gg);
        }
    }
    if (fok) X[j * (int64_t)k + lane] = x;
}


// fp64, k <= 32 (k % 2 == 0): the same with v_mfma_f64_16x16x4_f64 (2 x 2 tiles of 16 x 16, four held-out rows per
// instruction step, C/D map col = lane&15, row = (lane>>4) + 4v).
static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void cv_solve_mfma64_kernel(
    const int* __restrict__ colptr, const int* __restrict__ rowidx, const double* __restrict__ vals, int64_t ncols, int nrows,
    const double* __restrict__ F, const double* __restrict__ Gfull, double* __restrict__ X, int k, unsigned long long seed,
    unsigned long long threshold, int mask_zeros, int transposed, double l1, int nonneg, int maxit, int solver_mode) {
    constexpr int KP = 32, FS = 34, QCAP = 96;
    constexpr int WAVE_DOUBLES = 32 * FS + QCAP / 2;   // staged rows (aliased by G_local afterwards) | row queue (ints)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* Fst = reinterpret_cast<double*>(smem_raw) + (size_t)wave * WAVE_DOUBLES;
    int* hq = reinterpret_cast<int*>(Fst + 32 * FS);
    double* Gl = Fst;                                  // [c][r]
    const int64_t j = (int64_t)blockIdx.x * 4 + wave;
    if (j >= ncols) return;
    const int r = lane & 31, hh = lane >> 5;
    const bool fok = lane < k;
    const bool lin = lane < KP;
    const int ll = lin ? lane : 0;
    const unsigned col = (unsigned)j;
    const int r16 = lane & 15, kk = lane >> 4;        // MFMA phase: feature slot r16, K-slot kk (16x16x4 f64)
    f64x4 acc[2][2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int gi = 16 * ti + kk + 4 * v, gj = 16 * tj + r16;
                acc[ti][tj][v] = (gi < k && gj < k) ? Gfull[(int64_t)gj * k + gi] : (gi == gj ? 1.0 : 0.0);
            }
    int qn = 0;                                       // rows waiting in hq (wave-uniform)
    // apply the first `cnt` (<= 32) queued rows to the accumulator tile
    auto flush = [&](int cnt) {
        const bool ok = r < cnt;
        const int row = ok ? hq[r] : 0;
        const double* fsrc = F + (int64_t)row * k + 16 * hh;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int c0 = 16 * hh + 2 * q;
            const double2 v = (ok && c0 < k) ? *reinterpret_cast<const double2*>(fsrc + 2 * q) : make_double2(0.0, 0.0);
            *reinterpret_cast<double2*>(Fst + r * FS + c0) = v;
        }
        __builtin_amdgcn_wave_barrier();
        const int nst = (cnt + 3) >> 2;
#pragma unroll 2
        for (int s4 = 0; s4 < nst; ++s4) {
            const int t = 4 * s4 + kk;
            const double f0 = Fst[t * FS + r16], f1 = Fst[t * FS + 16 + r16];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(-f0, f0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-f0, f1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(-f1, f0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-f1, f1, acc[1][1], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    };
    // append the rows flagged in `held` (one per lane, row index `row`) to the queue; flush while >= 32 are waiting
    auto push = [&](bool held, int row) {
        const unsigned long long m = __ballot(held);
        if (m == 0ull) return;
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        if (held) hq[qn + rank] = row;
        qn += __popcll(m);
        __builtin_amdgcn_wave_barrier();
        while (qn >= 32) {
            flush(32);
            const int rest = qn - 32;
            const int moved = lane < rest ? hq[32 + lane] : 0;      // rest <= 63
            __builtin_amdgcn_wave_barrier();
            if (lane < rest) hq[lane] = moved;
            __builtin_amdgcn_wave_barrier();
            qn = rest;
        }
    };
    double b = 0.0;
    for (int t0 = colptr[j]; t0 < colptr[j + 1]; t0 += 64) {       // 64 nonzeros per step: lane-parallel hashing
        const int t = t0 + lane;
        const bool valid = t < colptr[j + 1];
        const int row = valid ? rowidx[t] : 0;
        const double a = valid ? vals[t] : 0.0;
        const bool held = valid && (transposed ? cv_hash_dev(seed, col, (unsigned)row) : cv_hash_dev(seed, (unsigned)row, col)) < threshold;
        // train right-hand side: lane = feature again, one nonzero at a time
        unsigned long long tm = __ballot(valid && !held);
        while (tm) {
            const int bit = __builtin_ctzll(tm);
            tm &= tm - 1;
            const int rw = __builtin_amdgcn_readlane(row, bit);
            const double av = lane_value(a, bit);
            if (fok) b = tfma(av, F[(int64_t)rw * k + lane], b);
        }
        if (mask_zeros) push(held, row);
    }
    if (!mask_zeros) {
        for (int r0 = 0; r0 < nrows; r0 += 64) {
            const int rw = r0 + lane;
            const bool held = rw < nrows &&
                (transposed ? cv_hash_dev(seed, col, (unsigned)rw) : cv_hash_dev(seed, (unsigned)rw, col)) < threshold;
            push(held, rw);
        }
    }
    if (qn > 0) flush(qn);
    // park G_local in LDS ([c][r]; symmetric)
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int v = 0; v < 4; ++v) Gl[(16 * ti + kk + 4 * v) * KP + 16 * tj + r16] = acc[ti][tj][v];
    __builtin_amdgcn_wave_barrier();
    double x = fok ? X[j * (int64_t)k + lane] : 0.0;
    if (solver_mode == 1) {
        if (l1 > 0.0 && fok) b -= l1;
        for (int c = 0; c < KP; ++c) {
            double s = Gl[c * KP + ll];
            for (int p = 0; p < c; ++p) s -= Gl[p * KP + ll] * Gl[p * KP + c];
            double dcc = __shfl(s, c, 64);
            if (!(dcc > 0.0)) dcc = tabs(dcc) + 1e-30;
            const double lcc = sqrt(dcc);
            if (lin) Gl[c * KP + lane] = lane == c ? lcc : (lane > c ? s / lcc : 0.0);
            RK_WAVE_SYNC();
        }
        double y = b;
        for (int i = 0; i < k; ++i) {
            const double yi = __shfl(y, i, 64) / Gl[i * KP + i];
            if (lane == i) y = yi;
            else if (lane > i) y -= Gl[i * KP + ll] * yi;
        }
        for (int i = k - 1; i >= 0; --i) {
            const double xi = __shfl(y, i, 64) / Gl[i * KP + i];
            if (lane == i) y = xi;
            else if (lane < i) y -= Gl[ll * KP + i] * xi;
        }
        x = y;
        if (nonneg) x = x > 0.0 ? x : 0.0;
    } else {
        const double gd = Gl[ll * KP + ll];
        cd_static_sweeps<double, KP>(b, x, gd, fok, l1, nonneg, maxit, [&](auto IC) { return Gl[decltype(IC)::value * KP + ll]; });
    }
    if (fok) X[j * (int64_t)k + lane] = x;
}

// Squared error and count over the held-out entries (fit_cv.hpp:1444-1494), one wavefront per column of A.
// out partials: [block] = {sum of squared errors (fp64), count}
template <class T>
__global__ __launch_bounds__(256) void cv_test_error_kernel(
    const int* __restrict__ colptr, const int* __restrict__ rowidx, const T* __restrict__ vals, int64_t ncols, int nrows,
    const T* __restrict__ W_T, const T* __restrict__ d, const T* __restrict__ H, int k, unsigned long long seed,
    unsigned long long threshold, int mask_zeros, double* __restrict__ partial_sq, unsigned long long* __restrict__ partial_n) {
    __shared__ double shs[4];
    __shared__ unsigned long long shn[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * 4 + wave;
    double acc = 0.0;
    unsigned long long cnt = 0;
    if (j < ncols) {
        const bool fok = lane < k;
        // lane f holds d_f * H(f, j) for features f and f + 64 (k <= 128); a held-out entry costs one gather of W_T(row, :) and a
        // wave reduction
        const bool fok2 = lane + 64 < k;
        const T hd = fok ? H[j * (int64_t)k + lane] * d[lane] : T(0);
        const T hd2 = fok2 ? H[j * (int64_t)k + lane + 64] * d[lane + 64] : T(0);
        const unsigned col = (unsigned)j;
        if (mask_zeros) {
            for (int t = colptr[j]; t < colptr[j + 1]; ++t) {
                const int row = rowidx[t];
                if (!(cv_hash_dev(seed, (unsigned)row, col) < threshold)) continue;
                T p = fok ? W_T[(int64_t)row * k + lane] * hd : T(0);
                if (fok2) p = tfma(W_T[(int64_t)row * k + lane + 64], hd2, p);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) p += shfl_xor_t(p, off);
                const T diff = vals[t] - p;
                acc += static_cast<double>(diff * diff);
                ++cnt;
            }
        } else {
            int t = colptr[j];
            const int te = colptr[j + 1];
            for (int r0 = 0; r0 < nrows; r0 += 64) {
                const int r = r0 + lane;
                const bool held = r < nrows && cv_hash_dev(seed, (unsigned)r, col) < threshold;
                unsigned long long m = __ballot(held);
                while (m) {
                    const int bit = __builtin_ctzll(m);
                    m &= m - 1;
                    const int row = r0 + bit;
                    while (t < te && rowidx[t] < row) ++t;
                    const T actual = (t < te && rowidx[t] == row) ? vals[t] : T(0);
                    T p = fok ? W_T[(int64_t)row * k + lane] * hd : T(0);
                    if (fok2) p = tfma(W_T[(int64_t)row * k + lane + 64], hd2, p);
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) p += shfl_xor_t(p, off);
                    const T diff = actual - p;
                    acc += static_cast<double>(diff * diff);
                    ++cnt;
                }
            }
        }
    }
    if (lane == 0) { shs[wave] = acc; shn[wave] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial_sq[blockIdx.x] = shs[0] + shs[1] + shs[2] + shs[3];
        partial_n[blockIdx.x] = shn[0] + shn[1] + shn[2] + shn[3];
    }
}

// ---------------------------------------------------------------------------
// k x k feature layer (reference features/L21.hpp:38-51, features/angular.hpp:67-103)
// ---------------------------------------------------------------------------
// L21: G(i,i) += lambda / ||factor.row(i)||_2 for rows with norm > 1e-10; sumsq[i] = sum_j X(i,j)^2
template <class T>
__global__ void l21_diag_kernel(T* __restrict__ G, const T* __restrict__ sumsq, int k, T lambda) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    const T nrm = sqrt(sumsq[i]);
    if (nrm > T(1e-10)) G[(int64_t)i * k + i] += lambda / nrm;
}
// angular, step 1: M(i,l) = norm_i * cos(i,l) / norm_l with cos = offdiag of the Gram of the row-normalised factor
// (rows with norm <= 1e-15 are left unscaled); Gf = factor factor^T (no eps).  grad(:,j) = M x_j.
template <class T>
__global__ void angular_matrix_kernel(const T* __restrict__ Gf, int k, T* __restrict__ M) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= k * k) return;
    const int l = e / k, i = e % k;                 // M stored [l][i]: column l contiguous in i
    const T ni = sqrt(Gf[(int64_t)i * k + i]), nl = sqrt(Gf[(int64_t)l * k + l]);
    const T inv_i = ni > T(1e-15) ? T(1) / ni : T(1), inv_l = nl > T(1e-15) ? T(1) / nl : T(1);
    M[e] = (i == l) ? T(0) : ni * (Gf[(int64_t)l * k + i] * inv_i * inv_l) * inv_l;
}
// angular, step 2: x_j <- max(0, x_j - lambda M x_j), one wavefront per column, lane = feature (k <= 64)
template <class T, int VPL>   // VPL features per lane: k <= 64 * VPL
__global__ __launch_bounds__(256) void angular_apply_kernel(T* __restrict__ X, int k, int64_t ncols, const T* __restrict__ M,
                                                            T lambda) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* Ms = reinterpret_cast<T*>(smem_raw);          // k*k, [l][i]
    for (int e = threadIdx.x; e < k * k; e += blockDim.x) Ms[e] = M[e];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t j = wid; j < ncols; j += nw) {
        T x[VPL], g[VPL];
#pragma unroll
        for (int v = 0; v < VPL; ++v) { const int f = lane + 64 * v; x[v] = f < k ? X[j * (int64_t)k + f] : T(0); g[v] = T(0); }
        for (int l = 0; l < k; ++l) {           // features in order: the reference's M x_j accumulation
            const T xl = __shfl(x[l >> 6], l & 63, 64);
#pragma unroll
            for (int v = 0; v < VPL; ++v) { const int f = lane + 64 * v; g[v] = tfma(Ms[l * k + (f < k ? f : 0)], xl, g[v]); }
        }
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const int f = lane + 64 * v;
            if (f < k) {
                const T val = x[v] - lambda * g[v];
                X[j * (int64_t)k + f] = val > T(0) ? val : T(0);
            }
        }
    }
}

template <class T, int KP>   // KP in {32, 64}: features padded to KP (k <= KP), lane r = feature r
__global__ __launch_bounds__(256) void masked_solve_kernel(
    const int* __restrict__ colptr, const int* __restrict__ rowidx, const T* __restrict__ vals,
    const int* __restrict__ mask_p, const int* __restrict__ mask_i, int64_t ncols,
    const T* __restrict__ F, const T* __restrict__ Gfull, T* __restrict__ X, int k, T l1, T l2,
    int nonneg, int maxit, T tol, int solver_mode, int warm) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    T* Gl = reinterpret_cast<T*>(smem_raw) + (size_t)wave * KP * KP;   // [c][r]
    const int64_t j = (int64_t)blockIdx.x * 4 + wave;
    if (j >= ncols) return;
    const bool fok = lane < k;
    const bool lin = lane < KP;                 // KP = 32: the upper half of the wave only takes part in shuffles
    const int ll = lin ? lane : 0;
    // G_loc = G_full (padded with identity)
    for (int c = 0; c < KP; ++c) {
        T v = (fok && c < k) ? Gfull[(int64_t)c * k + lane] : (c == lane ? T(1) : T(0));
        if (lin) Gl[c * KP + lane] = v;
    }
    // b over unmasked nonzeros; delta-G over ALL masked rows
    T b = T(0);
    const int as = colptr[j], ae = colptr[j + 1];
    int ms = mask_p[j];
    const int me = mask_p[j + 1];
    for (int t = as; t < ae; ++t) {
        const int row = rowidx[t];
        while (ms < me && mask_i[ms] < row) ++ms;
        const bool masked = ms < me && mask_i[ms] == row;
        if (!masked && fok) b = tfma(vals[t], F[(int64_t)row * k + lane], b);
    }
    for (int t = mask_p[j]; t < me; ++t) {
        const int row = mask_i[t];
        const T fr = fok ? F[(int64_t)row * k + lane] : T(0);
        for (int c = 0; c < k; ++c) {
            const T fc = __shfl(fr, c, 64);
            if (lin) Gl[c * KP + lane] -= fr * fc;
        }
    }
    if (fok) { b -= l1; Gl[lane * KP + lane] += l2; }   // fok implies lane < KP
    RK_WAVE_SYNC();
    T x = (warm && fok) ? X[j * (int64_t)k + lane] : T(0);
    if (solver_mode == 1) {
        // in-LDS Cholesky (left-looking) then forward/back substitution; x = clip(G_loc^-1 b)
        for (int c = 0; c < KP; ++c) {
            T s = Gl[c * KP + ll];
            for (int p = 0; p < c; ++p) s -= Gl[p * KP + ll] * Gl[p * KP + c];
            T dcc = __shfl(s, c, 64);
            if (!(dcc > T(0))) dcc = tabs(dcc) + T(1e-30);
            const T lcc = sqrt(dcc);
            if (lin) Gl[c * KP + lane] = lane == c ? lcc : (lane > c ? s / lcc : T(0));
            RK_WAVE_SYNC();
        }
        T y = b;
        for (int i = 0; i < k; ++i) {   // forward: y_i = (b_i - sum_{p<i} L_ip y_p) / L_ii
            const T yi = __shfl(y, i, 64) / Gl[i * KP + i];
            if (lane == i) y = yi;
            else if (lane > i) y -= Gl[i * KP + ll] * yi;
        }
        for (int i = k - 1; i >= 0; --i) {  // backward: x_i = (y_i - sum_{p>i} L_pi x_p) / L_ii
            const T xi = __shfl(y, i, 64) / Gl[i * KP + i];
            if (lane == i) y = xi;
            else if (lane < i) y -= Gl[ll * KP + i] * xi;
        }
        x = y;
        if (nonneg) x = x > T(0) ? x : T(0);
    } else {
        // cd_nnls_col_fixed with the relative-change stop, static coordinate sweeps (cd_static_sweeps_tol above)
        const T gd = Gl[ll * KP + ll];
        cd_static_sweeps_tol<T, KP>(b, x, gd, fok, nonneg, maxit, tol, k, [&](auto IC) { return Gl[decltype(IC)::value * KP + ll]; });
    }
    if (fok) X[j * (int64_t)k + lane] = x;
}

// Per-element loss term of the distribution losses (math/loss.hpp:512-536 compute_loss): GP (4) :382-398, NB (5) :415-426,
// Gamma (6) / inverse Gaussian (7) / Tweedie (8) deviance terms :439-505.  y, mu (>= 1e-10 already) and theta in double; the
// caller casts the term to Scalar as the reference does.
__device__ __forceinline__ double dist_loss_term(int loss_type, double y, double mu, double th, double power) {
    if (loss_type == 4) {
        const double opt = 1.0 + th;
        double nll = -log(mu / opt);
        if (y >= 1.0) {
            double inner = (mu + th * y) / opt;
            inner = inner > 1e-10 ? inner : 1e-10;
            nll -= (y - 1.0) * log(inner);
        }
        return nll + (mu + th * y) / opt;
    }
    if (loss_type >= 6) {
        const double yy = y > 1e-10 ? y : 1e-10;
        const double pp = loss_type == 6 ? 2.0 : (loss_type == 7 ? 3.0 : power);
        if (loss_type == 7) {
            const double df = yy - mu;
            return df * df / (mu * mu * yy);
        }
        if (fabs(pp - 1.0) < 1e-6) return 2.0 * (yy * log(yy / mu) - (yy - mu));
        if (fabs(pp - 2.0) < 1e-6) return 2.0 * (-log(yy / mu) + (yy - mu) / mu);
        const double omp = 1.0 - pp, tmp = 2.0 - pp;
        return 2.0 * (pow(yy, tmp) / (omp * tmp) - yy * pow(mu, omp) / omp + pow(mu, tmp) / tmp);
    }
    const double r = th > 1e-10 ? th : 1e-10;
    return -lgamma(y + r) + lgamma(r) - r * log(r / (r + mu)) - y * log(mu / (r + mu));
}

// Loss over (unmasked) nonzeros, fp64 accumulation, one wavefront per column
// (reference nmf/masked_nnls.hpp:250-282; also the nonzero pass of evaluate()):
//   partial[2b]   = sum (a - p)^2,  partial[2b+1] = sum p^2,   p = sum_f d_f W_T(f,i) H(f,j)
// loss_type != 0 (a fit with an explicit mask AND a distribution loss): partial[2b] = sum compute_loss(a, p, loss) with the
// reference's default theta = 0 (masked_nnls.hpp:277 passes no dispersion and applies no robust modifier), each term cast
// to Scalar as there.
template <class T>
__global__ __launch_bounds__(256) void loss_nonzeros_kernel(
    const int* __restrict__ colptr, const int* __restrict__ rowidx, const T* __restrict__ vals,
    const int* __restrict__ mask_p, const int* __restrict__ mask_i, int64_t ncols,
    const T* __restrict__ W_T, const T* __restrict__ d, const T* __restrict__ H, int k, int loss_type, double power,
    double* __restrict__ partial) {
    __shared__ double sh[8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * 4 + wave;
    double acc = 0, acc2 = 0;
    if (j < ncols) {
        int ms = mask_p ? mask_p[j] : 0;
        const int me = mask_p ? mask_p[j + 1] : 0;
        for (int t = colptr[j]; t < colptr[j + 1]; ++t) {
            const int row = rowidx[t];
            while (ms < me && mask_i[ms] < row) ++ms;
            if (ms < me && mask_i[ms] == row) continue;
            T p = T(0);
            for (int f = lane; f < k; f += 64)
                p += (W_T[(int64_t)row * k + f] * d[f]) * H[j * (int64_t)k + f];
            double pd = static_cast<double>(p);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) pd += __shfl_xor(pd, off, 64);
            if (loss_type == 0) {
                const double df = static_cast<double>(vals[t]) - pd;
                acc += df * df;   // identical in all lanes
            } else {
                const double mu = static_cast<double>(static_cast<T>(pd));          // the prediction in Scalar, as the reference forms it
                acc += static_cast<double>(static_cast<T>(dist_loss_term(loss_type, static_cast<double>(vals[t]), mu > 1e-10 ? mu : 1e-10, 0.0, power)));
            }
            acc2 += pd * pd;
        }
    }
    if (lane == 0) { sh[wave] = acc; sh[4 + wave] = acc2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * (int64_t)blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
        partial[2 * (int64_t)blockIdx.x + 1] = sh[4] + sh[5] + sh[6] + sh[7];
    }
}
// out[0] = sum of even partials, out[1] = sum of odd partials (fixed order)
static __global__ __launch_bounds__(256) void sum_partials2(const double* __restrict__ partial, int n,
                                                      double* __restrict__ out) {
    __shared__ double sh[4];
    double a0 = 0, a1 = 0;
    for (int i = threadIdx.x; i < n; i += 256) { a0 += partial[2 * i]; a1 += partial[2 * i + 1]; }
    const double s0 = block_sum_256(a0, sh);
    const double s1 = block_sum_256(a1, sh);
    if (threadIdx.x == 0) { out[0] = s0; out[1] = s1; }
}


// ---------------------------------------------------------------------------
// Column work order for the CD kernels: sort columns by DESCENDING sweep count of the previous solve (counting
// sort, 128 bins).  A wave runs until its slowest column converges (measured on the bench workload: mean 37 sweeps
// per column but 54 per 64-column wave), so grouping columns that need similar sweep counts removes ~30 % of the
// wasted lane-sweeps, and long-running waves start first.  Columns are independent, so any order gives identical solutions.
// ---------------------------------------------------------------------------
// Two launches, no global atomics, nothing to zero beforehand: block b counts ITS contiguous chunk of the columns per bin
// (order_hist_kernel -> part[b][128]); order_scatter_kernel, launched with the same grid, turns the table into "where
// does bin x of block b start" (exclusive scan over bins of the totals + the counts of the blocks before b) and ranks
// its chunk.  The sort is STABLE -- inside one bin the columns keep their ascending index: each wavefront takes a
// contiguous quarter of the chunk, its bins start after those of the wavefronts before it, and inside a 64-column round
// the rank is the number of lower lanes with the same key (ballot).  So the order is a function of `sweeps` alone: the
// same fit lays its columns out the same way every run, and anything that ever reduces ACROSS columns in work order stays
// reproducible (the round-5 probe of row sums formed in the CD epilogues needed it: profiles/r05_fused_norms_ab.txt).
constexpr int ORDER_BLOCKS_MAX = 128;
static __device__ __forceinline__ void order_hist_body(const int* __restrict__ sweeps, int64_t n, unsigned int* __restrict__ part /*nb x 128*/,
                                                       const unsigned bid, const unsigned nb) {
    __shared__ unsigned int sh[128];
    if (threadIdx.x < 128) sh[threadIdx.x] = 0;
    __syncthreads();
    const int64_t per = (n + nb - 1) / nb;
    const int64_t i0 = (int64_t)bid * per;
    const int64_t i1 = i0 + per < n ? i0 + per : n;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
        int key = sweeps[i];
        key = key < 0 ? 0 : (key > 127 ? 127 : key);
        atomicAdd(&sh[127 - key], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 128) part[(size_t)bid * 128 + threadIdx.x] = sh[threadIdx.x];
}
static __global__ __launch_bounds__(256) void order_hist_kernel(const int* __restrict__ sweeps, int64_t n,
                                                                  unsigned int* __restrict__ part /*gridDim.x x 128*/) {
    order_hist_body(sweeps, n, part, blockIdx.x, gridDim.x);
}
static __device__ __forceinline__ void order_scatter_body(const int* __restrict__ sweeps, int64_t n,
                                                          const unsigned int* __restrict__ part /*nb x 128*/, int* __restrict__ order,
                                                          const unsigned bid, const unsigned nb) {
    __shared__ unsigned int base[128], half_total;
    __shared__ unsigned int tsum[2][128], bsum[2][128];
    __shared__ unsigned int wcnt[4][128];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t per = (n + nb - 1) / nb;
    const int64_t i0 = (int64_t)bid * per;
    const int64_t i1 = i0 + per < n ? i0 + per : n;
    const int64_t wper = (per + 3) / 4;                       // this wavefront's contiguous quarter [j0, j1)
    const int64_t j0 = i0 + w * wper < i1 ? i0 + w * wper : i1;
    const int64_t j1 = j0 + wper < i1 ? j0 + wper : i1;
    for (int t = threadIdx.x; t < 4 * 128; t += 256) (&wcnt[0][0])[t] = 0;
    // bin totals over all blocks, and the part of them that lies in blocks before this one: thread (h, x) sums every
    // second row of the table for bin x -- gridDim.x / 2 independent coalesced loads per thread instead of gridDim.x
    // dependent-looking ones in half of the threads
    {
        const unsigned int x = threadIdx.x & 127u, h = threadIdx.x >> 7;
        unsigned int tot = 0, before = 0;
#pragma unroll 8
        for (unsigned int b = h; b < nb; b += 2) {
            const unsigned int c = part[(size_t)b * 128 + x];
            before += b < bid ? c : 0u;
            tot += c;
        }
        tsum[h][x] = tot;
        bsum[h][x] = before;
    }
    __syncthreads();
    for (int64_t i = j0 + lane; i < j1; i += 64) {            // per-wavefront bin counts of the quarter
        int key = sweeps[i];
        key = key < 0 ? 0 : (key > 127 ? 127 : key);
        atomicAdd(&wcnt[w][127 - key], 1u);
    }
    if (threadIdx.x < 128) {
        const unsigned int tot = tsum[0][threadIdx.x] + tsum[1][threadIdx.x];
        const unsigned int before = bsum[0][threadIdx.x] + bsum[1][threadIdx.x];
        // exclusive scan of the totals over the 128 bins (two waves, shuffle scan)
        unsigned int incl = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned int t = __shfl_up(incl, d, 64); if ((threadIdx.x & 63) >= d) incl += t; }
        if (threadIdx.x == 63) half_total = incl;        // total of bins 0..63
        base[threadIdx.x] = incl - tot + before;
    }
    __syncthreads();
    // wcnt[w][x] <- where wavefront w's first column of bin x goes (the bins of the wavefronts before it come first)
    if (threadIdx.x < 128) {
        unsigned int run = base[threadIdx.x] + (threadIdx.x >= 64 ? half_total : 0u);
        for (int v = 0; v < 4; ++v) { const unsigned int t = wcnt[v][threadIdx.x]; wcnt[v][threadIdx.x] = run; run += t; }
    }
    __syncthreads();
    volatile unsigned int* wb = wcnt[w];                       // this wavefront's running bin starts (LDS; one wavefront = program order)
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int64_t r = j0; r < j1; r += 64) {
        const int64_t i = r + lane;
        const bool valid = i < j1;
        int bin = 0;
        if (valid) {
            int key = sweeps[i];
            key = key < 0 ? 0 : (key > 127 ? 127 : key);
            bin = 127 - key;
        }
        // lanes with my bin: seven ballots, one per bit of the bin (no loop over the distinct keys of the round)
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 7; ++bit) {
            const unsigned long long bm = __ballot((bin >> bit) & 1);
            same &= ((bin >> bit) & 1) ? bm : ~bm;
        }
        unsigned int pos = 0;
        if (valid) {
            const unsigned int start = wb[bin];                 // every lane of the group reads the same word ...
            pos = start + (unsigned int)__popcll(same & lt);
            if ((same & lt) == 0ull) wb[bin] = start + (unsigned int)__popcll(same);      // ... its lowest lane moves it on
        }
        // Serpentine: positions are handed out longest-first; every second group of 16 workgroups (128 slots each: four
        // 32-column or eight 16-column wavefronts) is laid out in reverse, so that neighbouring workgroups -- which the
        // dispatcher places on the same XCD / CU one after the other -- mix long and short tiles instead of stacking the
        // longest ones (measured on C2's H side: 494 -> 470 us, tools/probe/cd_order_probe.py; any period from 8 to 128
        // workgroups gives the same).  Only whole groups of full blocks are reversed: a bijection on the positions.
        if (valid) {
            const unsigned int blk = pos >> 7, grp = blk >> 4;
            if ((grp & 1u) && (uint64_t)(grp + 1u) * 16u <= (uint64_t)(n >> 7)) pos = (((grp << 4) + 15u - (blk & 15u)) << 7) | (pos & 127u);
            order[pos] = (int)i;
        }
    }
}
static __global__ __launch_bounds__(256) void order_scatter_kernel(const int* __restrict__ sweeps, int64_t n,
                                                                     const unsigned int* __restrict__ part /*gridDim.x x 128*/,
                                                                     int* __restrict__ order) {
    order_scatter_body(sweeps, n, part, order, blockIdx.x, gridDim.x);
}

}  // namespace rk
