// ============================================================================
// kernels_cd_mfma64.hip.h -- coordinate-descent NNLS with the rank-1 residual updates on the MATRIX cores, 16 columns per
// wavefront and FOUR coordinates per instruction: v_mfma_f64_16x16x4_f64 (the fp64 default, k <= 64) and
// v_mfma_f32_16x16x4_f32 (selectable for fp32: RCPPML_CD_MFMA16).
//
// Reference routine: primitives/cpu/nnls_batch.hpp:70-132 (cd_nnls_col_fixed), prologue fused_nnls.hpp:116-123.
//
// Same idea as kernels_cd_mfma.hip.h: the KP x 16 residual block lives in KP/16 accumulator tiles and four consecutive
// coordinates fill the four K-slots of one instruction per tile.  fp64 C/D map: col = lane&15, row = (lane>>4) + 4v --
// coordinates 16t+4v+g (g = lane>>4) sit in accumulator element v of tile t, one per 16-lane row group, with no row
// permutation; fp32 C/D map: row = 4*(lane>>4) + v -- the logical rows are permuted inside each tile (4v+g <-> 4g+v, by
// permuting G's rows in the A operand) to obtain the same placement.  Either way the B operand B[kk = lane>>4][col] is
// exactly "step of coordinate 4q+kk for this lane's column".  A quad is solved in four phases: every lane evaluates the
// reference's scalar step on its own residual, the step of group p is broadcast to the other groups
// (v_permlane16_swap / v_permlane32_swap, no LDS) and applied as the lazy Gauss-Seidel correction
// b_g -= G(c_g, c_p) a_p with a per-lane coefficient that is ZERO for groups g <= p -- so groups that are already done
// re-evaluate their step on unchanged inputs and the last evaluation leaves {a_0|a_1|a_2|a_3} = the MFMA B operand in
// place.  G enters lane-distributed (one conflict-free LDS read per tile and quad, quad-major layout).
// Arithmetic: the fma chain per accumulator element is the reference's, in the reference's order; the step uses
// b * (1/G_cc) (one rounding more than b / G_cc) and the tolerance term a hardware reciprocal (Newton-refined in fp64).
// Measured (C2, 20 fixed sweeps): fp64 2.3-2.5x the lane-group kernel; fp32 equal to the 32-column kernel at 100 000
// columns, 10 % faster at 20 000, 18 % slower at 400 000 (twice the scalar work per column) -- hence opt-in for fp32.
// ============================================================================
#pragma once
#include "kernels.hip.h"

namespace rk {

// Gq[((q*NT + t) << 6) + lane] = -G(16t + (lane&15), 4q + (lane>>4)), q = quad of coordinates, NT = KP/16.
// tab[c] (c = 4q + g) = { 1/G(c,c) (0 if G(c,c) <= 0),  g > 0 ? G(c, 4q) : 0,  g > 1 ? G(c, 4q+1) : 0,  g > 2 ? G(c, 4q+2) : 0 }.
// PERM (f32 form): the f32 16x16x4 C/D map is row = 4*(lane>>4) + v (four CONSECUTIVE rows per lane), so the logical
// rows are permuted, logical 4v+g <-> physical 4g+v inside every tile, to keep "coordinate 4q+g in row group g".
template <class T> struct Vec4T;
template <> struct Vec4T<float> { typedef float4 type; };
template <> struct Vec4T<double> { typedef double4 type; };
// Row-group broadcasts.  v_permlane16_swap_b32 vdst, src: vdst rows 1,3 <-> src rows 0,2 (16-lane rows);
// v_permlane32_swap_b32 vdst, src: vdst rows 2,3 <-> src rows 0,1; with vdst == src the exchange is in place
// (probed on gfx950: tools/probe/permlane_probe.hip).  Only the groups BEHIND row P need the value; the others may
// receive anything finite (their coefficient is zero).
template <int P> __device__ __forceinline__ unsigned bcast_row32(unsigned a) {
    // builtins, not inline asm: hipcc then knows the instruction's hazards and needs one s_nop per swap instead of two (a
    // lone wave -- the W side of C2 -- pays an issue slot for every one of them)
    if constexpr (P == 2) {                 // row 3 <- row 2
        return __builtin_amdgcn_permlane16_swap(a, a, false, false)[0];
    } else {
        const auto tu = __builtin_amdgcn_permlane16_swap(a, a, false, false);   // [0] = [r0 r0 r2 r2], [1] = [r1 r1 r3 r3]
        if constexpr (P == 0) {             // rows 1,2,3 <- row 0
            return __builtin_amdgcn_permlane32_swap(tu[0], tu[0], false, false)[0];   // [r0 r0 r0 r0]
        } else {                            // rows 2,3 <- row 1
            return __builtin_amdgcn_permlane32_swap(tu[1], tu[1], false, false)[0];   // [r3 r3 r1 r1]
        }
    }
}
template <int P> __device__ __forceinline__ float bcast_row(float a) {
    return __uint_as_float(bcast_row32<P>(__float_as_uint(a)));
}
template <int P> __device__ __forceinline__ double bcast_row(double a) {
    const unsigned long long u = __double_as_longlong(a);
    const unsigned lo = bcast_row32<P>((unsigned)u), hi = bcast_row32<P>((unsigned)(u >> 32));
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

template <class T> struct CdStepOut64 { T a, nx; };
__device__ __forceinline__ float tmax(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double tmax(double a, double b) { return __builtin_fmax(a, b); }

// fp64: diff = b / G_cc as the reference divides (cd_quotient, kernels.hip.h: the reciprocal's quotient + one exact-residual
// correction); fp32: b * (1 / G_cc).  gd is only read in fp64.
template <bool SIMPLE, class T>
__device__ __forceinline__ CdStepOut64<T> cd_step64(T b, T xo, T gd, T ginv, bool active, T l1_cd, T l2_cd, T lo, T hi) {
    CdStepOut64<T> o;
    if constexpr (SIMPLE) {
        // a = max(diff, -xo), x = max(xo + diff, 0): the reference's clamped step with a two-instruction dependent chain
        // (see cd_scalar_step in kernels_cd_mfma.hip.h).  `active` and `if (g_diag <= 0) continue;` arrive folded into
        // ginv (= 0): diff = 0, a = 0, nx = xo
        const T diff = cd_quotient(b, gd, ginv);
        o.a = tmax(diff, -xo);
        o.nx = tmax(xo + diff, T(0));
    } else {
        T diff = cd_quotient(b, gd, ginv);
        diff -= l1_cd;
        diff = tfma(l2_cd, xo, diff);
        const T nv = xo + diff;
        const bool neg = nv < lo, up = nv > hi;
        T nx = neg ? lo : (up ? hi : nv);
        T a = neg ? lo - xo : (up ? hi - xo : diff);
        const bool on = active && (ginv > T(0));
        o.a = on ? a : T(0);
        o.nx = on ? nx : xo;
    }
    return o;
}

__device__ __forceinline__ f64x4 mfma16(double a, double b, f64x4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
template <class T> struct Acc4T;
template <> struct Acc4T<float> { typedef f32x4 type; };
template <> struct Acc4T<double> { typedef f64x4 type; };
// |a| / den with a hardware reciprocal: v_rcp_f32 (fp32, as the 32-column kernel) or v_rcp_f64 + one Newton step
__device__ __forceinline__ float fast_recip(float den) { return __builtin_amdgcn_rcpf(den); }
__device__ __forceinline__ double fast_recip(double den) {
    const double rc = __builtin_amdgcn_rcp(den);
    return __builtin_fma(__builtin_fma(-den, rc, 1.0), rc, rc);
}

template <class T, int NT, bool SIMPLE>   // KP = 16*NT rows (k <= KP), 16 columns per wave, 4 waves per block share G
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NT <= 2 ? 4 : (NT <= 4 ? 2 : 1), 8)))
void cd_mfma64_kernel(const T* __restrict__ G /* k x k, as rcppml_hip_gram wrote it */, const T* __restrict__ B,
                      T* __restrict__ X, int k, int64_t ncols, T l1_pre, int warm, int zero_init, T l1_cd,
                      T l2_cd, int nonneg, int maxit, T tol, T ub_cd, T ub_post,
                      int* __restrict__ sweeps, const int* __restrict__ order, unsigned long long* __restrict__ stats) {
    typedef typename Vec4T<T>::type Tab4;
    typedef typename Acc4T<T>::type Acc4;
    constexpr int KP = 16 * NT;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* Gs = reinterpret_cast<T*>(smem_raw);               // KP*KP, quad-major (layout comment above)
    Tab4* tab_s = reinterpret_cast<Tab4*>(Gs + KP * KP);     // KP x {1/G_cc, 3 in-quad couplings}
    [[maybe_unused]] T* gd_s = Gs + KP * KP + 4 * KP;        // fp64 only: KP x G_cc (the corrected quotient's exact residual needs it)
    // operand image and per-coordinate table formed here from the k x k Gram (layout comment above; until round 3 a launch of its own)
    {
        constexpr bool PERM = sizeof(T) == 4;
        auto gp = [&](int col, int row) { return (row < k && col < k) ? G[(int64_t)col * k + row] : (row == col ? T(1) : T(0)); };
        for (int e = threadIdx.x; e < KP * KP; e += blockDim.x) {
            const int qt = e >> 6, kk = (e >> 4) & 3, ps = e & 15;
            const int q = qt / NT, t = qt % NT;
            const int rl = PERM ? (ps >> 2) + 4 * (ps & 3) : ps;          // inverse of ps = 4 (rl & 3) + (rl >> 2)
            Gs[e] = -gp(4 * q + kk, 16 * t + rl);
        }
        for (int c = threadIdx.x; c < KP; c += blockDim.x) {
            const int gg = c & 3, qb = c & ~3;
            Tab4 v;
            const T gd = gp(c, c);
            v.x = gd > T(0) ? T(1) / gd : T(0);
            v.y = gg > 0 ? gp(qb + 0, c) : T(0);
            v.z = gg > 1 ? gp(qb + 1, c) : T(0);
            v.w = gg > 2 ? gp(qb + 2, c) : T(0);
            tab_s[c] = v;
            if constexpr (sizeof(T) == 8) gd_s[c] = gd > T(0) ? gd : T(0);
        }
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane >> 4, cl = lane & 15;
    const int64_t base = ((int64_t)blockIdx.x * (blockDim.x >> 6) + wave) * 16;
    if (base >= ncols) return;
    const int64_t slot = base + cl;
    const bool inb = slot < ncols;
    const int64_t j = (inb && order) ? order[slot] : slot;
    Acc4 acc[NT];
    T xr[NT][4];
    {
        const T* bj = B + j * (int64_t)k;
        const T* xj = X + j * (int64_t)k;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int row = 16 * t + 4 * v + g;
                const bool ok = inb && row < k;
                T bv = ok ? bj[row] : T(0);
                if (ok && l1_pre != T(0)) bv -= l1_pre;
                acc[t][v] = bv;
                xr[t][v] = (ok && !zero_init) ? xj[row] : T(0);
            }
    }
    if (warm) {   // B -= G X (fused_nnls.hpp:121-123): the same MFMA stream with x in place of the steps
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int q = 4 * t + v;
#pragma unroll
                for (int t2 = 0; t2 < NT; ++t2) {
                    const T av = Gs[((q * NT + t2) << 6) + lane];
                    acc[t2] = mfma16(av, xr[t][v], acc[t2]);
                }
            }
    }
    const T lo = nonneg ? T(0) : -INFINITY;
    const T hi = ub_cd > T(0) ? ub_cd : INFINITY;
    const bool check = tol > T(0);
    const T inv_k = T(1) / static_cast<T>(k);
    bool active = inb;
    int nsweep = 0;
    // operands of the first quad; every quad then requests the NEXT quad's operands before it starts computing
    Tab4 tb_c = tab_s[g];
    [[maybe_unused]] T gd_c = T(0);
    if constexpr (sizeof(T) == 8) gd_c = gd_s[g];
    T av_c[NT];
#pragma unroll
    for (int t2 = 0; t2 < NT; ++t2) av_c[t2] = Gs[(t2 << 6) + lane];
    for (int it = 0; it < maxit; ++it) {
        if (!__any(active)) break;
        T tsum = T(0);
        nsweep += active ? 1 : 0;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                constexpr int NQ = KP / 4;
                const int qn = (4 * t + v + 1) % NQ;
                const int tn = (v == 3 ? t + 1 : t) % NT;          // row tile of the next quad
                const Tab4 tb_n = tab_s[4 * qn + g];
                [[maybe_unused]] T gd_n = T(0);
                if constexpr (sizeof(T) == 8) gd_n = gd_s[4 * qn + g];
                T av_n[NT];
#pragma unroll
                for (int t2 = 0; t2 < NT; ++t2) av_n[t2] = Gs[((qn * NT + t2) << 6) + lane];
                const T b0 = acc[t][v];
                const T xo = xr[t][v];
                const T ginv = SIMPLE ? (active ? tb_c.x : T(0)) : tb_c.x;
                // phase p: group p's step is final; the groups behind it take the lazy correction, the others keep b
                const CdStepOut64<T> s0 = cd_step64<SIMPLE, T>(b0, xo, gd_c, ginv, active, l1_cd, l2_cd, lo, hi);
                const T b1 = tfma(-tb_c.y, bcast_row<0>(s0.a), b0);
                const CdStepOut64<T> s1 = cd_step64<SIMPLE, T>(b1, xo, gd_c, ginv, active, l1_cd, l2_cd, lo, hi);
                const T b2 = tfma(-tb_c.z, bcast_row<1>(s1.a), b1);
                const CdStepOut64<T> s2 = cd_step64<SIMPLE, T>(b2, xo, gd_c, ginv, active, l1_cd, l2_cd, lo, hi);
                const T b3 = tfma(-tb_c.w, bcast_row<2>(s2.a), b2);
                const CdStepOut64<T> s3 = cd_step64<SIMPLE, T>(b3, xo, gd_c, ginv, active, l1_cd, l2_cd, lo, hi);
                xr[t][v] = s3.nx;
                // |a| / (|x_new| + 1e-15)  (nnls_batch.hpp:117-120): v_rcp_f64 + one Newton step
                tsum = tfma(tabs(s3.a), fast_recip(tabs(s3.nx) + T(1e-15)), tsum);
                // the row tile that holds the NEXT quad's residuals goes first
#pragma unroll
                for (int s = 0; s < NT; ++s) {
                    const int t2 = (tn + s) % NT;
                    acc[t2] = mfma16(av_c[t2], s3.a, acc[t2]);
                }
                tb_c = tb_n;
                gd_c = gd_n;
#pragma unroll
                for (int t2 = 0; t2 < NT; ++t2) av_c[t2] = av_n[t2];
                __builtin_amdgcn_sched_barrier(0);      // one scheduling region per quad (see kernels_cd_mfma.hip.h)
            }
        // branch-free activity update (see kernels_cd_mfma.hip.h); the four row groups hold the four coordinate classes
        T tot = tsum + __shfl_xor(tsum, 16, 64);
        tot += __shfl_xor(tot, 32, 64);
        active = active && !(check && tot * inv_k < tol);
    }
    if (inb) {
        T* xj = X + j * (int64_t)k;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int row = 16 * t + 4 * v + g;
                if (row < k) {
                    T val = xr[t][v];
                    if (ub_post > T(0)) val = val < ub_post ? val : ub_post;
                    xj[row] = val;
                }
            }
        if (sweeps && g == 0) sweeps[j] = nsweep;
    }
    cd_stats_add(stats, (inb && g == 0) ? nsweep : 0, (inb && g == 0) ? 1 : 0);
}

}  // namespace rk
