// kernels_small.hip.h -- the WHOLE alternating-NNLS fit of a small sparse matrix as ONE persistent kernel on ONE XCD.
//
// Why: hawaiibirds (183 x 1183, 30 815 nonzeros, k = 10; BASELINE configs[0]) needs < 10 us of arithmetic per ALS iteration, but the
// multi-launch loop issues 15 kernels for it and returns the loss to the host every iteration: ~150 us per iteration, slower than the
// host's own cores (round-5 verdict, weak item 5).  Everything such a fit touches fits one XCD's 4 MiB L2, so the iteration's four
// grid-wide dependencies (row norms -> scaling, Gram -> solve, on both sides) become barriers INSIDE one kernel whose workgroups
// all sit on the same XCD: one L2, hence no write-back / invalidate of it -- a relaxed L2 atomic counter, `s_waitcnt vmcnt(0)` before
// it (stores are write-through: they are in the L2 once acknowledged), and every load of data another CU may have rewritten is an
// L1-bypassing (`sc1`) load (sm_ldg).
// Measured (tools/probe/grid_barrier_probe.hip, profiles/r06_grid_barrier.txt): 0.8 us per barrier among 32 workgroups of XCD 0,
// no stale read in 2 000 rounds; the same barrier with agent-scope fences across all 8 XCDs costs 2.5 us (64 workgroups) to 10 us (256),
// and the relaxed form across XCDs reads stale data -- which is why the kernel keeps to one XCD: the launch has 8 x NB workgroups,
// the dispatcher deals them round-robin over the XCDs (measured), and those that do not land on XCD 0 (HW_REG_XCC_ID) leave at once.
// The 32 that stay must all be resident at once (12 waves of up to 168 VGPRs: one workgroup per CU).  If they are not -- another
// persistent fit holds CUs of XCD 0, the device is partitioned differently -- a barrier gives up after 0.2 s and the caller restarts on
// the multi-launch loop: slower, never hung.
//
// The loop is nmf_fit<CPU> (reference inst/include/FactorNet/nmf/fit_cpu.hpp:444-1855) for the plain sparse
// MSE fit -- fused right-hand side + solve per column (primitives/cpu/fused_nnls.hpp:70-134 CD with the iteration-0 quirk, :185-219
// Cholesky + clip), L1 / L2 / upper bounds / non-negativity, L1 / L2 / no row scaling (nmf/variant_helpers.hpp:286-305), Gram-trick
// loss (fit_cpu.hpp:1729-1753) and the convergence rule (:1769-1809) evaluated ON THE DEVICE: the host reads the result once.
// Mapping: one wavefront per column, lane = factor (k <= 32); the column's right-hand side never leaves the registers between the
// sparse product and the solve (the reference's fused form); the k x k Gram and the k row norms are per-wave partial sums, summed
// in a fixed order by every workgroup for itself after the barrier (deterministic, identical in every workgroup).
#pragma once
#include "kernels.hip.h"

namespace rk {

constexpr int SM_WPB = 12;                      // wavefronts per workgroup (three per SIMD: the kernel's variants need <= 168 VGPRs)
constexpr int SM_NB = 32;                       // participating workgroups (one XCD: 32 CUs)

template <class T>
struct SmallFit {
    const int* Ap; const int* Ai; const T* Ax;      // CSC(A):   n columns
    const int* Tp; const int* Ti; const T* Tx;      // CSC(A^T): m columns
    int m, n, k;
    T* W; T* H; T* d;                               // k x m, k x n, k (in / out)
    T* Bw;                                          // k x m: raw right-hand side of the W half-update (the loss's cross term)
    T* part;                                        // SM_NB x (KP + KP * KP): per-workgroup partial row norms | partial Gram
    double* crossp;                                 // SM_NB partial cross terms
    const double* trAtA;
    T L1_H, L1_W, L2_H, L2_W, ub_H, ub_W, cd_tol;
    int nonneg_H, nonneg_W, norm_type, cd_maxit;
    int max_iter, patience;
    int iter0;                                      // iterations this fit has already run (a launch that continues one: warm start from its first iteration)
    double tol;
    double* loss_hist;                              // max_iter doubles (may be null)
    double* result;                                 // [0] iterations [1] converged [2] train loss [3] final tol [4] 1 = done (stays 0 when a barrier gave up) [5..7] workgroup 0's clock
    unsigned* sync;                                 // [0] barrier counter [1] live tickets [2] abort flag
};

__device__ __forceinline__ unsigned sm_xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
__device__ __forceinline__ unsigned sm_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Load of data another CU may have rewritten since this CU last read it (factors, partial sums): relaxed, agent scope = the `sc1` form
// of global_load, which does not hit in the CU's L1.  An L1 invalidate after the barrier (`buffer_inv sc0`) followed by ordinary loads
// is NOT enough on gfx950 -- measured: 2.2e8 stale reads in tools/probe/grid_barrier_probe.hip, none with these loads.
__device__ __forceinline__ float shfl_t(float v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ double shfl_t(double v, int src) { return __shfl(v, src, 64); }
template <class T> __device__ __forceinline__ T sm_ldg(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Barrier among the SM_NB workgroups of one XCD.  Returns false when it gave up (a participant never arrived: the launch did not
// spread as measured) -- the caller leaves and the host falls back to the multi-launch loop.
__device__ __forceinline__ bool sm_barrier(unsigned* sync, unsigned& gen, int* sh_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // every thread: its stores are acknowledged by the L2 (write-through L1)
    __syncthreads();
    if (threadIdx.x == 0) {
        ++gen;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // this workgroup's stores are in the L2 (one thread's wait covers its own; the barrier above the others')
        __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = wall_clock64();                                 // the 100 MHz constant clock: 2e7 ticks = 0.2 s
        int ok = 1;
        while (sm_ld(sync) < gen * (unsigned)SM_NB) {
            __builtin_amdgcn_s_sleep(1);
            if (sm_ld(sync + 2) || wall_clock64() - t0 > 20000000ll) { __hip_atomic_store(sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 0; break; }
        }
        *sh_flag = ok;
    }
    __syncthreads();
    return *sh_flag != 0;
}

// y = L^-T L^-1 b for ONE column held one coordinate per lane (lane = coordinate; Lrow[c] = L(lane, c), Lcol[c] = L(c, lane)): the
// substitutions of llt.solve column by column -- element i receives its subtractions in the order p = 0 .. i-1 (forward) and
// p = KP-1 .. i+1 (backward); divisions as the reference (fused_nnls.hpp:200-218).
template <class T, int KP>
__device__ __forceinline__ T sm_chol_solve(T b, const T (&Lrow)[KP], const T (&Lcol)[KP], T ldiag, int lane) {
    cd_static_for<0, KP>([&](auto IC) {
        constexpr int i = decltype(IC)::value;
        const T q = b / ldiag;                                   // lane i's is y_i
        const T yi = lane_value(q, i);
        b = lane == i ? yi : (lane > i ? tfma(-Lrow[i], yi, b) : b);
    });
    cd_static_for<0, KP>([&](auto IC) {
        constexpr int i = KP - 1 - decltype(IC)::value;
        const T q = b / ldiag;
        const T xi = lane_value(q, i);
        b = lane == i ? xi : (lane < i ? tfma(-Lcol[i], xi, b) : b);
    });
    return b;
}

// Sparse product of one column, lane = factor.  The column's (row, value) pairs are fetched 64 at a time, one per lane (ONE round trip to
// the L2 per batch instead of one per 64 / KP entries), handed to the NG = 64 / KP lane groups through ds_bpermute (group g takes
// entries g, g + NG, ...), and all of a batch's row gathers are issued before the first is consumed; the group sums are added with an
// xor tree (every lane ends with the full sum of its factor).  The walk is latency: with one or two waves per SIMD nothing hides a
// round trip, so the number of dependent round trips per column is what counts (two per batch).
template <class T, int KP>
__device__ __forceinline__ T sm_rhs(const int* __restrict__ ci, const T* __restrict__ cx, int start, int end, const T* F /* rewritten by other CUs between phases: no restrict */, int k, int lane) {
    constexpr int NG = 64 / KP, PER = 64 / NG;          // PER = entries of a batch per lane group (= KP)
    const int g = lane / KP, f = lane % KP;
    const bool fok = f < k;
    const T* Ff = F + (fok ? f : 0);
    T acc = T(0);
    for (int base = start; base < end; base += 64) {
        const int t = base + lane;
        const bool ok = t < end;
        const int rv = ok ? ci[t] : 0;
        const T vv = ok ? cx[t] : T(0);
        const int nb = end - base < 64 ? end - base : 64;
        constexpr int UB = PER < 16 ? PER : 16;            // gathers in flight per lane (32 of them spill at KP = 32)
#pragma unroll
        for (int u0 = 0; u0 < PER; u0 += UB) {
            if (u0 * NG >= nb) break;
            T ff[UB], va[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int src = (u0 + u) * NG + g;         // this group's (u0 + u)-th entry of the batch
                const int row = __shfl(rv, src, 64);
                va[u] = shfl_t(vv, src);
                ff[u] = (fok && (u0 + u) * NG < nb) ? sm_ldg(Ff + (int64_t)row * k) : T(0);      // (entries past the column's end carry value 0)
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) acc = tfma(va[u], ff[u], acc);
        }
    }
#pragma unroll
    for (int off = KP; off < 64; off <<= 1) acc += shfl_xor_t(acc, off);
    return acc;
}

// One half-update over this wave's columns (columns gw, gw + nw, ...): fused right-hand side + solve, result written to X, raw
// right-hand side kept in Braw (W side: the loss needs it), the wave's partial row norms returned.
template <class T, int KP, bool CHOL>
__device__ __forceinline__ T sm_half_update(const int* __restrict__ cp, const int* __restrict__ ci, const T* __restrict__ cx, int ncols,
                                            const T* F, T* X, T* Braw, int k, const T* Gl /* LDS: KP x KP, G(c, r) at [c * KP + r] */,
                                            const T* Ll /* LDS: Cholesky factor, L(r, c) at [c * KP + r] */, T l1, T ub, int nonneg, int warm, int maxit,
                                            T tol, int norm_type, int gw, int nw, int lane) {
    const int f = lane % KP;
    const bool fok = lane < k;                                     // the lanes that own a coordinate of the solve
    T gcol[KP], lrow[KP];
    T gd = T(1);
    if constexpr (CHOL) {
#pragma unroll
        for (int c = 0; c < KP; ++c) { gcol[c] = Ll[f * KP + c]; lrow[c] = Ll[c * KP + f]; }      // gcol = L(c, lane) (column `lane` of L), lrow = L(lane, c)
        gd = Ll[f * KP + f];
    } else {
#pragma unroll
        for (int c = 0; c < KP; ++c) { gcol[c] = Gl[c * KP + f]; lrow[c] = T(0); }
        gd = Gl[f * KP + f];
    }
    T nacc = T(0);
    for (int j = gw; j < ncols; j += nw) {
        T b = sm_rhs<T, KP>(ci, cx, cp[j], cp[j + 1], F, k, lane);
        if (Braw && fok) Braw[(int64_t)j * k + lane] = b;
        b = fok ? b - l1 : T(0);                                   // b - 0 is exact
        T x;
        if constexpr (CHOL) {
            x = sm_chol_solve<T, KP>(b, lrow, gcol, gd, lane);
            if (nonneg) x = x > T(0) ? x : T(0);
            x = fok ? x : T(0);
        } else {
            x = fok ? sm_ldg(X + (int64_t)j * k + lane) : T(0);             // iteration 0: the caller's start, uncorrected (fused_nnls.hpp:116-123)
            if (warm) {
                const T xw = x;
                cd_static_for<0, KP>([&](auto IC) {
                    constexpr int i = decltype(IC)::value;
                    b = tfma(-gcol[i], lane_value(xw, i), b);
                });
            }
            (void)cd_static_sweeps_tol<T, KP>(b, x, gd, fok, nonneg, maxit, tol, k, [&](auto IC) { return gcol[decltype(IC)::value]; });
        }
        if (ub > T(0)) x = x < ub ? x : ub;
        if (fok) X[(int64_t)j * k + lane] = x;
        nacc += norm_type == 1 ? x * x : (x < T(0) ? -x : x);
    }
    return nacc;
}

// ---------------------------------------------------------------------------
// k <= 16: FOUR columns per wavefront, one per 16-lane DPP row (lane = (row g, factor f)).  With one column per wave three quarters of
// the lanes idle and a wave walks its columns one after the other -- hawaiibirds' H side is 1 183 columns on 256 waves, five dependent
// solves deep; four abreast it is two.  Per column the arithmetic is the one-column form's statement for statement: the broadcast
// of coordinate i is `row_newbcast:i` inside the row instead of v_readlane, the relative-change sum is the same xor tree (its upper
// levels added zeros), and a row whose column has met the stop takes no further step (1 / G_ii := 0 makes every later step exactly 0).
// ---------------------------------------------------------------------------
template <int I> __device__ __forceinline__ float sm_row_bcast(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + I, 0xf, 0xf, true));
}
template <int I> __device__ __forceinline__ int sm_row_bcast(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + I, 0xf, 0xf, true); }
template <int I> __device__ __forceinline__ double sm_row_bcast(double v) {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, 0x150 + I, 0xf, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), 0x150 + I, 0xf, 0xf, true);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

template <class T, bool CHOL>
__device__ __forceinline__ T sm_half_update4(const int* __restrict__ cp, const int* __restrict__ ci, const T* __restrict__ cx, int ncols,
                                             const T* F, T* X, T* Braw, int k, const T* Gl, const T* Ll, T l1, T ub, int nonneg, int warm, int maxit,
                                             T tol, int norm_type, int gw, int nw, int lane) {
    constexpr int KP = 16;
    const int g = lane >> 4, f = lane & 15;
    const bool fok = f < k;
    T gcol[KP], lrow[KP];
    T gd = T(1);
    if constexpr (CHOL) {
#pragma unroll
        for (int c = 0; c < KP; ++c) { gcol[c] = Ll[f * KP + c]; lrow[c] = Ll[c * KP + f]; }
        gd = Ll[f * KP + f];
    } else {
#pragma unroll
        for (int c = 0; c < KP; ++c) { gcol[c] = Gl[c * KP + f]; lrow[c] = T(0); }
        gd = Gl[f * KP + f];
    }
    const bool alive = fok && gd > T(0);
    const T ginv = alive ? T(1) / gd : T(0);
    const T pinf = static_cast<T>(__builtin_inff());
    const T inf_rt = maxit >= 0 ? pinf : T(0);                     // +inf at run time (cd_static_max)
    const bool check = tol > T(0);
    const T inv_k = T(1) / static_cast<T>(k);
    const T* Ff = F + (fok ? f : 0);
    T nacc = T(0);
    for (int j0 = 4 * gw; j0 < ncols; j0 += 4 * nw) {
        const int j = j0 + g;
        const bool cok = j < ncols;
        // ---- sparse product: every row walks ITS column, sixteen entries per round trip, sixteen gathers in flight
        const int start = cok ? cp[j] : 0, end = cok ? cp[j + 1] : 0;
        T b = T(0);
        for (int base = start; __any(base < end); base += 16) {
            const int t = base + f;
            const bool ok = t < end;
            const int rv = ok ? ci[t] : 0;
            const T vv = ok ? cx[t] : T(0);
            T ff[16], va[16];
            cd_static_for<0, 16>([&](auto IC) {
                constexpr int u = decltype(IC)::value;
                const int row = sm_row_bcast<u>(rv);
                va[u] = sm_row_bcast<u>(vv);
                ff[u] = (fok && base + u < end) ? sm_ldg(Ff + (int64_t)row * k) : T(0);
            });
#pragma unroll
            for (int u = 0; u < 16; ++u) b = tfma(va[u], ff[u], b);
        }
        if (Braw && fok && cok) Braw[(int64_t)j * k + f] = b;
        b = fok ? b - l1 : T(0);
        T x;
        if constexpr (CHOL) {
            cd_static_for<0, KP>([&](auto IC) {                    // forward, then backward substitution (sm_chol_solve, inside the row)
                constexpr int i = decltype(IC)::value;
                const T yi = sm_row_bcast<i>(b / gd);
                b = f == i ? yi : (f > i ? tfma(-lrow[i], yi, b) : b);
            });
            cd_static_for<0, KP>([&](auto IC) {
                constexpr int i = KP - 1 - decltype(IC)::value;
                const T xi = sm_row_bcast<i>(b / gd);
                b = f == i ? xi : (f < i ? tfma(-gcol[i], xi, b) : b);
            });
            x = b;
            if (nonneg) x = x > T(0) ? x : T(0);
            x = (fok && cok) ? x : T(0);
        } else {
            x = (fok && cok) ? sm_ldg(X + (int64_t)j * k + f) : T(0);
            if (warm) {
                const T xw = x;
                cd_static_for<0, KP>([&](auto IC) {
                    constexpr int i = decltype(IC)::value;
                    b = tfma(-gcol[i], sm_row_bcast<i>(xw), b);
                });
            }
            bool act = cok;                                         // row-uniform: this row's column still sweeps
            for (int it = 0; it < maxit; ++it) {
                if (!__any(act)) break;
                const T xe = !alive ? T(0) : (nonneg ? x : pinf);
                const T gi = act ? ginv : T(0);
                T aown = T(0);
                cd_static_for<0, KP>([&](auto IC) {
                    constexpr int i = decltype(IC)::value;
                    const T diff = cd_static_diff(b, gd, gi, T(0));
                    const T ad = cd_static_max(diff, -xe, inf_rt);
                    const T ad_i = sm_row_bcast<i>(ad);
                    b = tfma(-gcol[i], ad_i, b);
                    aown = f == i ? ad_i : aown;
                });
                const T xn = x + aown;
                const bool moved = xn != x;
                x = xn;
                if (check) {
                    T term = fok ? (aown < T(0) ? -aown : aown) / ((x < T(0) ? -x : x) + T(1e-15)) : T(0);
#pragma unroll
                    for (int off = 8; off > 0; off >>= 1) term += shfl_xor_t(term, off);
                    if (term * inv_k < tol) act = false;
                } else {
                    const unsigned long long mv = __ballot(moved);
                    if (!((mv >> (lane & 48)) & 0xffffull)) act = false;
                }
            }
        }
        if (ub > T(0)) x = x < ub ? x : ub;
        if (fok && cok) X[(int64_t)j * k + f] = x;
        nacc += norm_type == 1 ? x * x : (x < T(0) ? -x : x);
    }
    nacc += shfl_xor_t(nacc, 16);                                   // the four rows' partial norms of factor f
    nacc += shfl_xor_t(nacc, 32);
    return nacc;
}

// Unblocked lower Cholesky of the KP x KP matrix in LDS (A(r, c) at [c * KP + r]) by ONE wavefront, left-looking as chol_factor_kernel
// above (Eigen::LLT, fused_nnls.hpp:185): lane = row holds its row in REGISTERS, L(j, p) reaches the other rows by v_readlane -- a
// dependent chain of ~KP^2 / 2 fmas instead of as many LDS round trips (the LDS form cost ~5 us per factorisation at KP = 16, twice
// per iteration while seven waves wait).  L overwrites the lower triangle, the upper one is zeroed.
template <class T, int KP>
__device__ __forceinline__ void sm_chol_factor(T* A, int lane) {
    const int r = lane % KP;
    T a[KP];
#pragma unroll
    for (int c = 0; c < KP; ++c) a[c] = A[c * KP + r];
    cd_static_for<0, KP>([&](auto JC) {
        constexpr int j = decltype(JC)::value;
        T sv = a[j];
        cd_static_for<0, j>([&](auto PC) {
            constexpr int pp = decltype(PC)::value;
            sv -= a[pp] * lane_value(a[pp], j);                     // L(r, p) L(j, p): row j's entries p < j are final
        });
        T djj = lane_value(sv, j);
        if (!(djj > T(0))) djj = tabs(djj) + T(1e-30);
        const T ljj = sqrt(djj);
        a[j] = r == j ? ljj : sv / ljj;                            // (rows above j hold garbage here: zeroed on the way out)
    });
    if (lane < KP) {
#pragma unroll
        for (int c = 0; c < KP; ++c) A[c * KP + r] = r >= c ? a[c] : T(0);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <class T, int KP, bool CHOL>
__global__ __launch_bounds__(64 * SM_WPB) void als_small_kernel(SmallFit<T> P) {
    if (sm_xcc_id() != 0) return;
    __shared__ unsigned sh_ticket;
    __shared__ int sh_flag;
    __shared__ T Gh[KP * KP], Gsaved[KP * KP], Gwt[KP * KP], Lf[CHOL ? KP * KP : 1];
    __shared__ T dsh[KP], nsh[KP];
    __shared__ T red_t[64 * SM_WPB];
    __shared__ T wnorm[SM_WPB * KP];
    __shared__ T wtile[KP == 16 ? SM_WPB * KP * KP : 1];
    __shared__ double red[64];
    __shared__ double sh_loss[2];
    if (threadIdx.x == 0) sh_ticket = __hip_atomic_fetch_add(P.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned me = sh_ticket;
    if (me >= (unsigned)SM_NB) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gw = (int)me * SM_WPB + wave, nw = SM_NB * SM_WPB;
    const int k = P.k, m = P.m, n = P.n;
    const int PS = KP + KP * KP;                                   // stride of a workgroup's partial record
    unsigned gen = 0;
    const T eps = T(1e-15);

    // per-wave partial Gram of this wave's columns of X (lane r < KP: row r; the lanes above repeat) -> workgroup sum in LDS `dst`
    // in wave order -> part[me]; optional scaling of the columns by d (dsh) on the way, and the loss's cross term
    auto gram_pass = [&](T* X, int ncols, bool scale, const T* Bw, double* cross_out) {
        const int f = lane % KP;
        const bool fok = lane < k;
        T g[KP];
#pragma unroll
        for (int c = 0; c < KP; ++c) g[c] = T(0);
        const T dv = scale ? dsh[f] : T(1);
        double cacc = 0.0;
        if constexpr (KP == 16) {                                   // four columns abreast, one per 16-lane row (sm_half_update4's mapping)
            const int gr = lane >> 4;
            const bool f16 = f < k;
            for (int j0 = 4 * gw; j0 < ncols; j0 += 4 * nw) {
                const int j = j0 + gr;
                const bool cok = f16 && j < ncols;
                T x = cok ? sm_ldg(X + (int64_t)j * k + f) : T(0);
                if (scale) {
                    x = cok ? x / dv : T(0);
                    if (cok) X[(int64_t)j * k + f] = x;
                }
                if (Bw && cok) cacc += static_cast<double>(dv) * static_cast<double>(x) * static_cast<double>(sm_ldg(Bw + (int64_t)j * k + f));
                cd_static_for<0, KP>([&](auto IC) {
                    constexpr int c = decltype(IC)::value;
                    g[c] = tfma(x, sm_row_bcast<c>(x), g[c]);
                });
            }
#pragma unroll
            for (int c = 0; c < KP; ++c) { g[c] += shfl_xor_t(g[c], 16); g[c] += shfl_xor_t(g[c], 32); }
        } else
        for (int j = gw; j < ncols; j += nw) {
            T x = fok ? sm_ldg(X + (int64_t)j * k + lane) : T(0);
            if (scale) {
                x = fok ? x / dv : T(0);
                if (fok) X[(int64_t)j * k + lane] = x;
            }
            if (Bw && fok) cacc += static_cast<double>(dv) * static_cast<double>(x) * static_cast<double>(sm_ldg(Bw + (int64_t)j * k + lane));
            cd_static_for<0, KP>([&](auto IC) {
                constexpr int c = decltype(IC)::value;
                g[c] = tfma(x, lane_value(x, c), g[c]);
            });
        }
        if constexpr (KP == 16) {
            // every wave parks its partial tile in its own LDS slot, then one pass adds the slots in wave order (one barrier instead of
            // one per wave)
            if (lane < KP) {
#pragma unroll
                for (int c = 0; c < KP; ++c) wtile[wave * KP * KP + c * KP + lane] = g[c];
            }
            __syncthreads();
            for (int e = threadIdx.x; e < KP * KP; e += blockDim.x) {
                T sv = wtile[e];
                for (int w = 1; w < SM_WPB; ++w) sv += wtile[w * KP * KP + e];
                P.part[(size_t)me * PS + KP + e] = sv;
            }
            __syncthreads();
        } else {
            T* dst = Gwt;                                           // workgroup accumulator (every caller re-reads the sum from `part`)
            for (int w = 0; w < SM_WPB; ++w) {
                if (wave == w && lane < KP) {
#pragma unroll
                    for (int c = 0; c < KP; ++c) dst[c * KP + lane] = w == 0 ? g[c] : dst[c * KP + lane] + g[c];
                }
                __syncthreads();
            }
            for (int e = threadIdx.x; e < KP * KP; e += blockDim.x) P.part[(size_t)me * PS + KP + e] = dst[e];
        }
        if (cross_out) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) cacc += __shfl_xor(cacc, off, 64);
            if (lane == 0) red[wave] = cacc;
            __syncthreads();
            if (threadIdx.x == 0) {
                double s = 0.0;
                for (int w = 0; w < SM_WPB; ++w) s += red[w];
                cross_out[me] = s;
            }
        }
    };
    // after a barrier: the sum over the SM_NB workgroups' records of `count` consecutive values at offset `off` -> LDS dst[0 .. count).
    // Every load is independent (a thread owns one value of a group of workgroups and issues its loads back to back: one round trip to
    // the L2, not SM_NB dependent ones); the order of the additions is fixed -- inside a group ascending, then the groups ascending --
    // so every workgroup forms bitwise the same sums.
    auto part_sum = [&](int off, int count, T* dst) {
        constexpr int NT = 64 * SM_WPB;
        int groups = 1;                                             // a power of two <= min(NT / count, SM_NB); count in {16, 32, 256, 1024}
        while (2 * groups * count <= NT && 2 * groups <= SM_NB) groups *= 2;
        const int per = SM_NB / groups;                              // workgroups per group
        const int epp = count < NT ? count : NT;                     // values per pass (threads beyond groups * epp idle)
        for (int e0 = 0; e0 < count; e0 += epp) {
            const int e = e0 + (int)threadIdx.x % epp, h = (int)threadIdx.x / epp;
            if (e < count && h < groups) {
                T sacc = T(0);
                for (int b0 = 0; b0 < per; b0 += 8) {                 // eight loads in flight per thread
                    T v[8];
#pragma unroll
                    for (int b = 0; b < 8; ++b) v[b] = b0 + b < per ? sm_ldg(P.part + (size_t)(h * per + b0 + b) * PS + off + e) : T(0);
#pragma unroll
                    for (int b = 0; b < 8; ++b) sacc += v[b];          // (+ 0 beyond the group's last workgroup: exact)
                }
                red_t[h * epp + (e - e0)] = sacc;
            }
            __syncthreads();
            if (e < count && h == 0) {
                T sacc = red_t[e - e0];
                for (int hh = 1; hh < groups; ++hh) sacc += red_t[hh * epp + (e - e0)];
                dst[e] = sacc;
            }
            __syncthreads();
        }
    };
    // G = summed partial Grams + eps (+ l2) on the diagonal, identity padding -> LDS dst
    auto gram_sum = [&](T* dst, T l2) {
        part_sum(KP, KP * KP, dst);
        for (int e = threadIdx.x; e < KP * KP; e += blockDim.x) {
            T sv = dst[e];
            const int r = e % KP, c = e / KP;
            if (r == c) { if (r < k) { sv += eps; sv += l2; } else sv = T(1); }
            else if (r >= k || c >= k) sv = T(0);
            dst[e] = sv;
        }
        __syncthreads();
    };
    // workgroup sum of the waves' partial row norms -> part[me][0 .. KP)
    auto norm_store = [&](T nacc) {
        // (only the lanes below k hold a coordinate: the others carry zeros)
        if (lane < KP) wnorm[wave * KP + lane] = nacc;
        __syncthreads();
        if (threadIdx.x < KP) {
            T sv = wnorm[threadIdx.x];
            for (int w = 1; w < SM_WPB; ++w) sv += wnorm[w * KP + threadIdx.x];          // wave order
            P.part[(size_t)me * PS + threadIdx.x] = sv;
        }
        __syncthreads();
    };
    // after a barrier: d from the summed row norms (scaling_finalize above): dsh, and the caller's d
    auto norm_sum = [&]() {
        part_sum(0, KP, nsh);
        if (threadIdx.x < KP) {
            T sv = nsh[threadIdx.x];
            T dv = T(1);
            if (P.norm_type != 2) { if (P.norm_type == 1) sv = sqrt(sv); dv = sv + eps; }
            dsh[threadIdx.x] = dv;
            if (me == 0 && (int)threadIdx.x < k) P.d[threadIdx.x] = dv;
        }
        __syncthreads();
    };
    auto factor = [&](const T* G) {
        if constexpr (CHOL) {
            for (int e = threadIdx.x; e < KP * KP; e += blockDim.x) Lf[e] = G[e];
            __syncthreads();
            if (wave == 0) sm_chol_factor<T, KP>(Lf, lane);
            __syncthreads();
        }
    };

    // ---- prologue: Gram of the starting W_T
    gram_pass(P.W, m, false, nullptr, nullptr);
    if (!sm_barrier(P.sync, gen, &sh_flag)) return;
    gram_sum(Gh, P.L2_H);

    double prev_loss = sizeof(T) == 4 ? (double)3.402823466e+38f : 1.7976931348623157e308;
    int patience_counter = 0, iterations = 0, converged = 0;
    double final_tol = 0, train_loss = 0, last_loss = 0;
    // where workgroup 0's time goes, in ticks of the 100 MHz wall clock: the fused half-updates | waiting at the four barriers (= the
    // slowest workgroup's lead) | everything else; reported in result[5 .. 7]
    long long tk_half = 0, tk_bar = 0;
    const long long tk_start = wall_clock64();
    for (int iter = 0; iter < P.max_iter; ++iter) {
        const int warm = iter + P.iter0 > 0 ? 1 : 0;
        // ================= H half-update (fit_cpu.hpp:486-645)
        factor(Gh);
        T nacc;
        long long tk0 = wall_clock64();
        if (KP == 16 && n > nw)
            nacc = sm_half_update4<T, CHOL>(P.Ap, P.Ai, P.Ax, n, P.W, P.H, nullptr, k, Gh, Lf, P.L1_H, P.ub_H, P.nonneg_H, warm, P.cd_maxit, P.cd_tol,
                                            P.norm_type, gw, nw, lane);
        else
            nacc = sm_half_update<T, KP, CHOL>(P.Ap, P.Ai, P.Ax, n, P.W, P.H, nullptr, k, Gh, Lf, P.L1_H, P.ub_H, P.nonneg_H, warm, P.cd_maxit, P.cd_tol,
                                               P.norm_type, gw, nw, lane);
        tk_half += wall_clock64() - tk0;
        norm_store(nacc);
        tk0 = wall_clock64();
        if (!sm_barrier(P.sync, gen, &sh_flag)) return;
        tk_bar += wall_clock64() - tk0;
        norm_sum();                                                // :645 extract_scaling
        gram_pass(P.H, n, P.norm_type != 2, nullptr, nullptr);     // scaled H, and its Gram's partials
        tk0 = wall_clock64();
        if (!sm_barrier(P.sync, gen, &sh_flag)) return;
        tk_bar += wall_clock64() - tk0;
        // ================= W half-update (:711-893)
        gram_sum(Gsaved, T(0));                                    // :715-722 G_saved = gram(H) + eps
        for (int e = threadIdx.x; e < KP * KP; e += blockDim.x) { const int r = e % KP, c = e / KP; Gh[e] = Gsaved[e] + ((r == c && r < k) ? P.L2_W : T(0)); }
        __syncthreads();
        factor(Gh);
        tk0 = wall_clock64();
        if (KP == 16 && m > nw)
            nacc = sm_half_update4<T, CHOL>(P.Tp, P.Ti, P.Tx, m, P.H, P.W, P.Bw, k, Gh, Lf, P.L1_W, P.ub_W, P.nonneg_W, warm, P.cd_maxit, P.cd_tol,
                                            P.norm_type, gw, nw, lane);
        else
            nacc = sm_half_update<T, KP, CHOL>(P.Tp, P.Ti, P.Tx, m, P.H, P.W, P.Bw, k, Gh, Lf, P.L1_W, P.ub_W, P.nonneg_W, warm, P.cd_maxit, P.cd_tol,
                                               P.norm_type, gw, nw, lane);
        tk_half += wall_clock64() - tk0;
        norm_store(nacc);
        tk0 = wall_clock64();
        if (!sm_barrier(P.sync, gen, &sh_flag)) return;
        tk_bar += wall_clock64() - tk0;
        norm_sum();                                                // :893
        gram_pass(P.W, m, P.norm_type != 2, P.Bw, P.crossp);       // scaled W_T, its Gram's partials, the cross term's partials
        tk0 = wall_clock64();
        if (!sm_barrier(P.sync, gen, &sh_flag)) return;
        tk_bar += wall_clock64() - tk0;
        // ================= loss (:1729-1753), formed by every workgroup for itself: identical arithmetic, identical decision
        gram_sum(Gwt, T(0));
        {
            double acc = 0.0;
            for (int e = threadIdx.x; e < k * k; e += blockDim.x) {
                const int r = e % k, c = e / k;
                acc += static_cast<double>(dsh[r]) * static_cast<double>(dsh[c]) * static_cast<double>(Gwt[c * KP + r]) * static_cast<double>(Gsaved[c * KP + r]);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
            if (lane == 0) red[wave] = acc;
            __syncthreads();
            if (wave == 0) {
                double cross = lane < SM_NB ? sm_ldg(P.crossp + lane) : 0.0;          // one load per lane, then a fixed xor tree
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) cross += __shfl_xor(cross, off, 64);
                if (lane == 0) {
                    double recon = 0.0;
                    for (int w = 0; w < SM_WPB; ++w) recon += red[w];
                    sh_loss[0] = P.trAtA[0] - 2.0 * cross + recon;
                }
            }
            __syncthreads();
        }
        double loss_val = sh_loss[0];
        if (sizeof(T) == 4) loss_val = static_cast<double>(static_cast<float>(loss_val));
        last_loss = loss_val;
        if (me == 0 && threadIdx.x == 0 && P.loss_hist) P.loss_hist[iter] = loss_val;
        // the next H half-update's Gram = this loss Gram (+ L2_H): gram(W_T) + eps of the scaled W_T
        for (int e = threadIdx.x; e < KP * KP; e += blockDim.x) { const int r = e % KP, c = e / KP; Gh[e] = Gwt[e] + ((r == c && r < k) ? P.L2_H : T(0)); }
        __syncthreads();
        bool hit = false;
        if (iter > 0) {                                            // :1769-1775
            const double rel = fabs(prev_loss - loss_val) / (fabs(prev_loss) + 1e-15);
            final_tol = rel;
            hit = rel < P.tol;
        }
        prev_loss = loss_val;
        iterations = iter + 1;
        if (iter > 0) {                                            // :1797-1809
            if (hit) { if (++patience_counter >= P.patience) { converged = 1; train_loss = prev_loss; break; } }
            else patience_counter = 0;
        }
    }
    if (!converged) train_loss = last_loss;
    if (me == 0 && threadIdx.x == 0) {
        P.result[0] = iterations; P.result[1] = converged; P.result[2] = train_loss; P.result[3] = final_tol; P.result[4] = 1.0;
        P.result[5] = (double)tk_half; P.result[6] = (double)tk_bar; P.result[7] = (double)(wall_clock64() - tk_start);
    }
}

}  // namespace rk
