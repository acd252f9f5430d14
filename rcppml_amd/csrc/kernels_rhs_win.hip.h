// ============================================================================
// kernels_rhs_win.hip.h -- the SpMM-like right-hand side  B(:,j) = sum_i A(i,j) F(:,i)  (reference primitives/cpu/rhs.hpp:52-70,
// fused_nnls.hpp:109-114) with F streamed through a RING of LDS row tiles and the nonzeros scheduled over a sliding WINDOW.
//
// Round-4 successor of kernels_rhs_tiled.hip.h (same gather-from-LDS step: 1 v_add_u32_dpp + 1 ds_read_b128 + 1 v_mov_b32_dpp +
// 2 v_pk_fma_f32 per four nonzero slots and 256-byte slice; output columns stationary in registers).  What changed, and why:
//
//  * Ring instead of two slabs.  LDS holds NBUF = 4 tiles of 32 KiB.  In phase t the tiles t .. t+2 are complete and tile t+3
//    is landing (LDS-DMA), so a nonzero of tile u may be consumed in any of the phases u-2, u-1, u.  With one tile per phase
//    (r3) the number of nonzeros of a (column, tile) pair is Poisson(2.56) at C2 and a fixed S = 4 slots wastes 40 % of the
//    steps AND spills 7 % of the nonzeros; over a window of three tiles the planner (rw_fill_kernel: earliest-deadline-first
//    per column) evens the bursts out: 1.75 slots per phase and column -> fill 0.72 with 2 % spilled, 2 per phase -> fill
//    0.64 with 0.7 % spilled.
//  * Fractional slot rates.  A phase carries CLO slots per column, phases with (t & 3) < nhi one more (rate CLO + nhi/4):
//    two unrolled phase bodies, the choice is a wave-uniform branch per phase.
//  * No spill launch, no tail launch.  What the window cannot place (the "overflow", <= 2-3 %) is added by the kernel that
//    sums the row partitions (rhs_win_finish_kernel); columns beyond the last whole workgroup are ordinary (zero-padded)
//    columns of the last column block.
//  * Row partitions on both sides (P >= 2 when the matrix is large): fewer, longer phases per workgroup -- the per-phase
//    cost that is not slot steps (barrier skew, pipeline refill, LDS-DMA issue) is paid ntiles / P times.
//
// Slot stream (built once per fit): [column block][partition][phase][wave] blocks of sb_lo / sb_hi bytes; a block holds the
// values of its NR*C steps x 4 lane groups (step-major), then their u16 ring offsets (byte offset of the row inside the
// 128 KiB ring + the 1 KiB zero row in front of it, >> 4; 0 = an empty slot).  One or two 16-byte-per-lane LDS-DMA instructions fetch a block two phases ahead.
// Summation order per output element: partition by partition; inside a partition phase by phase in slot order (= row
// order); overflow nonzeros last, in row order.  Fixed, so results are deterministic run to run.
// ============================================================================
#pragma once
#include "kernels_rhs_tiled.hip.h"

namespace rk {

constexpr int RW_NBUF = 4;                   // ring buffers
constexpr int RW_W = RW_NBUF - 1;            // complete tiles visible in a phase
constexpr int RW_TB = 32768;                 // bytes per tile
constexpr int RW_RING = RW_NBUF * RW_TB;     // 128 KiB
constexpr int RW_ZROW = 1024;                // LDS bytes [0, 1024): a row of zeros -- what an empty slot (value 0, offset 0) reads, so
                                             // that padding never multiplies 0 with a row of F (0 x Inf would reach every column)
constexpr int RW_MAXCAND = 12;

struct RhsWinGeom {
    int64_t ncols;        // output columns
    int64_t nrows;        // rows of the sparse matrix = rows (k-vectors) of F
    int rowb;             // bytes per row of F
    int R;                // rows per tile = RW_TB / rowb
    int ntiles;           // ceil(nrows / R)
    int P;                // row partitions: tiles [p*ntiles/P, (p+1)*ntiles/P)
    int NW, nr;           // waves per workgroup, rounds per wave (4 columns each)
    int clo, nhi;         // slots per (column, phase): clo, one more in phases with (phase & 3) < nhi
    int ncb;              // column blocks (workgroups per partition)
    int sb_lo, sb_hi;     // bytes of a (wave, phase) slot block (multiples of 16)
    int maxph;            // phases of the longest partition
    int64_t region;       // bytes of slot stream per (column block, partition)
    int dbg;              // timing ablations (results are wrong): 1 = no compute, 2 = no tile copies, 4 = no slot copies
};

__host__ __device__ __forceinline__ int rw_hi_before(int t, int nhi) { return (t >> 2) * nhi + ((t & 3) < nhi ? (t & 3) : nhi); }
__host__ __device__ __forceinline__ int rw_cap(const RhsWinGeom& G, int t) { return G.clo + (((t & 3) < G.nhi) ? 1 : 0); }
__host__ __device__ __forceinline__ int rw_t0(const RhsWinGeom& G, int p) { return (int)((int64_t)G.ntiles * p / G.P); }
// byte offset of the (phase, wave) block inside a (column block, partition) region
__host__ __device__ __forceinline__ int64_t rw_block_off(const RhsWinGeom& G, int t, int w) {
    const int hb = rw_hi_before(t, G.nhi);
    const int sb = ((t & 3) < G.nhi) ? G.sb_hi : G.sb_lo;
    return (int64_t)G.NW * ((int64_t)t * G.sb_lo + (int64_t)hb * (G.sb_hi - G.sb_lo)) + (int64_t)w * sb;
}

// ---------------------------------------------------------------------------
// Planner walk: one THREAD per (column, row partition) -- 200 000 independent walks of ~100 nonzeros on C2's two sides alike;
// one thread per column left the 20 000 columns of C2's transpose to 79 workgroups.  Nonzeros in row order, earliest-
// deadline-first into the phases of the window.  Rows must ascend inside a column (CSC invariant): the partition's run of
// nonzeros is found by bisection, and a row outside the partition's range or a descent raises `unsorted`.
// place(e, t, rank) for a scheduled nonzero (t = phase inside the partition, rank = its slot among the phase's cap),
// spill(e) for one that no phase of its window can take.
// q = the column's round inside its wave, ub = steps of the first batch of a phase: step q*cap + rank < ub is read from LDS
// BEFORE the barrier that publishes the newest tile of the window, so such a slot only takes nonzeros of the older tiles.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int rw_lower_bound(const int* __restrict__ rowidx, int lo, int hi, int row) {     // first e in [lo, hi) with rowidx[e] >= row
    while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);          /* (lo + hi overflows an int above 2^30 nonzero positions: configs[3] at full extent) */
        if (rowidx[mid] < row) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}
template <class Place, class Spill>
__device__ __forceinline__ void rw_walk_part(const int* __restrict__ rowidx, int start, int end, int R, int ntiles, int P, int p,
                                             int clo, int nhi, int q, int ub, int* unsorted, Place place, Spill spill) {
    const int t0 = (int)((int64_t)ntiles * p / P), t1 = (int)((int64_t)ntiles * (p + 1) / P);
    const int s = p == 0 ? start : rw_lower_bound(rowidx, start, end, t0 * R);
    const int e1 = p == P - 1 ? end : rw_lower_bound(rowidx, start, end, t1 * R);
    int cur = 0, used = 0, prow = -1;
    for (int e = s; e < e1; ++e) {
        const int row = rowidx[e];
        const int ta = row / R;
        if (row < prow || ta < t0 || ta >= t1 || row < 0) {                     // not a sorted column
            if (unsorted) *unsorted = 1;
            return;
        }
        prow = row;
        const int u = ta - t0;
        const int lo = u - (RW_W - 1) > 0 ? u - (RW_W - 1) : 0;
        if (cur < lo) { cur = lo; used = 0; }
        for (;;) {
            if (cur > u) { spill(e); break; }
            const int cap = clo + (((cur & 3) < nhi) ? 1 : 0);
            if (used >= cap) { ++cur; used = 0; continue; }
            if (q * cap + used < ub && u > cur + RW_W - 2) { ++used; continue; }       // a first-batch slot: this tile is too new for it
            place(e, cur, used);
            ++used;
            break;
        }
    }
    if (e1 < s && unsorted) *unsorted = 1;
}

struct RwCand { int n; int clo[RW_MAXCAND]; int nhi[RW_MAXCAND]; };

__device__ __forceinline__ int rw_round_of(int64_t j, int nr, int NW) {          // the round (0 .. nr-1) column j occupies in its wave
    const int cpw = 4 * nr;
    return (int)((j % ((int64_t)cpw * NW)) % cpw) >> 2;
}
__host__ __device__ __forceinline__ int rw_ub(int rowb) { return rowb == 256 ? 4 : (rowb == 512 ? 2 : 1); }   // steps of a phase's first batch

// overflow count of every candidate rate on a SAMPLE of the columns (every `stride`-th): chooses the rate; [RW_MAXCAND] = nonzeros seen
static __global__ __launch_bounds__(256) void rw_survey_kernel(const int* __restrict__ colptr, const int* __restrict__ rowidx,
                                                        int64_t ncols, int64_t stride, int R, int ntiles, int P, int nr, int NW, int ub,
                                                        RwCand C, unsigned long long* __restrict__ ovf) {
    __shared__ unsigned long long sh[RW_MAXCAND + 1];
    if (threadIdx.x <= RW_MAXCAND) sh[threadIdx.x] = 0;
    __syncthreads();
    // one thread per (sampled column, partition, candidate): the walks are serial, so the candidates run side by side
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c = (int)(idx % C.n);
    const int64_t seg = idx / C.n;
    const int64_t j = (seg / P) * stride;
    const int p = (int)(seg % P);
    if (j < ncols) {
        const int start = colptr[j], end = colptr[j + 1];
        const int q = rw_round_of(j, nr, NW);
        unsigned cnt = 0, seen = 0;
        rw_walk_part(rowidx, start, end, R, ntiles, P, p, C.clo[c], C.nhi[c], q, ub, nullptr,
                     [&](int, int, int) { ++seen; }, [&](int) { ++cnt; ++seen; });
        if (cnt) atomicAdd(&sh[c], (unsigned long long)cnt);
        if (c == 0 && seen) atomicAdd(&sh[RW_MAXCAND], (unsigned long long)seen);
    }
    __syncthreads();
    if (threadIdx.x <= RW_MAXCAND && sh[threadIdx.x]) atomicAdd(&ovf[threadIdx.x], sh[threadIdx.x]);
}

// overflow nonzeros per (column, partition) for the chosen rate (feeds the exclusive scan that makes the overflow pointers);
// also the sortedness check of the whole matrix
static __global__ __launch_bounds__(256) void rw_ovcount_kernel(const int* __restrict__ colptr, const int* __restrict__ rowidx,
                                                         RhsWinGeom G, int* __restrict__ ovcnt, int* __restrict__ unsorted) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t j = idx / G.P;
    const int p = (int)(idx % G.P);
    if (j >= G.ncols) return;
    int cnt = 0;
    rw_walk_part(rowidx, colptr[j], colptr[j + 1], G.R, G.ntiles, G.P, p, G.clo, G.nhi, rw_round_of(j, G.nr, G.NW), rw_ub(G.rowb),
                 unsorted, [](int, int, int) {}, [&](int) { ++cnt; });
    ovcnt[idx] = cnt;
}

// scatter the nonzeros into the slot stream / the overflow lists (ovptr: one entry per (column, partition), column-major).
// vals == NULL: the index half of a DEFERRED plan -- offsets and overflow rows now, and for every nonzero where its value will
// go (dest[e]: element index into the slot stream, or 0x80000000 | position in the overflow list), so that the values can be
// scattered by a coalesced pass (rw_values_kernel) once they have arrived (the plugin builds plans while they cross PCIe).
template <class T>
__global__ __launch_bounds__(256) void rw_fill_kernel(const int* __restrict__ colptr, const int* __restrict__ rowidx,
                                                      const T* __restrict__ vals, RhsWinGeom G, char* __restrict__ slots,
                                                      const int* __restrict__ ovptr, int* __restrict__ ovrow, T* __restrict__ ovval,
                                                      unsigned* __restrict__ dest) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t j = idx / G.P;
    const int p = (int)(idx % G.P);
    if (j >= G.ncols) return;
    const int cpw = 4 * G.nr, cpb = cpw * G.NW;
    const int64_t cb = j / cpb;
    const int jj = (int)(j - cb * cpb);
    const int w = jj / cpw, jw = jj - w * cpw;
    const int q = jw >> 2, g = jw & 3;
    int ob = ovptr[idx];
    const int t0 = rw_t0(G, p);
    const int64_t reg_off = (cb * G.P + p) * G.region;
    rw_walk_part(rowidx, colptr[j], colptr[j + 1], G.R, G.ntiles, G.P, p, G.clo, G.nhi, q, rw_ub(G.rowb), nullptr,
                 [&](int e, int t, int rank) {
                     const int cap = rw_cap(G, t);
                     const int nst = G.nr * cap;
                     const int64_t boff = reg_off + rw_block_off(G, t, w);
                     char* blk = slots + boff;
                     const int sidx = (q * cap + rank) * 4 + g;
                     const int row = rowidx[e];
                     const int ta = row / G.R;
                     const unsigned off = (unsigned)RW_ZROW + (unsigned)((ta - t0) & (RW_NBUF - 1)) * (unsigned)RW_TB + (unsigned)(row - ta * G.R) * (unsigned)G.rowb;
                     if (vals) reinterpret_cast<T*>(blk)[sidx] = vals[e];
                     else dest[e] = (unsigned)(boff / (int64_t)sizeof(T)) + (unsigned)sidx;
                     reinterpret_cast<uint16_t*>(blk + (size_t)nst * 4 * sizeof(T))[sidx] = (uint16_t)(off >> 4);
                 },
                 [&](int e) {
                     ovrow[ob] = rowidx[e];
                     if (vals) ovval[ob] = vals[e];
                     else dest[e] = 0x80000000u | (unsigned)ob;
                     ++ob;
                 });
}
// the value half of a deferred plan: one thread per nonzero, coalesced reads
template <class T>
__global__ __launch_bounds__(256) void rw_values_kernel(const T* __restrict__ vals, const unsigned* __restrict__ dest, int64_t nnz,
                                                        T* __restrict__ slots, T* __restrict__ ovval) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= nnz) return;
    const unsigned d = dest[e];
    if (d & 0x80000000u) ovval[d & 0x7fffffffu] = vals[e];
    else slots[d] = vals[e];
}

__device__ __forceinline__ const char* rw_uniform_ptr(const char* p) {         // tell hipcc the pointer is wave-uniform (SGPR pair)
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}
// 16 bytes per lane global -> LDS for the lanes with lane_off < limit (run-time limit, no branch)
__device__ __forceinline__ void rw_glds16_lim(const char* base_uniform, unsigned lane_off, unsigned lds_addr_uniform, unsigned limit) {
    unsigned long long keep;
    asm volatile("s_mov_b32 m0, %3\n\tv_cmp_gt_u32 vcc, %4, %1\n\ts_and_saveexec_b64 %0, vcc\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(base_uniform), "s"(lds_addr_uniform), "s"(limit) : "vcc", "scc");
}
// the same with the non-temporal hint: the slot stream is read ONCE, by one workgroup -- it should not displace the rows of F that
// every workgroup of the partition re-reads from the XCD's L2 (MI355X_MICROARCH.md "nt-weights": -18 % issue-to-landed for a stream
// one CU reads once)
__device__ __forceinline__ void rw_glds16_lim_nt(const char* base_uniform, unsigned lane_off, unsigned lds_addr_uniform, unsigned limit) {
    unsigned long long keep;
    asm volatile("s_mov_b32 m0, %3\n\tv_cmp_gt_u32 vcc, %4, %1\n\ts_and_saveexec_b64 %0, vcc\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b64 exec, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(base_uniform), "s"(lds_addr_uniform), "s"(limit) : "vcc", "scc");
}

// ---------------------------------------------------------------------------
// The kernel.  NV = 256-byte slices per row of F, CLO / NHI = slot rate (phase t carries CLO + ((t & 3) < NHI) slots per
// column), NR rounds per wave, NW waves per workgroup.  grid = ncb * P workgroups (blockIdx % P = partition: with P | 8 a
// whole XCD streams one partition of F), 128 KiB ring + two stages of slot blocks of dynamic LDS.
// Phase t: [slots(t) already in registers] [compute: LDS reads of batch b+1 issued before the FMAs of batch b; between the
// batches one LDS-DMA piece of tile t+3 / of slots(t+2)] [vmcnt: tile t+3 and slots(t+1) landed] [slots(t+1) -> registers]
// [barrier].  The loop body is FOUR phases, each with its compile-time step count: one instruction stream, no if / else
// between two phase bodies (with a run-time choice hipcc gave the accumulators different registers in the two bodies and
// moved all of them at every phase), and every slot-stream address is the group's base plus a constant.
// ---------------------------------------------------------------------------
template <class T, int NR, int C>
struct RwBlk { static constexpr int bytes = (NR * C * 4 * ((int)sizeof(T) + 2) + 15) & ~15; };

template <class T, int NV, int CLO, int NHI, int NR, int NW>
__global__ __launch_bounds__(64 * NW) void rhs_win_kernel(const char* __restrict__ slots, const T* __restrict__ F, RhsWinGeom G,
                                                          T* __restrict__ Bout) {
    typedef typename RtVec<T>::type V;
    constexpr int VN = RtVec<T>::N;
    constexpr int CHI = CLO + 1;
    constexpr int NSTMAX = NR * CHI;
    constexpr int NB = (NSTMAX + 15) / 16;      // slot registers (value, offset) per lane and phase
    constexpr int UB = NV == 1 ? 4 : (NV == 2 ? 2 : 1);     // steps per batch of LDS reads; two batches in flight
    constexpr int NCH = RW_TB / 1024;           // KiB-chunks per tile
    constexpr int CBASE = NCH / NW, CEXTRA = NCH % NW;
    constexpr int NPF = CBASE + (CEXTRA > 0 ? 1 : 0);
    constexpr int SBL = RwBlk<T, NR, CLO>::bytes, SBH = RwBlk<T, NR, CHI>::bytes;     // bytes of a "lo" / "hi" slot block
    constexpr int NSD = (SBH + 1023) / 1024;    // LDS-DMA instructions per slot block
    constexpr int GSTRIDE = NW * (4 * SBL + NHI * (SBH - SBL));      // slot bytes of a group of four phases (all waves)
    extern __shared__ char rt_slab[];           // ring of F tiles + the two-stage slot ring
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, u = lane & 15;
    const int p = blockIdx.x % G.P;
    const int64_t cb = blockIdx.x / G.P;
    const int t0 = rw_t0(G, p), t1 = rw_t0(G, p + 1);
    const int nph = t1 - t0, nph4 = (nph + 3) & ~3;
    [[maybe_unused]] const int dbg = G.dbg;
    const int64_t col0 = (cb * NW + w) * (int64_t)(4 * NR);       // first column of this wave
    const int k = G.rowb / (int)sizeof(T);
    const int64_t fbytes = G.nrows * (int64_t)G.rowb;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)rt_slab;
    const char* sreg = rw_uniform_ptr(slots + (cb * G.P + p) * G.region);

    V acc[NR][NV];
#pragma unroll
    for (int q = 0; q < NR; ++q)
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int e = 0; e < VN; ++e) acc[q][v][e] = T(0);

    // this wave's run of KiB-chunks of every tile: CBASE of them, one more for the first CEXTRA waves
    const int cstart = w * CBASE + (w < CEXTRA ? w : CEXTRA);
    const unsigned choff = (unsigned)cstart * 1024u + (unsigned)lane * 16u;
    const unsigned xpiece = (unsigned)((CEXTRA > 0 && w < CEXTRA) ? CBASE : (CBASE > 0 ? CBASE - 1 : 0)) * 1024u;
    const unsigned m0keep = rt_m0_save();
    // Tiles.  Tile tr of the partition goes to ring buffer tr & 3.  Past the end of the partition the NEXT tiles of F are
    // fetched (nobody reads them; their buffer is free), past the end of F the last tile again; only the last tile of F can
    // be short: its source offsets are clamped.  Everything per-phase is a handful of scalar instructions.
    const char* Fp = rw_uniform_ptr(reinterpret_cast<const char*>(F) + (int64_t)t0 * RW_TB);
    const int last_tr = G.ntiles - 1 - t0;                                      // last tile of F, relative to this partition
    const unsigned short_lim = (unsigned)(fbytes - (int64_t)(G.ntiles - 1) * RW_TB) - 16u;
    const char* nx_src = Fp; unsigned nx_lim = RW_TB - 16u, nx_dst = 0;
    auto tile_set = [&](int tr, int buf) {
        const int te = tr < last_tr ? tr : last_tr;
        nx_src = Fp + ((int64_t)te << 15);
        static_assert(RW_TB == 32768, "tile_set shifts by 15");
        nx_lim = te == last_tr ? short_lim : (unsigned)RW_TB - 16u;
        nx_dst = lds0 + RW_ZROW + (unsigned)buf * RW_TB + cstart * 1024;
    };
    auto tile_piece = [&](int pi) {
        const unsigned po = pi < CBASE ? (unsigned)pi * 1024u : xpiece;
        unsigned o = choff + po;
        o = o < nx_lim ? o : nx_lim;
#ifdef RW_ABLATE
        rw_glds16_lim(nx_src, o, nx_dst + po, (dbg & 2) ? 0u : 0xffffffffu);
#else
        rt_glds16_nc(nx_src, o, nx_dst + po);
#endif
    };
    const unsigned sl0 = lds0 + RW_ZROW + RW_RING + w * SBH;                    // this wave's slot area of stage 0
    char* const slp = rt_slab + RW_ZROW + RW_RING + w * SBH;
    constexpr int STAGE = NW * SBH;
    const int wlo = w * SBL, whi = w * SBH;
    // slot block of phase 4 grp + K for this wave: group base + compile-time offsets
    auto slot_piece = [&](const char* grp, auto KK, auto ST, auto II, bool live) {
        constexpr int K = decltype(KK)::value;
        constexpr int i = decltype(II)::value;
        constexpr int stage = decltype(ST)::value;
        constexpr int SB = K < NHI ? SBH : SBL;
        constexpr int pre = NW * (K * SBL + (K < NHI ? K : NHI) * (SBH - SBL));
        if constexpr (SB - 1024 * i > 0) {
#ifdef RW_ABLATE
            const unsigned left = (live && !(dbg & 4)) ? SB - 1024 * i : 0;
#else
            const unsigned left = live ? SB - 1024 * i : 0;
#endif
#ifdef RW_SLOTS_DEFAULT_POLICY
            rw_glds16_lim(grp + (pre + 1024 * i) + (K < NHI ? whi : wlo), 16u * lane, sl0 + stage * STAGE + 1024 * i, left);
#else
            rw_glds16_lim_nt(grp + (pre + 1024 * i) + (K < NHI ? whi : wlo), 16u * lane, sl0 + stage * STAGE + 1024 * i, left);
#endif
        }
    };
    // the wave's slots of one stage -> registers: the kernel wants step 16 b + u of lane group g in lane 16 g + u, i.e. slot
    // (16 b + u) * 4 + g.  Two per-lane base addresses for the whole kernel; stage, block b and the start of the offsets are
    // immediates of the reads.  Lanes past the last step of a block read whatever follows it (never consumed; the LDS
    // allocation carries 1 KiB of slack behind the last stage).
    char* const pv = slp + ((lane & 15) * 4 + (lane >> 4)) * (int)sizeof(T);
    char* const po = slp + ((lane & 15) * 4 + (lane >> 4)) * 2;
    auto slots_read = [&](auto ST, auto CC, T (&cv)[NB], unsigned (&co)[NB]) {
        constexpr int stage = decltype(ST)::value;
        constexpr int nst = NR * decltype(CC)::value;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            cv[b] = *reinterpret_cast<const T*>(pv + stage * STAGE + b * 64 * (int)sizeof(T));
            co[b] = (unsigned)*reinterpret_cast<const uint16_t*>(po + stage * STAGE + nst * 4 * (int)sizeof(T) + b * 128) << 4;
        }
    };
    // Two sets of slot registers: phase t consumes set t & 1 and fills the other one MID-phase with slots(t+1) (this wave's own
    // LDS-DMA data: a vmcnt wait, no barrier), so no LDS round trip sits between the barrier and the first step.
    T cv[2][NB];
    unsigned co[2][NB];
    V f[2][UB][NV];
    // batch 0 of a phase is PRE-ISSUED before the barrier that ends the previous phase (the planner keeps the newest tile out
    // of the first UB steps of every phase), so the LDS pipeline is full when the barrier opens
    int lbase = (lane & 15) * 16;                   // re-materialised once per phase (see slots_read), NOT per batch: a fresh VALU
                                                    // result in front of every batch's first DPP add costs two instructions and the
                                                    // DPP hazard's wait states, nine times per phase
    auto reads = [&](auto QB, auto CC, const unsigned (&cox)[NB]) {
        constexpr int b = decltype(QB)::value;
        constexpr int NST = NR * decltype(CC)::value;
#ifdef RW_ASM_READS
        // LDS reads hipcc does not see: no per-register lgkmcnt bookkeeping (it waits in front of every FMA pair), one counted
        // wait per batch (rt_wait_lgkm in the phase) with a scheduling barrier behind it, because nothing else ties the FMAs
        // to that wait.  The batch's addresses first, each in its own register: re-using one address register behind a read
        // that still holds it costs hazard wait states.
        int ad[UB];
        rt_static_for<0, UB>([&](auto I) {
            constexpr int i = decltype(I)::value;
            constexpr int step = b * UB + i;
            if constexpr (step < NST) ad[i] = lbase + rt_bc<(step & 15)>((int)cox[step >> 4]);
        });
        rt_static_for<0, UB>([&](auto I) {
            constexpr int i = decltype(I)::value;
            constexpr int step = b * UB + i;
            if constexpr (step < NST) {
                rt_static_for<0, NV>([&](auto VV) {
                    constexpr int v = decltype(VV)::value;
                    V got;
                    const int aa = ad[i];
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(got) : "v"(aa), "n"(256 * v));
                    f[b & 1][i][v] = got;
                });
            }
        });
#else
        rt_static_for<0, UB>([&](auto I) {
            constexpr int i = decltype(I)::value;
            constexpr int step = b * UB + i;
            if constexpr (step < NST) {
                const int a = lbase + rt_bc<(step & 15)>((int)cox[step >> 4]);
#pragma unroll
                for (int v = 0; v < NV; ++v) f[b & 1][i][v] = *reinterpret_cast<const V*>(rt_slab + a + 256 * v);
            }
        });
#endif
    };
    // one phase: K = t & 3 (compile time); sgrp = slot-stream base of the group of phase t
    auto phase = [&](auto KK, int t, const char* sgrp) {
        constexpr int K = decltype(KK)::value;
        constexpr int C = CLO + (K < NHI ? 1 : 0);
        constexpr int NST = NR * C;
        constexpr int NBATCH = (NST + UB - 1) / UB;
        constexpr int K1 = (K + 1) & 3, K2 = (K + 2) & 3;
        constexpr int C1 = CLO + (K1 < NHI ? 1 : 0);
        asm volatile("" : "+v"(lbase));
        constexpr int S = K & 1;                            // slot register set of this phase
        constexpr int MID = NBATCH / 2;                     // the batch after which slots(t+1) go to registers
        constexpr int NSDK = ((K2 < NHI ? SBH : SBL) + 1023) / 1024;      // LDS-DMA pieces of the slot block fetched in this phase
        constexpr int NPTK = NPF + NSDK;
        tile_set(t + RW_W, (K + RW_W) & (RW_NBUF - 1));
        const bool slot_live = t + 2 < nph4;
        const char* sg2 = K + 2 >= 4 ? sgrp + GSTRIDE : sgrp;
        auto issue_piece = [&](auto PI) {
            constexpr int pi = decltype(PI)::value;
            if constexpr (pi < NPF) tile_piece(pi);
            else slot_piece(sg2, std::integral_constant<int, K2>{}, std::integral_constant<int, (K & 1)>{}, std::integral_constant<int, pi - NPF>{}, slot_live);
        };
        // pieces issued up to and including batch b: those with pi * NBATCH / NPT <= b
        auto next_slots = [&](auto NISSUED) {
            rt_wait_vm<decltype(NISSUED)::value>();          // everything older than this phase's own pieces: slots(t+1) have landed
            slots_read(std::integral_constant<int, (K1 & 1)>{}, std::integral_constant<int, C1>{}, cv[S ^ 1], co[S ^ 1]);
        };
#ifdef RW_NO_COMPUTE
        if constexpr (true) {
#else
        if constexpr (NST == 0) {
#endif
            rt_static_for<0, NPTK>([&](auto PI) { issue_piece(PI); });
            next_slots(std::integral_constant<int, NPTK>{});
        } else {
            rt_static_for<0, NBATCH>([&](auto QB) {
                constexpr int b = decltype(QB)::value;
                constexpr int nnext = (b + 1 < NBATCH) ? ((NST - (b + 1) * UB) < UB ? (NST - (b + 1) * UB) : UB) * NV : 0;
                if constexpr (b + 1 < NBATCH) reads(std::integral_constant<int, b + 1>{}, std::integral_constant<int, C>{}, co[S]);
                __builtin_amdgcn_sched_barrier(0);
                rt_wait_lgkm<nnext>();                  // LDS returns in order: everything but the reads just issued is back
#ifdef RW_ASM_READS
                __builtin_amdgcn_sched_barrier(0);
#endif
                rt_static_for<0, UB>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    constexpr int step = b * UB + i;
                    if constexpr (step < NST) {
                        const T val = rt_bcast_val<(step & 15)>(cv[S][step >> 4]);
#pragma unroll
                        for (int v = 0; v < NV; ++v)
#pragma unroll
                            for (int e = 0; e < VN; ++e) acc[step / C][v][e] = rt_fma(val, f[b & 1][i][v][e], acc[step / C][v][e]);
                    }
                });
                __builtin_amdgcn_sched_barrier(0);
#ifdef RW_PIECES_EARLY
                if constexpr (b == 0) rt_static_for<0, NPTK>([&](auto PI) { issue_piece(PI); });
                if constexpr (b == MID) next_slots(std::integral_constant<int, NPTK>{});
#else
                rt_static_for<0, NPTK>([&](auto PI) {
                    if constexpr (decltype(PI)::value * NBATCH / NPTK == b) issue_piece(PI);
                });
                if constexpr (b == MID) {
                    constexpr int nissued = ((b + 1) * NPTK + NBATCH - 1) / NBATCH;      // pieces with pi * NBATCH / NPTK <= b
                    next_slots(std::integral_constant<int, (nissued < NPTK ? nissued : NPTK)>{});
                }
#endif
            });
        }
        // every FMA of this phase is done HERE (keeps hipcc from rotating the tail of the phase below the barrier)
#pragma unroll
        for (int q = 0; q < NR; ++q)
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("" : "+v"(acc[q][v]));
        // batch 0 of the next phase (it may only touch tiles that were complete during this phase), then publish tile t+3
#pragma unroll
        for (int b = 0; b < NB; ++b) asm volatile("" : "+v"(cv[S ^ 1][b]), "+v"(co[S ^ 1][b]));
#ifndef RW_NO_COMPUTE
        reads(std::integral_constant<int, 0>{}, std::integral_constant<int, C1>{}, co[S ^ 1]);
#endif
        rt_wait_vm<NSDK>();                      // all but the slot pieces just issued: tile t+3 has landed
#ifndef RW_NO_BARRIER
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
    };

    if (nph > 0) {
        if (threadIdx.x < RW_ZROW / 16) *reinterpret_cast<V*>(rt_slab + threadIdx.x * 16) = acc[0][0];      // the zero row (acc is still 0)
        // prologue: the first RW_W tiles and the first two slot blocks
        for (int tr = 0; tr < RW_W; ++tr) {
            tile_set(tr, tr);
#pragma unroll
            for (int pi = 0; pi < NPF; ++pi)
                if (pi < CBASE || w < CEXTRA) tile_piece(pi);
        }
        rt_static_for<0, NSD>([&](auto I) { slot_piece(sreg, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, I, true); });
        rt_static_for<0, NSD>([&](auto I) { slot_piece(sreg, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, I, true); });
        rt_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        slots_read(std::integral_constant<int, 0>{}, std::integral_constant<int, CLO + (0 < NHI ? 1 : 0)>{}, cv[0], co[0]);
        reads(std::integral_constant<int, 0>{}, std::integral_constant<int, CLO + (0 < NHI ? 1 : 0)>{}, co[0]);
        const char* sgrp = sreg;
        // whole groups of four phases: a partition whose phase count is not a multiple of four runs up to three phases of
        // empty slots at its end (the planner leaves them zero) -- early exits from the middle of the loop body made hipcc
        // keep the accumulators of the exit paths in scratch
        for (int t = 0; t < nph; t += 4, sgrp += GSTRIDE) {
            phase(std::integral_constant<int, 0>{}, t, sgrp);
            phase(std::integral_constant<int, 1>{}, t + 1, sgrp);
            phase(std::integral_constant<int, 2>{}, t + 2, sgrp);
            phase(std::integral_constant<int, 3>{}, t + 3, sgrp);
        }
        rt_wait_lgkm<0>();                         // the reads pre-issued by the last phase
    }
    rt_wait_vm<0>();                               // nothing may still be landing in LDS when the workgroup retires
    rt_m0_restore(m0keep);

    const int64_t ncp = (int64_t)G.ncb * (4 * NR * NW);          // padded column count of a partition slab
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const int64_t j = col0 + 4 * q + g;
        if (j < G.ncols) {
            T* dst = Bout + ((G.P > 1 ? (int64_t)p * ncp : 0) + j) * k;
#pragma unroll
            for (int v = 0; v < NV; ++v) *reinterpret_cast<V*>(dst + (64 * v + 4 * u) * 4 / (int)sizeof(T)) = acc[q][v];
        }
    }
}

}  // namespace rk
