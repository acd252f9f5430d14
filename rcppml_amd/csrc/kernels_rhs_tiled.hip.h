// ============================================================================
// kernels_rhs_tiled.hip.h -- the SpMM-like right-hand side  B(:,j) = sum_i A(i,j) F(:,i)  with the factor staged
// through LDS in row tiles (reference primitives/cpu/rhs.hpp:52-70, fused_nnls.hpp:109-114).
//
// Why: rhs_stage_kernel gathers one k-row of F per nonzero through the vector-memory path (nnz*k*s_v = 5.1 GB per
// launch at C2) and is bound by the texture-addresser / L1 delivery rate (17 TB/s measured, 23 TB/s with every access
// an L1 hit).  LDS delivers 256 B/clk/CU for ds_read_b128 -- four times the L1 rate -- and needs no address coalescing.
//
// Mapping.  The rows of F are cut into tiles of R = 64 KiB / row-bytes rows; a tile is a CONTIGUOUS 64 KiB block of F
// (k is the leading dimension) and is copied into LDS by LDS-DMA (global_load_lds_dwordx4, double-buffered: tile t+1
// lands while tile t is consumed).  The output columns are STATIONARY in registers: a wavefront owns 4*nr columns for
// the whole kernel, lane group g (16 lanes = one DPP row) owns column 4q+g of round q and holds 16 bytes of it per
// 256-byte slice of the row, so B never leaves the register file until the epilogue and every CU streams all of its
// partition of F exactly once (workgroups * |F| bytes of L2 -> LDS traffic instead of nnz * row-bytes of gathers).
//
// The nonzeros come from a tile-partitioned, slot-padded copy of A built once per fit (rhs_tiled_fill_kernel): for
// every (column, tile) pair exactly S slots {byte offset of the row inside the tile (u16), value}; pairs with more
// than S nonzeros spill the rest into an overflow CSC that the gather kernel handles first (its result seeds the
// accumulators), empty slots hold (0, 0).  The slots of one wave and tile are contiguous, ordered step-major, so 16
// steps x 4 lane groups are ONE coalesced load; step i of the block reaches its lane group by DPP row_newbcast:i folded
// into the consuming v_add_u32 (LDS address) -- no LDS crossbar, no scalar traffic, no per-column control flow.
// Per step (4 nonzero slots): 1 v_add_u32_dpp + 1 ds_read_b128 + 1 v_mov_b32_dpp + 2 v_pk_fma_f32 per 256-byte slice.
//
// Summation order per output element: overflow nonzeros (column order), then tile by tile in row order -- fixed, so
// the result is deterministic run to run.  With P > 1 row partitions (the W side, whose factor H does not fit an
// XCD's L2: partition p is streamed by the workgroups with blockIdx % P == p, i.e. by whole XCDs) each partition
// writes its own slab and rhs_tiled_reduce_kernel adds them in partition order.
// ============================================================================
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace rk {

constexpr int RT_SLAB_BYTES = 65536;      // one LDS tile of F; two of them are resident
constexpr int RT_MAX_HIST = 32;

struct RhsTiledGeom {
    int64_t ncols;        // columns of the sparse matrix = output columns
    int64_t ncols_tiled;  // columns [0, ncols_tiled) go through the tiled kernel, the rest through the gather kernel
    int64_t nrows;        // rows of the sparse matrix = rows (k-vectors) of F
    int rowb;             // bytes per row of F (k * sizeof(T)), 256 or 512 ... multiple of 256
    int rshift;           // log2(rows per tile), rows per tile = 65536 / rowb
    int ntiles;           // ceil(nrows / R)
    int P;                // row partitions (tiles [p*ntiles/P, (p+1)*ntiles/P))
    int NW;               // waves per workgroup
    int nr;               // rounds per wave (4 columns each)
    int S;                // slots per (column, tile)
    int ncb;              // column blocks = workgroups per partition
    int dbg;              // -DRCPPML_EXPERIMENTS builds only: 1 = no compute, 2 = no LDS-DMA, 4 = no slot loads
};

__device__ __forceinline__ int64_t rt_slot_index(const RhsTiledGeom& G, int64_t j, int tile, int rank) {   // j < G.ncols_tiled
    const int cpw = 4 * G.nr;                 // columns per wave
    const int cpb = cpw * G.NW;               // columns per workgroup
    const int64_t cb = j / cpb;
    const int jj = (int)(j - cb * cpb);
    const int w = jj / cpw, jw = jj - w * cpw;
    const int q = jw >> 2, g = jw & 3;
    const int step = q * G.S + rank;
    return (((cb * G.ntiles + tile) * G.NW + w) * (int64_t)(G.nr * G.S) + step) * 4 + g;
}

// ---------------------------------------------------------------------------
// Walk one column with one wavefront: every nonzero learns its tile and its rank inside the (column, tile) segment.
// Rows must be sorted inside the column (CSC invariant); `unsorted` is raised otherwise.
// fn(e, row, tile, rank, ovf_pos) is called with all lanes converged; lanes without a nonzero get e < 0.
// ovf_pos = running count of nonzeros of this column with rank >= S before this one (meaningful when rank >= S).
// ---------------------------------------------------------------------------
template <class Fn>
__device__ __forceinline__ void rt_walk_column(const int* __restrict__ rowidx, int start, int end, int rshift, int S,
                                               int* unsorted, Fn fn) {
    const int lane = threadIdx.x & 63;
    int carry = 0;          // nonzeros of the segment that runs into this chunk, seen in earlier chunks
    int ovbase = 0;
    for (int e0 = start; e0 < end; e0 += 64) {
        const int e = e0 + lane;
        const bool valid = e < end;
        const int row = valid ? rowidx[e] : 0x7fffffff;
        const int prow = (valid && e > start) ? rowidx[e - 1] : -1;
        if (valid && prow > row && unsorted) *unsorted = 1;
        const int tile = row >> rshift;
        const bool head = valid && (e == start || (prow >> rshift) != tile);
        const unsigned long long heads = __ballot(head);
        const unsigned long long le = heads & ((2ull << lane) - 1ull);
        int rank;
        if (le) rank = lane - (63 - __builtin_clzll(le));
        else rank = carry + lane;
        const bool ovf = valid && rank >= S;
        const unsigned long long om = __ballot(ovf);
        const int opos = ovbase + __builtin_popcountll(om & ((1ull << lane) - 1ull));
        fn(valid ? e : -1, row, tile, rank, opos);
        ovbase += __builtin_popcountll(om);
        const int last = (end - e0 - 1) < 63 ? (end - e0 - 1) : 63;
        carry = __shfl(rank, last, 64) + 1;
    }
}

// histogram of segment lengths (bins 1..RT_MAX_HIST-1, longer segments clamp into the last bin) + sortedness flag
static __global__ __launch_bounds__(256) void rhs_tiled_hist_kernel(const int* __restrict__ colptr, const int* __restrict__ rowidx,
                                                             int64_t ncols, int rshift, unsigned long long* __restrict__ hist,
                                                             int* __restrict__ unsorted) {
    __shared__ unsigned int lh[RT_MAX_HIST];
    if (threadIdx.x < RT_MAX_HIST) lh[threadIdx.x] = 0;
    __syncthreads();
    const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j < ncols) {
        const int start = colptr[j], end = colptr[j + 1];
        rt_walk_column(rowidx, start, end, rshift, 0x7fffffff, unsorted, [&](int e, int row, int tile, int rank, int) {
            if (e >= 0) {
                const bool lastofseg = (e + 1 >= end) || ((rowidx[e + 1] >> rshift) != tile);
                if (lastofseg) atomicAdd(&lh[(rank + 1) < RT_MAX_HIST ? (rank + 1) : RT_MAX_HIST - 1], 1u);
            }
        });
    }
    __syncthreads();
    if (threadIdx.x < RT_MAX_HIST && lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)lh[threadIdx.x]);
}

// overflow nonzeros per column for a given S (feeds the exclusive scan that makes the overflow column pointers)
static __global__ __launch_bounds__(256) void rhs_tiled_ovcount_kernel(const int* __restrict__ colptr, const int* __restrict__ rowidx,
                                                                int64_t ncols, int rshift, int S, int* __restrict__ ovcnt) {
    const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= ncols) return;
    const int start = colptr[j], end = colptr[j + 1];
    int total = 0;
    rt_walk_column(rowidx, start, end, rshift, S, nullptr, [&](int e, int, int, int rank, int opos) {
        const bool ovf = e >= 0 && rank >= S;
        const unsigned long long om = __ballot(ovf);
        total += __builtin_popcountll(om);
    });
    if ((threadIdx.x & 63) == 0) ovcnt[j] = total;
}

// scatter the nonzeros into the slot stream / the overflow CSC
template <class T>
__global__ __launch_bounds__(256) void rhs_tiled_fill_kernel(const int* __restrict__ colptr, const int* __restrict__ rowidx,
                                                             const T* __restrict__ vals, RhsTiledGeom G,
                                                             T* __restrict__ svals, uint16_t* __restrict__ soffs,
                                                             const int* __restrict__ ovptr, int* __restrict__ ovrow,
                                                             T* __restrict__ ovval) {
    const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= G.ncols_tiled) return;
    const int start = colptr[j], end = colptr[j + 1];
    const int ob = ovptr ? ovptr[j] : 0;
    const int rmask = (1 << G.rshift) - 1;
    rt_walk_column(rowidx, start, end, G.rshift, G.S, nullptr, [&](int e, int row, int tile, int rank, int opos) {
        if (e < 0) return;
        const T v = vals[e];
        if (rank < G.S) {
            const int64_t s = rt_slot_index(G, j, tile, rank);
            svals[s] = v;
            soffs[s] = (uint16_t)((row & rmask) * G.rowb);
        } else {
            ovrow[ob + opos] = row;
            ovval[ob + opos] = v;
        }
    });
}

// ---------------------------------------------------------------------------
// DPP helpers: value of lane (row*16 + I) for every lane of its 16-lane row (row_newbcast, gfx90a+)
// ---------------------------------------------------------------------------
template <int I> __device__ __forceinline__ int rt_bc(int v) {
    return __builtin_amdgcn_update_dpp(0, v, 0x150 + I, 0xf, 0xf, true);
}
template <int I> __device__ __forceinline__ float rt_bcast_val(float v) { return __int_as_float(rt_bc<I>(__float_as_int(v))); }
template <int I> __device__ __forceinline__ double rt_bcast_val(double v) {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = (unsigned)rt_bc<I>((int)(unsigned)u), hi = (unsigned)rt_bc<I>((int)(unsigned)(u >> 32));
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
template <int B, int E, class Fn> __device__ __forceinline__ void rt_static_for(Fn&& fn) {
    if constexpr (B < E) {
        fn(std::integral_constant<int, B>{});
        rt_static_for<B + 1, E>(fn);
    }
}
__device__ __forceinline__ float rt_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double rt_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <class T> struct RtVec;
template <> struct RtVec<float> { typedef float type __attribute__((ext_vector_type(4))); static constexpr int N = 4; };
template <> struct RtVec<double> { typedef double type __attribute__((ext_vector_type(2))); static constexpr int N = 2; };

// ---------------------------------------------------------------------------
// Vector-memory traffic of the tile loop is issued from inline asm so that hipcc keeps NO vmcnt bookkeeping for it: with
// compiler-visible loads it drains vmcnt(0) in the middle of the compute phase (observed: second batch of every tile),
// which serialises the LDS-DMA of the next tile and the slot prefetch with the FMAs.  The waits are placed by hand
// (rt_wait_vm<N>: loads return in order, so "all but the newest N" = the next tile's F and slots have landed while the
// slots of the tile after it stay in flight).
// ---------------------------------------------------------------------------
// 16 bytes per lane, global (uniform base + per-lane 32-bit offset) -> LDS (uniform address in M0, lane-linear).
// M0 is saved once before a run of these and restored after it (rt_m0_save / rt_m0_restore).
__device__ __forceinline__ void rt_glds16(const char* base_uniform, unsigned lane_off, unsigned lds_addr_uniform) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(lane_off), "s"(base_uniform), "s"(lds_addr_uniform) : "memory");
}
__device__ __forceinline__ unsigned rt_m0_save() {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0" : "=s"(keep));
    return keep;
}
__device__ __forceinline__ void rt_m0_restore(unsigned keep) { asm volatile("s_mov_b32 m0, %0" ::"s"(keep)); }
// 4 bytes per lane, global -> LDS (the slot prefetch: no VGPR is in flight, so there is nothing hipcc could copy or
// spill before the data has landed -- with register destinations it did exactly that once the loop was unrolled)
__device__ __forceinline__ void rt_glds4(const char* base_uniform, unsigned lane_off, unsigned lds_addr_uniform) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1"
                 :: "v"(lane_off), "s"(base_uniform), "s"(lds_addr_uniform) : "memory");
}
// The same two copies for use INSIDE the compute loop: no "memory" clobber (the LDS they fill is not read before the next
// barrier, and a clobber between the batches would pin hipcc's own LDS reads around it), and the partial last piece of a
// slot copy is predicated inside the asm (a compiler-visible `if` between the batches makes hipcc sink the FMAs and spill).
__device__ __forceinline__ void rt_glds16_nc(const char* base_uniform, unsigned lane_off, unsigned lds_addr_uniform) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(lane_off), "s"(base_uniform), "s"(lds_addr_uniform));
}
template <int LIMIT>   // lanes with lane_off >= LIMIT do not take part
__device__ __forceinline__ void rt_glds4_nc(const char* base_uniform, unsigned lane_off, unsigned lds_addr_uniform) {
    if constexpr (LIMIT >= 256) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1"
                     :: "v"(lane_off), "s"(base_uniform), "s"(lds_addr_uniform));
    } else {
        unsigned long long keep;
        asm volatile("s_mov_b32 m0, %3\n\tv_cmp_gt_u32 vcc, %4, %1\n\ts_and_saveexec_b64 %0, vcc\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b64 exec, %0"
                     : "=&s"(keep) : "v"(lane_off), "s"(base_uniform), "s"(lds_addr_uniform), "n"(LIMIT) : "vcc", "scc");
    }
}
#ifdef RCPPML_EXPERIMENTS
// probe build: the same copies with a RUN-TIME lane limit (0 = nobody takes part), so that the ablation switches of
// profiles/ablate_rhs_tiled.sh need no branch between the batches either
__device__ __forceinline__ void rt_glds16_rt(const char* base_uniform, unsigned lane_off, unsigned lds_addr_uniform, unsigned limit) {
    unsigned long long keep;
    asm volatile("s_mov_b32 m0, %3\n\tv_cmp_gt_u32 vcc, %4, %1\n\ts_and_saveexec_b64 %0, vcc\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(base_uniform), "s"(lds_addr_uniform), "s"(limit) : "vcc", "scc");
}
__device__ __forceinline__ void rt_glds4_rt(const char* base_uniform, unsigned lane_off, unsigned lds_addr_uniform, unsigned limit) {
    unsigned long long keep;
    asm volatile("s_mov_b32 m0, %3\n\tv_cmp_gt_u32 vcc, %4, %1\n\ts_and_saveexec_b64 %0, vcc\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b64 exec, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(base_uniform), "s"(lds_addr_uniform), "s"(limit) : "vcc", "scc");
}
#endif
template <int N> __device__ __forceinline__ void rt_wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// s_waitcnt lgkmcnt(N) only
template <int N> __device__ __forceinline__ void rt_wait_lgkm() {
    static_assert(N >= 0 && N < 16, "lgkmcnt is a 4-bit counter");
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

// ---------------------------------------------------------------------------
// The kernel.  NV = 256-byte slices per row of F (row bytes = 256 NV), S slots per (column, tile), NR = rounds per wave
// (compile time: a runtime round count would put branches between the batches, and hipcc then sinks the FMAs of every
// batch below them and spills), NW = waves per workgroup (12 or 16: whole waves per SIMD; register budget 168 / 128).
// grid = P * ncb workgroups of 64*NW threads, 128 KiB of dynamic LDS (one workgroup per CU).
//
// Tile t of a partition: [slots(t) -> working copy] [issue LDS-DMA of tile t+1 into the other buffer: every wave copies
// its own contiguous run of KiB-chunks] [issue the slot loads of tile t+2] [compute tile t: LDS reads of batch b+1 are
// issued before the FMAs of batch b, one lgkmcnt wait per batch] [vmcnt: tile t+1 and slots(t+1) landed] [barrier].
// The loop is issue-bound (rocprofv3: SIMDs > 85 % busy), so everything around the five instructions of a step is kept
// off the per-tile path: no guards or clamps in the copy (a short last tile takes a separate branch), no M0 save per
// chunk, one LDS wait per batch.
// ---------------------------------------------------------------------------
template <class T, int NV, int S, int NR, int NW>
__global__ __launch_bounds__(64 * NW) void rhs_tiled_kernel(const T* __restrict__ svals, const uint16_t* __restrict__ soffs,
                                                            const T* __restrict__ F, RhsTiledGeom G,
                                                            const T* __restrict__ Binit, T* __restrict__ Bout) {
    typedef typename RtVec<T>::type V;
    constexpr int VN = RtVec<T>::N;
    constexpr int NST = NR * S;                 // steps per (wave, tile)
    constexpr int NB = (NST + 15) / 16;         // slot registers (value, offset) per lane and tile
    constexpr int UB = NV == 1 ? 4 : (NV == 2 ? 2 : 1);     // steps per batch of LDS reads; two batches in flight
    constexpr int NBATCH = (NST + UB - 1) / UB;
    constexpr int NCH = RT_SLAB_BYTES / 1024;   // KiB-chunks per tile
    constexpr int CBASE = NCH / NW, CEXTRA = NCH % NW;
    extern __shared__ char rt_slab[];           // 2 x 64 KiB of F + the two-stage slot ring
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, u = lane & 15;
    const int p = blockIdx.x % G.P;
    const int64_t cb = blockIdx.x / G.P;
    const int t0 = (int)((int64_t)G.ntiles * p / G.P), t1 = (int)((int64_t)G.ntiles * (p + 1) / G.P);
    const int64_t col0 = (cb * NW + w) * (int64_t)(4 * NR);       // first column of this wave
    const int k = G.rowb / (int)sizeof(T);
    const int64_t fbytes = G.nrows * (int64_t)G.rowb;
    const char* Fb = reinterpret_cast<const char*>(F);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)rt_slab;

    V acc[NR][NV];
    if (Binit != nullptr) {
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            int64_t j = col0 + 4 * q + g;
            j = j < G.ncols_tiled ? j : G.ncols_tiled - 1;          // columns past the end are never stored
#pragma unroll
            for (int v = 0; v < NV; ++v) acc[q][v] = *reinterpret_cast<const V*>(Binit + j * k + (64 * v + 4 * u) * 4 / (int)sizeof(T));
        }
    } else {
#pragma unroll
        for (int q = 0; q < NR; ++q)
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int e = 0; e < VN; ++e) acc[q][v][e] = T(0);
    }
    // Make hipcc settle its OWN vmcnt bookkeeping here: a use of every accumulator forces its wait for the Binit loads now.
    // Left pending, that bookkeeping follows the accumulators into the tile loop as `s_waitcnt vmcnt(7..4)` before the first
    // FMA of every batch -- which, with the asm loads below in flight (invisible to hipcc), waits for the next tile's LDS-DMA
    // in the middle of the compute phase.
#pragma unroll
    for (int q = 0; q < NR; ++q)
#pragma unroll
        for (int v = 0; v < NV; ++v) asm volatile("" : "+v"(acc[q][v]));
    rt_wait_vm<0>();

    // this wave's run of KiB-chunks of every tile: CBASE of them, one more for the first CEXTRA waves
    const int cstart = w * CBASE + (w < CEXTRA ? w : CEXTRA);
    const unsigned choff = (unsigned)cstart * 1024u + (unsigned)lane * 16u;
    const unsigned m0keep = rt_m0_save();
    auto slab_load = [&](int tile, int buf) {
        const int64_t base = (int64_t)tile * RT_SLAB_BYTES;
        const char* src0 = Fb + base;
        const unsigned ldst = lds0 + buf * RT_SLAB_BYTES + cstart * 1024;
        if (fbytes - base >= RT_SLAB_BYTES) {
#pragma unroll
            for (int i = 0; i < CBASE; ++i) rt_glds16(src0, choff + i * 1024, ldst + i * 1024);
            if (CEXTRA > 0 && w < CEXTRA) rt_glds16(src0, choff + CBASE * 1024, ldst + CBASE * 1024);
        } else {                                                            // short last tile: stay inside F
            const unsigned lim = (unsigned)(fbytes - base) - 16u;
#pragma unroll
            for (int i = 0; i < CBASE + (CEXTRA > 0 ? 1 : 0); ++i) {
                if (i < CBASE || w < CEXTRA) {
                    unsigned o = choff + i * 1024;
                    o = o < lim ? o : lim;
                    rt_glds16(src0, o, ldst + i * 1024);
                }
            }
        }
    };
    // slot stream: [column block][tile][wave][step][lane group]; one tile further = NW * NST * 4 slots further.
    // Prefetched by LDS-DMA into a two-stage ring behind the two F tiles (each wave owns VB + OB bytes per stage).
    constexpr int64_t tstride = (int64_t)NW * NST * 4;
    constexpr int VB = NST * 4 * (int)sizeof(T), OB = NST * 4 * 2;             // bytes of values / offsets per (wave, tile)
    constexpr int NCV = (VB + 255) / 256, NCO = (OB + 255) / 256;             // dword copies per (wave, tile)
    constexpr int STAGE = NW * (VB + OB);
    const int64_t slot0 = ((cb * G.ntiles + t0) * NW + w) * (int64_t)(NST * 4);
    const char* svp = reinterpret_cast<const char*>(svals + slot0);
    const char* sop = reinterpret_cast<const char*>(soffs + slot0);
    const unsigned sl0 = lds0 + 2 * RT_SLAB_BYTES + w * (VB + OB);             // this wave's slot area of stage 0
    char* const slp = rt_slab + 2 * RT_SLAB_BYTES + w * (VB + OB);
    auto slots_issue = [&](int stage, int ord /* tile - t0 */) {
        const unsigned dst = sl0 + stage * STAGE;
        const char* sv = svp + (int64_t)ord * tstride * (int64_t)sizeof(T);
        const char* so = sop + (int64_t)ord * tstride * 2;
#pragma unroll
        for (int i = 0; i < NCV; ++i)
            if (256 * i + 4 * lane < VB) rt_glds4(sv + 256 * i, 4u * lane, dst + 256 * i);
#pragma unroll
        for (int i = 0; i < NCO; ++i)          // offsets travel as dwords too (two per lane): sub-dword LDS-DMA pads every lane to a dword
            if (256 * i + 4 * lane < OB) rt_glds4(so + 256 * i, 4u * lane, dst + VB + 256 * i);
    };
    // The copies of the NEXT tiles are not issued in one burst at the top of a tile (a wave stalls ~100 cycles per LDS-DMA
    // instruction while the CU's address path is busy with the other waves' bursts): they are spread over the batches of
    // the compute loop, one piece at a time -- F pieces first, the slot pieces of tile t+2 last, so that "all but the
    // newest NCV + NCO" at the end of the tile still means "tile t+1 and slots(t+1) have landed".  No guards: past the end
    // of the partition the same last tile / slots are fetched again (harmless), offsets are clamped into F.
    constexpr int NPF = CBASE + (CEXTRA > 0 ? 1 : 0);
    constexpr int NPT = NPF + NCV + NCO;
    const char* nx_src0 = Fb; unsigned nx_ldst = 0, nx_lim = 0, nx_sdst = 0;
    const char* nx_sv = svp; const char* nx_so = sop;
#ifdef RCPPML_EXPERIMENTS
    const bool nx_slab = !(G.dbg & 2), nx_slots = !(G.dbg & 4);
#else
    [[maybe_unused]] constexpr bool nx_slab = true, nx_slots = true;
#endif
    // piece CBASE exists for the first CEXTRA waves only; the others copy their own last piece once more (same bytes to the
    // same place) rather than branch
    const unsigned xpiece = (unsigned)(w < CEXTRA ? CBASE : CBASE - 1) * 1024u;
    auto issue_piece = [&](auto PI) {
        constexpr int pi = decltype(PI)::value;
        if constexpr (pi < NPF) {
            const unsigned po = pi < CBASE ? (unsigned)pi * 1024u : xpiece;
            unsigned o = choff + po;
            o = o < nx_lim ? o : nx_lim;
#ifdef RCPPML_EXPERIMENTS
            rt_glds16_rt(nx_src0, o, nx_ldst + po, nx_slab ? 0xffffffffu : 0u);     // ablation switch without a branch
#else
            rt_glds16_nc(nx_src0, o, nx_ldst + po);
#endif
        } else if constexpr (pi < NPF + NCV) {
            constexpr int i = pi - NPF;
#ifdef RCPPML_EXPERIMENTS
            rt_glds4_rt(nx_sv + 256 * i, 4u * lane, nx_sdst + 256 * i, nx_slots ? (unsigned)(VB - 256 * i) : 0u);
#else
            rt_glds4_nc<VB - 256 * i>(nx_sv + 256 * i, 4u * lane, nx_sdst + 256 * i);
#endif
        } else {
            constexpr int i = pi - NPF - NCV;
#ifdef RCPPML_EXPERIMENTS
            rt_glds4_rt(nx_so + 256 * i, 4u * lane, nx_sdst + VB + 256 * i, nx_slots ? (unsigned)(OB - 256 * i) : 0u);
#else
            rt_glds4_nc<OB - 256 * i>(nx_so + 256 * i, 4u * lane, nx_sdst + VB + 256 * i);
#endif
        }
    };
    // the wave's slots of one stage -> registers (lane 16 g' + i' of block b holds slot (16 b + i') * 4 + g' ... i.e. the
    // lanes read the stream in order: lane l of block b = slot 64 b + l, which is step 16 b + l / 4, lane group l % 4;
    // the kernel wants step 16 b + u for group g in lane 16 g + u)
    auto slots_read = [&](int stage, T (&cv)[NB], unsigned (&co)[NB]) {
        const char* src = slp + stage * STAGE;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int step = 16 * b + u;
            const int sidx = (step < NST ? step : 0) * 4 + g;             // lanes past the last step re-read step 0 (never consumed)
            cv[b] = *reinterpret_cast<const T*>(src + sidx * (int)sizeof(T));
            co[b] = *reinterpret_cast<const uint16_t*>(src + VB + sidx * 2);
        }
    };
    auto compute = [&](const T (&cv)[NB], const unsigned (&co)[NB], int buf) {
        const int lbase = buf * RT_SLAB_BYTES + u * 16;
        V f[2][UB][NV];
        auto reads = [&](auto QB) {
            constexpr int b = decltype(QB)::value;
            rt_static_for<0, UB>([&](auto I) {
                constexpr int i = decltype(I)::value;
                constexpr int step = b * UB + i;
                if constexpr (step < NST) {
                    const int a = lbase + rt_bc<(step & 15)>((int)co[step >> 4]);
#pragma unroll
                    for (int v = 0; v < NV; ++v) f[b & 1][i][v] = *reinterpret_cast<const V*>(rt_slab + a + 256 * v);
                }
            });
        };
        reads(std::integral_constant<int, 0>{});
        rt_static_for<0, NBATCH>([&](auto QB) {
            constexpr int b = decltype(QB)::value;
            constexpr int nnext = (b + 1 < NBATCH) ? ((NST - (b + 1) * UB) < UB ? (NST - (b + 1) * UB) : UB) * NV : 0;
            if constexpr (b + 1 < NBATCH) reads(std::integral_constant<int, b + 1>{});
            __builtin_amdgcn_sched_barrier(0);
            rt_wait_lgkm<nnext>();                  // LDS returns in order: everything but the reads just issued is back
            rt_static_for<0, UB>([&](auto I) {
                constexpr int i = decltype(I)::value;
                constexpr int step = b * UB + i;
                if constexpr (step < NST) {
                    const T val = rt_bcast_val<(step & 15)>(cv[step >> 4]);
#pragma unroll
                    for (int v = 0; v < NV; ++v)
#pragma unroll
                        for (int e = 0; e < VN; ++e) acc[step / S][v][e] = rt_fma(val, f[b & 1][i][v][e], acc[step / S][v][e]);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
            rt_static_for<0, NPT>([&](auto PI) {
                if constexpr (decltype(PI)::value * NBATCH / NPT == b) issue_piece(PI);
            });
        });
        // every FMA of this tile is done HERE: without the tie hipcc rotates the tail of the compute phase below the
        // barrier into the next tile, and the extra live ranges spill
#pragma unroll
        for (int q = 0; q < NR; ++q)
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("" : "+v"(acc[q][v]));
    };

    if (t0 < t1) {
        slab_load(t0, 0);
        slots_issue(0, 0);
        slots_issue(1, t0 + 1 < t1 ? 1 : 0);
    }
    rt_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    T cv[NB];
    unsigned co[NB];
    if (t0 < t1) slots_read(0, cv, co);
    for (int t = t0; t < t1; ++t) {
        const int buf = (t - t0) & 1;
        // slots(t) were requested from the ring at the end of the previous tile (this wave's own LDS-DMA data, complete by
        // its vmcnt wait), so their LDS latency ran under the barrier; they must be in registers before their stage is refilled
#pragma unroll
        for (int b = 0; b < NB; ++b) asm volatile("" : "+v"(cv[b]), "+v"(co[b]));
        rt_wait_lgkm<0>();
        {   // what the compute loop copies meanwhile: tile t+1 -> the other buffer, slots(t+2) -> the stage just read
            const int tn = t + 1 < t1 ? t + 1 : t1 - 1, ts = t + 2 < t1 ? t + 2 : t1 - 1;
            const int64_t base = (int64_t)tn * RT_SLAB_BYTES;
            const int64_t left = fbytes - base;
            nx_src0 = Fb + base;
            nx_ldst = lds0 + (buf ^ 1) * RT_SLAB_BYTES + cstart * 1024;
            nx_lim = (unsigned)(left < RT_SLAB_BYTES ? left : RT_SLAB_BYTES) - 16u;
            nx_sdst = sl0 + buf * STAGE;
            nx_sv = svp + (int64_t)(ts - t0) * tstride * (int64_t)sizeof(T);
            nx_so = sop + (int64_t)(ts - t0) * tstride * 2;
        }
#ifdef RCPPML_EXPERIMENTS
        if (!(G.dbg & 1)) compute(cv, co, buf);
        else rt_static_for<0, NPT>([&](auto PI) { issue_piece(PI); });
#else
        compute(cv, co, buf);
#endif
        rt_wait_vm<NCV + NCO>();                   // all but the slot copies just issued: tile t+1 and slots(t+1) have landed
        slots_read(buf ^ 1, cv, co);               // slots(t+1) (past the last tile: a harmless re-read)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    rt_wait_vm<0>();                               // nothing may still be landing in LDS when the workgroup retires
    rt_m0_restore(m0keep);

#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const int64_t j = col0 + 4 * q + g;
        if (j < G.ncols_tiled) {
            T* dst = Bout + (int64_t)p * G.ncols_tiled * k + j * k;
#pragma unroll
            for (int v = 0; v < NV; ++v) *reinterpret_cast<V*>(dst + (64 * v + 4 * u) * 4 / (int)sizeof(T)) = acc[q][v];
        }
    }
}

// ---------------------------------------------------------------------------
// The spilled nonzeros (overflow CSC: about a dozen per column at C2).  One 16-lane group per column -- a whole row of
// F per gather, no cross-lane reduction -- and U gathers in flight per lane; writes every column of B (zeros where
// nothing spilled), which then seeds the accumulators of rhs_tiled_kernel.
// ---------------------------------------------------------------------------
template <class T, int NV, int U>
__global__ __launch_bounds__(256) void rhs_tiled_spill_kernel(const int* __restrict__ ovptr, const int* __restrict__ ovrow,
                                                              const T* __restrict__ ovval, int64_t ncols,
                                                              const T* __restrict__ F, int k, T* __restrict__ B) {
    typedef typename RtVec<T>::type V;
    constexpr int VN = RtVec<T>::N;
    const int64_t j = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (j >= ncols) return;
    const int u = threadIdx.x & 15;
    const int start = ovptr[j], end = ovptr[j + 1];
    V acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int e = 0; e < VN; ++e) acc[v][e] = T(0);
    const T* Fl = F + (4 * u) * 4 / (int)sizeof(T);
    for (int i = start; i < end; i += U) {
        int r[U];
        T a[U];
#pragma unroll
        for (int x = 0; x < U; ++x) {
            const int ii = i + x < end ? i + x : end - 1;
            r[x] = ovrow[ii];
            const T av = ovval[ii];
            a[x] = i + x < end ? av : T(0);
        }
        V f[U][NV];
#pragma unroll
        for (int x = 0; x < U; ++x)
#pragma unroll
            for (int v = 0; v < NV; ++v) f[x][v] = *reinterpret_cast<const V*>(Fl + (int64_t)r[x] * k + 64 * v * 4 / (int)sizeof(T));
#pragma unroll
        for (int x = 0; x < U; ++x)
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int e = 0; e < VN; ++e) acc[v][e] = rt_fma(a[x], f[x][v][e], acc[v][e]);
    }
    T* dst = B + j * k + (4 * u) * 4 / (int)sizeof(T);
#pragma unroll
    for (int v = 0; v < NV; ++v) *reinterpret_cast<V*>(dst + 64 * v * 4 / (int)sizeof(T)) = acc[v];
}

// ---------------------------------------------------------------------------
// The tail columns (those beyond the last whole workgroup of the tiled kernel: 1 696 of C2's 100 000): a whole 256-thread
// workgroup per column, sixteen 16-lane groups each gathering every sixteenth nonzero's row of F (a whole row per gather,
// U in flight), partial rows summed through LDS in group order (fixed -> deterministic).  One wave per column, as the generic
// gather kernel runs them, makes these few columns a 16 us latency chain of ~200 dependent gathers.
// ---------------------------------------------------------------------------
template <class T, int NV, int U>
__global__ __launch_bounds__(256) void rhs_tail_kernel(const int* __restrict__ colptr, const int* __restrict__ rowidx,
                                                       const T* __restrict__ vals, int64_t col0, int64_t ncols,
                                                       const T* __restrict__ F, int k, T* __restrict__ B) {
    typedef typename RtVec<T>::type V;
    constexpr int VN = RtVec<T>::N;
    __shared__ V part[16][16 * NV];
    const int64_t j = col0 + blockIdx.x;
    if (j >= ncols) return;
    const int g = threadIdx.x >> 4, u = threadIdx.x & 15;
    const int start = colptr[j], end = colptr[j + 1];
    V acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int e = 0; e < VN; ++e) acc[v][e] = T(0);
    const T* Fl = F + (4 * u) * 4 / (int)sizeof(T);
    for (int i = start + g; i < end; i += 16 * U) {
        int r[U];
        T a[U];
#pragma unroll
        for (int x = 0; x < U; ++x) {
            const int ii = i + 16 * x;
            const bool ok = ii < end;
            r[x] = rowidx[ok ? ii : end - 1];
            a[x] = ok ? vals[ii] : T(0);
        }
        V f[U][NV];
#pragma unroll
        for (int x = 0; x < U; ++x)
#pragma unroll
            for (int v = 0; v < NV; ++v) f[x][v] = *reinterpret_cast<const V*>(Fl + (int64_t)r[x] * k + 64 * v * 4 / (int)sizeof(T));
#pragma unroll
        for (int x = 0; x < U; ++x)
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int e = 0; e < VN; ++e) acc[v][e] = rt_fma(a[x], f[x][v][e], acc[v][e]);
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) part[g][16 * v + u] = acc[v];
    __syncthreads();
    if (g == 0) {
        T* dst = B + j * k + (4 * u) * 4 / (int)sizeof(T);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            V s = part[0][16 * v + u];
#pragma unroll
            for (int q = 1; q < 16; ++q) s += part[q][16 * v + u];
            *reinterpret_cast<V*>(dst + 64 * v * 4 / (int)sizeof(T)) = s;
        }
    }
}

// B (+)= sum_p Bp[p], partition order (deterministic)
template <class T>
__global__ __launch_bounds__(256) void rhs_tiled_reduce_kernel(const T* __restrict__ Bp, int P, int64_t n4, int accumulate,
                                                               T* __restrict__ B) {
    typedef typename RtVec<T>::type V;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    V s;
    if (accumulate) s = reinterpret_cast<const V*>(B)[i];
    else
#pragma unroll
        for (int e = 0; e < RtVec<T>::N; ++e) s[e] = T(0);
    for (int p = 0; p < P; ++p) s += reinterpret_cast<const V*>(Bp)[p * n4 + i];
    reinterpret_cast<V*>(B)[i] = s;
}

}  // namespace rk
