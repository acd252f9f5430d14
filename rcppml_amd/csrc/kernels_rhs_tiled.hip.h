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
    int64_t nrows;        // rows of the sparse matrix = rows (k-vectors) of F
    int rowb;             // bytes per row of F (k * sizeof(T)), 256 or 512 ... multiple of 256
    int rshift;           // log2(rows per tile), rows per tile = 65536 / rowb
    int ntiles;           // ceil(nrows / R)
    int P;                // row partitions (tiles [p*ntiles/P, (p+1)*ntiles/P))
    int NW;               // waves per workgroup
    int nr;               // rounds per wave (4 columns each)
    int S;                // slots per (column, tile)
    int ncb;              // column blocks = workgroups per partition
    int dbg;              // experiments: 1 = no compute, 2 = no LDS-DMA, 4 = no slot loads
};

__device__ __forceinline__ int64_t rt_slot_index(const RhsTiledGeom& G, int64_t j, int tile, int rank) {
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
__global__ __launch_bounds__(256) void rhs_tiled_hist_kernel(const int* __restrict__ colptr, const int* __restrict__ rowidx,
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
__global__ __launch_bounds__(256) void rhs_tiled_ovcount_kernel(const int* __restrict__ colptr, const int* __restrict__ rowidx,
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
    if (j >= G.ncols) return;
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

__device__ __forceinline__ void rt_glds16(const char* gsrc, char* lds_dst_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst_uniform, 16, 0, 0);
}

// ---------------------------------------------------------------------------
// The kernel.  NV = 256-byte slices per row of F (row bytes = 256 NV), S slots per (column, tile), NR = rounds per wave
// (compile time: a runtime round count would put branches between the batches, and hipcc then sinks the FMAs of every
// batch below them and spills), UB = steps per batch of LDS reads (UB reads in flight per wave and slice).
// grid = P * ncb workgroups of 64*NW threads, 128 KiB of dynamic LDS (one workgroup per CU).
// ---------------------------------------------------------------------------
template <class T, int NV, int S, int NR, int UB>
__global__ __launch_bounds__(1024) void rhs_tiled_kernel(const T* __restrict__ svals, const uint16_t* __restrict__ soffs,
                                                         const T* __restrict__ F, RhsTiledGeom G,
                                                         const T* __restrict__ Binit, T* __restrict__ Bout) {
    typedef typename RtVec<T>::type V;
    constexpr int VN = RtVec<T>::N;
    constexpr int NST = NR * S;                 // steps per (wave, tile)
    constexpr int NB = (NST + 15) / 16;         // coalesced slot loads per (wave, tile)
    static_assert(NST % UB == 0, "whole batches");
    extern __shared__ char rt_slab[];           // 2 x 64 KiB
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, u = lane & 15;
    const int p = blockIdx.x % G.P;
    const int64_t cb = blockIdx.x / G.P;
    const int t0 = (int)((int64_t)G.ntiles * p / G.P), t1 = (int)((int64_t)G.ntiles * (p + 1) / G.P);
    const int64_t col0 = (cb * G.NW + w) * (int64_t)(4 * NR);       // first column of this wave
    const int k = G.rowb / (int)sizeof(T);
    const int64_t fbytes = G.nrows * (int64_t)G.rowb;
    const char* Fb = reinterpret_cast<const char*>(F);

    V acc[NR][NV];
    if (Binit != nullptr) {
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            int64_t j = col0 + 4 * q + g;
            j = j < G.ncols ? j : G.ncols - 1;          // columns past the end are never stored
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

    const int lane16 = lane * 16;
    auto slab_load = [&](int tile, int buf) {
        const int64_t base = (int64_t)tile * RT_SLAB_BYTES;
        const int64_t left = fbytes - base;                                 // the last tile may be short: stay inside F
        const int lim = (int)(left < RT_SLAB_BYTES ? left : RT_SLAB_BYTES) - 16;
        const char* src0 = Fb + base;
#pragma unroll
        for (int i = 0; i < 8; ++i) {                                       // NW >= 8: at most 8 KiB-chunks per wave
            const int c = w + i * G.NW;
            if (c < RT_SLAB_BYTES / 1024) {
                int o = c * 1024 + lane16;
                o = o < lim ? o : lim;
                rt_glds16(src0 + o, rt_slab + buf * RT_SLAB_BYTES + c * 1024);
            }
        }
    };
    // slot stream: [column block][tile][wave][step][lane group]; one tile further = NW * NST * 4 slots further
    const int64_t tstride = (int64_t)G.NW * NST * 4;
    const T* svp = svals + ((cb * G.ntiles + t0) * G.NW + w) * (int64_t)(NST * 4) + g;
    const uint16_t* sop = soffs + ((cb * G.ntiles + t0) * G.NW + w) * (int64_t)(NST * 4) + g;
    int loff[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int step = 16 * b + u;
        loff[b] = (step < NST ? step : 0) * 4;      // lanes past the last step re-read step 0; their rounds do not exist
    }
    T sv[NB];
    int so[NB];
    auto slots_load = [&]() {
        // the values are NOT touched here -- any use would make hipcc drain vmcnt (and with it the LDS-DMA) on the spot
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            sv[b] = svp[loff[b]];
            so[b] = sop[loff[b]];
        }
        svp += tstride;
        sop += tstride;
    };

    if (t0 < t1) {
        slab_load(t0, 0);
        slots_load();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = t0; t < t1; ++t) {
        const int buf = (t - t0) & 1;
        T cv[NB];
        int co[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) { cv[b] = sv[b]; co[b] = so[b]; }
        if (t + 1 < t1) {
            if (!(G.dbg & 2)) slab_load(t + 1, buf ^ 1);
            if (!(G.dbg & 4)) slots_load();
        }
        const int lbase = buf * RT_SLAB_BYTES + u * 16;
        if (!(G.dbg & 1))
        rt_static_for<0, NST / UB>([&](auto QB) {
            constexpr int s0 = decltype(QB)::value * UB;
            V f[UB][NV];
            rt_static_for<0, UB>([&](auto I) {
                constexpr int i = decltype(I)::value;
                constexpr int step = s0 + i;
                const int a = lbase + rt_bc<(step & 15)>(co[step >> 4]);
#pragma unroll
                for (int v = 0; v < NV; ++v) f[i][v] = *reinterpret_cast<const V*>(rt_slab + a + 256 * v);
            });
            __builtin_amdgcn_sched_barrier(0);
            rt_static_for<0, UB>([&](auto I) {
                constexpr int i = decltype(I)::value;
                constexpr int step = s0 + i;
                const T val = rt_bcast_val<(step & 15)>(cv[step >> 4]);
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int e = 0; e < VN; ++e) acc[step / S][v][e] = rt_fma(val, f[i][v][e], acc[step / S][v][e]);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next tile (LDS-DMA) and the next slots have landed
        __syncthreads();
    }

#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const int64_t j = col0 + 4 * q + g;
        if (j < G.ncols) {
            T* dst = Bout + (int64_t)p * G.ncols * k + j * k;
#pragma unroll
            for (int v = 0; v < NV; ++v) *reinterpret_cast<V*>(dst + (64 * v + 4 * u) * 4 / (int)sizeof(T)) = acc[q][v];
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
