// rhs_win_finish.hip.h -- run_plan of a window-form right-hand side (kernels_rhs_win.hip.h): the tile kernel, then the pass that sums the
// partial slabs of the row partitions (partition order) and adds the overflow nonzeros (row order): rhs_win_finish_kernel.  Included by
// ONE translation unit (ops_rhs_win.hip); the tile-loop kernels live in ops_rhs_win_{f32,f64}_nv*.hip.
// Reference semantics: primitives/cpu/rhs.hpp:52-70, fused_nnls.hpp:109-114.
//
// Round 5, measured and not kept (rocprofv3 on tools/rhs_tiled_bench.py, C2 fp32, profiles/r05_rhs_finish_probes.txt):
// the finishing pass takes 22-25 us per call.  (1) A form with the partition count as a template parameter, two columns per lane group
// and every independent load issued at once (P x 2 slab pieces + the overflow pointers, then the entries, then the row gathers: three
// round trips instead of 5-11): 25 / 31 us (H / W) -- not faster, the pass is not bound by its dependent chain.  (2) The pass timed
// alone, back to back, with no tile kernel in front of it: the same -- it does not wait for the tile kernel's dirty L2 lines.
// (3) With almost nothing to gather (slot rate 2.0, 0.6 % overflow): 19.5 us on BOTH sides -- 80 % of the pass is the slab sum
// (H: 51 MB read + 26 MB written = 3.9 TB/s, i.e. the slabs come back from HBM, not from the Infinity Cache).  (4) A slab stride that is
// an odd multiple of 4 KiB on the W side (5 MiB apart, the P pieces of a column could alias onto one channel): no change.
#pragma once
#include "rhs_win_impl.hip.h"

namespace rk {

// ---------------------------------------------------------------------------
// B(:,j) = sum_p Bp[p](:,j) (partition order) + the overflow nonzeros of column j (row order).  One 16-lane group per
// column: a whole row of F per gather, U gathers in flight.  With P == 1 the tiled kernel has written B itself and this
// kernel only adds the overflow (accumulate = 1); it is not launched at all when P == 1 and nothing overflowed.
// ---------------------------------------------------------------------------
// Round 5: the partition loop is unrolled by four (four slab pieces in flight; the additions keep their partition order), the two
// overflow pointers are loaded before the slab pieces are consumed and 256-byte rows gather eight overflow rows at a time:
// 25.3 -> 23.7 us per call at C2 (the pass is bound by its slab traffic, not by these chains: header comment above).
template <class T, int NV, int U>
__global__ __launch_bounds__(256) void rhs_win_finish_kernel(const T* __restrict__ Bp, int P, int64_t ncp, int accumulate,
                                                               const int* __restrict__ ovptr, const int* __restrict__ ovrow,
                                                               const T* __restrict__ ovval, int64_t ncols,
                                                               const T* __restrict__ F, int k, T* __restrict__ B) {
    typedef typename RtVec<T>::type V;
    constexpr int VN = RtVec<T>::N;
    constexpr int SL = 64 * 4 / (int)sizeof(T);
    const int64_t j = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (j >= ncols) return;
    const int u = threadIdx.x & 15;
    const int lo = (4 * u) * 4 / (int)sizeof(T);
    int start = 0, end = 0;
    if (ovptr) { start = ovptr[j * P]; end = ovptr[(j + 1) * P]; }       // one pointer per (column, partition)
    V acc[NV];
    const T* src = accumulate ? B : Bp;
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = *reinterpret_cast<const V*>(src + j * k + lo + SL * v);
    if (!accumulate) {
        int p = 1;
        for (; p + 3 < P; p += 4) {
            V t[4][NV];
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int v = 0; v < NV; ++v) t[x][v] = *reinterpret_cast<const V*>(Bp + ((int64_t)(p + x) * ncp + j) * k + lo + SL * v);
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int v = 0; v < NV; ++v) acc[v] += t[x][v];
        }
        for (; p < P; ++p)
#pragma unroll
            for (int v = 0; v < NV; ++v) acc[v] += *reinterpret_cast<const V*>(Bp + ((int64_t)p * ncp + j) * k + lo + SL * v);
    }
    const T* Fl = F + lo;
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
            for (int v = 0; v < NV; ++v) f[x][v] = *reinterpret_cast<const V*>(Fl + (int64_t)r[x] * k + SL * v);
#pragma unroll
        for (int x = 0; x < U; ++x)
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int e = 0; e < VN; ++e) acc[v][e] = rt_fma(a[x], f[x][v][e], acc[v][e]);
    }
    T* dst = B + j * k + lo;
#pragma unroll
    for (int v = 0; v < NV; ++v) *reinterpret_cast<V*>(dst + SL * v) = acc[v];
}

}  // namespace rk

namespace rw_launch {

template <class T>
void run_plan(rcppml_hip_ctx* c, const rcppml_rhs_plan* pl, const T* F, T* B) {
    using namespace rk;
    const RhsWinGeom& G = pl->WG;
    const int NV = G.rowb / 256;
    T* Bout = G.P > 1 ? (T*)pl->Bp : B;
    // (RCPPML_RW_DBG, -DRCPPML_EXPERIMENTS builds only: 8 = no tile-kernel launch, 16 = no finishing launch -- timing probes)
    if (G.dbg & 8) { /* skip */ } else
    if constexpr (std::is_same<T, float>::value) {
        if (NV == 1) rcppml_rw_launch_f32_nv1(c, pl, F, Bout);
        else if (NV == 2) rcppml_rw_launch_f32_nv2(c, pl, F, Bout);
        else throw std::runtime_error("rhs_planned: unsupported row size");
    } else {
        if (NV == 1) rcppml_rw_launch_f64_nv1(c, pl, F, Bout);
        else if (NV == 2) rcppml_rw_launch_f64_nv2(c, pl, F, Bout);
        else if (NV == 4) rcppml_rw_launch_f64_nv4(c, pl, F, Bout);
        else throw std::runtime_error("rhs_planned: unsupported row size");
    }
    if ((G.P > 1 || pl->ovnnz > 0) && !(G.dbg & 16)) {
        const unsigned grid = (unsigned)((G.ncols + 15) / 16);
        const int64_t ncp = (int64_t)G.ncb * (4 * G.nr * G.NW);
        const int* ovp = pl->ovnnz > 0 ? pl->ovptr : nullptr;
        const int acc = G.P > 1 ? 0 : 1;
        if (NV == 1)
            hipLaunchKernelGGL((rhs_win_finish_kernel<T, 1, 8>), dim3(grid), dim3(256), 0, c->stream, (const T*)pl->Bp, G.P, ncp, acc, ovp,
                               (const int*)pl->ovrow, (const T*)pl->ovval, G.ncols, F, pl->k, B);
        else if (NV == 2)
            hipLaunchKernelGGL((rhs_win_finish_kernel<T, 2, 4>), dim3(grid), dim3(256), 0, c->stream, (const T*)pl->Bp, G.P, ncp, acc, ovp,
                               (const int*)pl->ovrow, (const T*)pl->ovval, G.ncols, F, pl->k, B);
        else
            hipLaunchKernelGGL((rhs_win_finish_kernel<T, 4, 2>), dim3(grid), dim3(256), 0, c->stream, (const T*)pl->Bp, G.P, ncp, acc, ovp,
                               (const int*)pl->ovrow, (const T*)pl->ovval, G.ncols, F, pl->k, B);
        HIPCHK(hipGetLastError());
    }
}

}  // namespace rw_launch
