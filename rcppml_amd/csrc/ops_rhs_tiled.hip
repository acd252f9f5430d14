// ops_rhs_tiled.hip -- planned (LDS row-tiled) form of the sparse right-hand side: plan = the tile-partitioned slot copy
// of one CSC matrix for one rank and precision, built once per fit on the device; rcppml_hip_rhs_planned runs it.
// Reference semantics: primitives/cpu/rhs.hpp:52-70, fused_nnls.hpp:109-114 (same numbers as rcppml_hip_rhs up to
// summation order).
#include <hipcub/hipcub.hpp>
#include <memory>
#include <mutex>
#include "common.hip.h"
#include "kernels_rhs_tiled.hip.h"

using namespace rk;

struct rcppml_rhs_plan {
    int dtype = 0, k = 0, device = 0;
    RhsTiledGeom G{};
    void* svals = nullptr;
    uint16_t* soffs = nullptr;
    int* ovptr = nullptr;
    int* ovrow = nullptr;
    void* ovval = nullptr;
    void* Bp = nullptr;          // P > 1: per-partition partial outputs
    int64_t ovnnz = 0, nnz = 0, nslots = 0;
    double ov_fraction = 0.0, fill = 0.0;
    ~rcppml_rhs_plan() {
        for (void* p : {svals, (void*)soffs, (void*)ovptr, (void*)ovrow, ovval, Bp})
            if (p) (void)hipFree(p);
    }
};

namespace {

struct Tmp {
    void* p = nullptr;
    explicit Tmp(size_t bytes) { HIPCHK(hipMalloc(&p, bytes < 16 ? 16 : bytes)); }
    ~Tmp() { if (p) (void)hipFree(p); }
};

constexpr int RT_DYN_LDS = 2 * RT_SLAB_BYTES;

template <class K>
void set_lds_once(K kernel, int device) {
    // per-device attribute, set to the one size this kernel ever asks for; serialised (concurrent fits from host threads)
    static std::mutex mu;
    static bool done[64] = {};
    std::lock_guard<std::mutex> lk(mu);
    if (!done[device & 63]) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RT_DYN_LDS));
        done[device & 63] = true;
    }
}

// compiled (S, NR) shapes.  NV = 1 (256-byte rows): S in {2,3,4,5} x NR in {4,6,8,10,12}, S in {6,8} x NR in {4,6,8};
// NV = 2 (512-byte rows): S in {2,3,4,5,6,8} x NR in {2,4,6}.  UB = steps per batch of LDS reads: 8 reads of 16 bytes in flight per lane where the step count divides.
constexpr int rt_ub(int S, int NR, int NV) {
    int best = 1;
    for (int d = 1; d <= NR * S; ++d)
        if ((NR * S) % d == 0 && d * NV <= (NV == 1 ? (S == 5 && NR == 12 ? 5 : 8) : 6)) best = d;
    return best;
}
inline bool rt_shape_ok(int NV, int S, int NR) {
    const bool s_ok = S == 2 || S == 3 || S == 4 || S == 5 || S == 6 || S == 8;
    if (!s_ok) return false;
    if (NV == 1) return (S <= 5) ? (NR == 4 || NR == 6 || NR == 8 || NR == 10 || NR == 12) : (NR == 4 || NR == 6 || NR == 8);
    if (NV == 2) return NR == 2 || NR == 4 || NR == 6;
    return false;
}

template <class T, int NV, int S, int NR>
void launch_tiled(rcppml_hip_ctx* c, const rcppml_rhs_plan* pl, const T* F, const T* Binit, T* Bout) {
    constexpr int UB = rt_ub(S, NR, NV);
    auto kern = rhs_tiled_kernel<T, NV, S, NR, UB>;
    set_lds_once(kern, c->device);
    const RhsTiledGeom& G = pl->G;
    hipLaunchKernelGGL(kern, dim3((unsigned)(G.P * G.ncb)), dim3(64 * G.NW), RT_DYN_LDS, c->stream,
                       (const T*)pl->svals, (const uint16_t*)pl->soffs, F, G, Binit, Bout);
    HIPCHK(hipGetLastError());
}
template <class T, int NV, int S>
void launch_tiled_nr(rcppml_hip_ctx* c, const rcppml_rhs_plan* pl, const T* F, const T* Binit, T* Bout) {
    const int nr = pl->G.nr;
    if constexpr (NV == 1) {
        if (nr == 4) return launch_tiled<T, NV, S, 4>(c, pl, F, Binit, Bout);
        if (nr == 6) return launch_tiled<T, NV, S, 6>(c, pl, F, Binit, Bout);
        if (nr == 8) return launch_tiled<T, NV, S, 8>(c, pl, F, Binit, Bout);
        if constexpr (S <= 5) {
            if (nr == 10) return launch_tiled<T, NV, S, 10>(c, pl, F, Binit, Bout);
            if (nr == 12) return launch_tiled<T, NV, S, 12>(c, pl, F, Binit, Bout);
        }
    } else {
        if (nr == 2) return launch_tiled<T, NV, S, 2>(c, pl, F, Binit, Bout);
        if (nr == 4) return launch_tiled<T, NV, S, 4>(c, pl, F, Binit, Bout);
        if (nr == 6) return launch_tiled<T, NV, S, 6>(c, pl, F, Binit, Bout);
    }
    throw std::runtime_error("rhs_planned: unsupported round count");
}
template <class T, int NV>
void launch_tiled_s(rcppml_hip_ctx* c, const rcppml_rhs_plan* pl, const T* F, const T* Binit, T* Bout) {
    switch (pl->G.S) {
        case 2: launch_tiled_nr<T, NV, 2>(c, pl, F, Binit, Bout); break;
        case 3: launch_tiled_nr<T, NV, 3>(c, pl, F, Binit, Bout); break;
        case 4: launch_tiled_nr<T, NV, 4>(c, pl, F, Binit, Bout); break;
        case 5: launch_tiled_nr<T, NV, 5>(c, pl, F, Binit, Bout); break;
        case 6: launch_tiled_nr<T, NV, 6>(c, pl, F, Binit, Bout); break;
        case 8: launch_tiled_nr<T, NV, 8>(c, pl, F, Binit, Bout); break;
        default: throw std::runtime_error("rhs_planned: unsupported slot count");
    }
}

template <class T>
void run_plan(rcppml_hip_ctx* c, const rcppml_rhs_plan* pl, const T* F, T* B) {
    const RhsTiledGeom& G = pl->G;
    const bool ov = pl->ovnnz > 0;
    if (ov) {          // the spilled nonzeros first: their sums seed the accumulators
        if (rcppml_hip_rhs(c, pl->dtype, pl->ovptr, pl->ovrow, pl->ovval, G.ncols, F, pl->k, B) != 0)
            throw std::runtime_error("rhs_planned: overflow pass: " + rcppml_err());
    }
    const int NV = G.rowb / 256;
    const T* Binit = (G.P == 1 && ov) ? B : nullptr;
    T* Bout = G.P == 1 ? B : (T*)pl->Bp;
    if (NV == 1) launch_tiled_s<T, 1>(c, pl, F, Binit, Bout);
    else if (NV == 2) launch_tiled_s<T, 2>(c, pl, F, Binit, Bout);
    else throw std::runtime_error("rhs_planned: unsupported row size");
    if (G.P > 1) {
        const int64_t n4 = G.ncols * pl->k / RtVec<T>::N;
        hipLaunchKernelGGL(rhs_tiled_reduce_kernel<T>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, c->stream,
                           (const T*)pl->Bp, G.P, n4, ov ? 1 : 0, B);
        HIPCHK(hipGetLastError());
    }
}

template <class T>
rcppml_rhs_plan* build_plan(rcppml_hip_ctx* c, int dtype, const int* colptr, const int* rowidx, const T* vals, int64_t ncols,
                            int64_t nrows, int k, int partitions, int force_S) {
    const int rowb = k * (int)sizeof(T);
    if (rowb != 256 && rowb != 512) return nullptr;               // rows of one or two 256-byte slices
    if (ncols <= 0 || nrows <= 0) return nullptr;
    if (nrows * (int64_t)rowb < 16) return nullptr;
    int nnz_i = 0;
    HIPCHK(hipMemcpyAsync(&nnz_i, colptr + ncols, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (nnz_i <= 0) return nullptr;
    std::unique_ptr<rcppml_rhs_plan> pl(new rcppml_rhs_plan());
    pl->dtype = dtype; pl->k = k; pl->device = c->device; pl->nnz = nnz_i;
    RhsTiledGeom& G = pl->G;
    G.ncols = ncols; G.nrows = nrows; G.rowb = rowb;
    const int R = RT_SLAB_BYTES / rowb;
    G.rshift = 0;
    while ((1 << G.rshift) < R) ++G.rshift;
    G.ntiles = (int)((nrows + R - 1) / R);
    // partitions: one per XCD when the factor is far beyond an XCD's 4 MiB L2, else a single one
    int P = partitions;
    if (P <= 0) P = (nrows * (int64_t)rowb > (12ll << 20)) ? 8 : 1;
    if (P > G.ntiles) P = G.ntiles;
    G.P = P;
    // segment-length histogram -> S
    Tmp dh((RT_MAX_HIST + 1) * sizeof(unsigned long long));
    HIPCHK(hipMemsetAsync(dh.p, 0, (RT_MAX_HIST + 1) * sizeof(unsigned long long), c->stream));
    unsigned long long* dhist = (unsigned long long*)dh.p;
    int* dflag = (int*)(dhist + RT_MAX_HIST);
    const unsigned gcol = (unsigned)((ncols + 3) / 4);
    hipLaunchKernelGGL(rhs_tiled_hist_kernel, dim3(gcol), dim3(256), 0, c->stream, colptr, rowidx, ncols, G.rshift, dhist, dflag);
    HIPCHK(hipGetLastError());
    unsigned long long hh[RT_MAX_HIST + 1];
    HIPCHK(hipMemcpyAsync(hh, dh.p, sizeof(hh), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if ((int)(hh[RT_MAX_HIST] & 0xffffffffull) != 0) return nullptr;             // rows not sorted inside a column
    const double nseg = (double)ncols * (double)G.ntiles;
    const int cand[6] = {2, 3, 4, 5, 6, 8};
    int S = 0;
    double best = 0;
    for (int ci = 0; ci < 6; ++ci) {
        const int s = cand[ci];
        double ovf = 0;
        for (int b = s + 1; b < RT_MAX_HIST; ++b) ovf += (double)hh[b] * (b - s);
        const double cost = nseg * s + 6.0 * ovf;          // a spilled nonzero costs about six slot steps in the gather kernel
        if (S == 0 || cost < best) { S = s; best = cost; }
    }
    if (force_S > 0) {
        bool okS = false;
        for (int ci = 0; ci < 6; ++ci) okS |= cand[ci] == force_S;
        if (!okS) throw std::runtime_error("rhs_plan: slot count must be one of 2,3,4,5,6,8");
        S = force_S;
    }
    G.S = S;
    { const char* e = getenv("RCPPML_RT_DBG"); G.dbg = e ? atoi(e) : 0; }
    // workgroup shape: as close to one workgroup per CU (per partition: num_cu / P) as the column count allows
    const int NV = rowb / 256;
    const int64_t want_wg = std::max<int64_t>(1, c->num_cu / P);
    const int64_t need = (ncols + want_wg - 1) / want_wg;         // columns per workgroup
    int bestNW = 0, bestnr = 0;
    int64_t bestcap = -1, maxcap = -1;
    int maxNW = 0, maxnr = 0;
    for (int NW = 16; NW >= 8; --NW)
        for (int nr = 1; nr <= 12; ++nr) {
            if (!rt_shape_ok(NV, S, nr)) continue;
            const int64_t cap = 4ll * nr * NW;
            if (cap >= need && (bestcap < 0 || cap < bestcap)) { bestcap = cap; bestNW = NW; bestnr = nr; }
            if (cap > maxcap) { maxcap = cap; maxNW = NW; maxnr = nr; }
        }
    if (bestcap < 0) { bestNW = maxNW; bestnr = maxnr; bestcap = maxcap; }        // more workgroups than CUs
    G.NW = bestNW; G.nr = bestnr;
    G.ncb = (int)((ncols + bestcap - 1) / bestcap);

    // overflow column pointers
    Tmp cnt(((size_t)ncols + 1) * sizeof(int));
    HIPCHK(hipMemsetAsync(cnt.p, 0, ((size_t)ncols + 1) * sizeof(int), c->stream));
    hipLaunchKernelGGL(rhs_tiled_ovcount_kernel, dim3(gcol), dim3(256), 0, c->stream, colptr, rowidx, ncols, G.rshift, S, (int*)cnt.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMalloc((void**)&pl->ovptr, ((size_t)ncols + 1) * sizeof(int)));
    {
        size_t sb = 0;
        HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, sb, (int*)cnt.p, pl->ovptr, (int)(ncols + 1), c->stream));
        Tmp st(sb);
        HIPCHK(hipcub::DeviceScan::ExclusiveSum(st.p, sb, (int*)cnt.p, pl->ovptr, (int)(ncols + 1), c->stream));
        int ovn = 0;
        HIPCHK(hipMemcpyAsync(&ovn, pl->ovptr + ncols, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        pl->ovnnz = ovn;
    }
    pl->ov_fraction = (double)pl->ovnnz / (double)pl->nnz;
    if (force_S <= 0 && pl->ov_fraction > 0.35) return nullptr;   // too irregular for fixed slots: the gather kernel is the better tool
    pl->nslots = (int64_t)G.ncb * G.ntiles * G.NW * (int64_t)(G.nr * S) * 4;
    pl->fill = (double)(pl->nnz - pl->ovnnz) / (double)pl->nslots;
    HIPCHK(hipMalloc(&pl->svals, (size_t)pl->nslots * sizeof(T)));
    HIPCHK(hipMalloc((void**)&pl->soffs, (size_t)pl->nslots * sizeof(uint16_t)));
    HIPCHK(hipMemsetAsync(pl->svals, 0, (size_t)pl->nslots * sizeof(T), c->stream));
    HIPCHK(hipMemsetAsync(pl->soffs, 0, (size_t)pl->nslots * sizeof(uint16_t), c->stream));
    HIPCHK(hipMalloc((void**)&pl->ovrow, (size_t)std::max<int64_t>(pl->ovnnz, 1) * sizeof(int)));
    HIPCHK(hipMalloc(&pl->ovval, (size_t)std::max<int64_t>(pl->ovnnz, 1) * sizeof(T)));
    hipLaunchKernelGGL(rhs_tiled_fill_kernel<T>, dim3(gcol), dim3(256), 0, c->stream, colptr, rowidx, vals, G, (T*)pl->svals,
                       pl->soffs, (const int*)pl->ovptr, pl->ovrow, (T*)pl->ovval);
    HIPCHK(hipGetLastError());
    if (G.P > 1) HIPCHK(hipMalloc(&pl->Bp, (size_t)G.P * (size_t)ncols * (size_t)k * sizeof(T)));
    HIPCHK(hipStreamSynchronize(c->stream));          // temporaries die here
    return pl.release();
}

}  // namespace

extern "C" int rcppml_hip_rhs_plan_create(rcppml_hip_ctx* c, int dtype, const int* col_ptr, const int* row_idx,
                                          const void* values, int64_t ncols, int64_t nrows, int k, int partitions, int slots,
                                          rcppml_rhs_plan** out) {
    try {
        if (!out) throw std::runtime_error("rhs_plan_create: null output");
        *out = nullptr;
        HIPCHK(hipSetDevice(c->device));
        if (ncols > 0x7ffffff0ll || nrows > 0x7ffffff0ll) return 0;
        if (dtype == RCPPML_F32) *out = build_plan<float>(c, dtype, col_ptr, row_idx, (const float*)values, ncols, nrows, k, partitions, slots);
        else *out = build_plan<double>(c, dtype, col_ptr, row_idx, (const double*)values, ncols, nrows, k, partitions, slots);
        return 0;
    }
    RCPPML_CATCH_RET
}

extern "C" void rcppml_hip_rhs_plan_destroy(rcppml_rhs_plan* plan) { delete plan; }

extern "C" int rcppml_hip_rhs_plan_info(const rcppml_rhs_plan* pl, double* out10) {
    if (!pl || !out10) return 1;
    const RhsTiledGeom& G = pl->G;
    out10[0] = G.P; out10[1] = G.NW; out10[2] = G.nr; out10[3] = G.S; out10[4] = G.ncb; out10[5] = G.ntiles;
    out10[6] = (double)pl->nslots; out10[7] = (double)pl->ovnnz; out10[8] = pl->fill;
    out10[9] = (double)pl->nslots * ((pl->dtype == RCPPML_F32 ? 4 : 8) + 2);      // bytes of the slot stream
    return 0;
}

extern "C" int rcppml_hip_rhs_planned(rcppml_hip_ctx* c, const rcppml_rhs_plan* plan, const void* F, void* B) {
    try {
        if (!plan) throw std::runtime_error("rhs_planned: null plan");
        if (plan->device != c->device) throw std::runtime_error("rhs_planned: plan belongs to another device");
        if (reinterpret_cast<uintptr_t>(F) % 16 || reinterpret_cast<uintptr_t>(B) % 16)
            throw std::runtime_error("rhs_planned: F and B must be 16-byte aligned");
        HIPCHK(hipSetDevice(c->device));
        if (plan->dtype == RCPPML_F32) run_plan<float>(c, plan, (const float*)F, (float*)B);
        else run_plan<double>(c, plan, (const double*)F, (double*)B);
        return 0;
    }
    RCPPML_CATCH_RET
}
