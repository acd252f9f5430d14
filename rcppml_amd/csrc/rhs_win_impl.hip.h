// rhs_win_impl.hip.h -- planner and launcher of the window form of the sparse right-hand side (kernels_rhs_win.hip.h).
// Included by one translation unit per precision (ops_rhs_win.hip, ops_rhs_win_f64.hip) so the two sets of kernel
// instantiations build in parallel.  Reference semantics: primitives/cpu/rhs.hpp:52-70, fused_nnls.hpp:109-114.
#pragma once
#include <algorithm>
#include <memory>
#include <mutex>
#include "rhs_plan.hip.h"
#include "scan.hip.h"

namespace rw_launch {
using namespace rk;

constexpr int RW_MAX_LDS = 160 * 1024;
constexpr int rw_sbh(int NR, int CLO, int tsize) { return (NR * (CLO + 1) * 4 * (tsize + 2) + 15) & ~15; }
constexpr int rw_dyn_lds(int NW, int NR, int CLO, int tsize) { return RW_ZROW + RW_RING + 2 * NW * rw_sbh(NR, CLO, tsize) + 1024; }   // + slack: slot reads past a block's last step

// Compiled shapes.  NW = 12 (three waves per SIMD, 168 VGPRs) or 16 (four, 128 VGPRs); columns per workgroup = 4 NR NW.
// A shape must fit two stages of slot blocks into the 32 KiB of LDS behind the ring.
#ifdef RW_DEV          // development build: a handful of shapes, compiles in a minute
#define RW_SHAPES_NV1(X) X(12, 17) X(16, 10) X(12, 8) X(8, 26) X(16, 13) X(12, 14)
#define RW_SHAPES_NV2(X) X(12, 8)
#define RW_SHAPES_NV4(X) X(12, 2)
constexpr int RW_MAX_CLO = 1;
#else
#define RW_SHAPES_NV1(X) X(12, 8) X(12, 12) X(12, 14) X(12, 17) X(16, 4) X(16, 8) X(16, 10) X(16, 13)
#define RW_SHAPES_NV2(X) X(12, 4) X(12, 6) X(12, 8) X(12, 9) X(12, 10) X(12, 12) X(16, 2) X(16, 4) X(16, 6) X(8, 13) X(8, 16) X(8, 20)
#define RW_SHAPES_NV4(X) X(12, 2) X(12, 4) X(12, 6) X(16, 1) X(16, 2) X(8, 8) X(8, 10)
constexpr int RW_MAX_CLO = 5;
#endif
// slot rates compiled: CLO + NHI / 4 with quarter steps up to 2 slots per phase, half steps above
// shapes hipcc cannot keep in registers (checked with tools/kres.sh: every compiled kernel must report ScratchSize 0)
constexpr bool shape_spills(int NV, int CLO, int NHI, int NW, int NR) {       // NHI < 0: unknown yet
    return NV == 2 && ((NW == 16 && NR == 6 && CLO >= 4) || (NW == 12 && NR == 10 && CLO >= 4) || (NW == 12 && NR == 12 && CLO >= 2) ||
                       (NW == 8 && NR == 20 && CLO <= 1 && NHI == 4));
}
constexpr bool rate_compiled(int CLO, int NHI) {
    return CLO >= 0 && CLO <= RW_MAX_CLO && NHI >= 1 && NHI <= 4 && (CLO <= 1 || NHI == 2 || NHI == 4);
}

inline bool shape_ok(int NV, int CLO, int NW, int NR, int tsize, int NHI = -1) {
    if (CLO < 0 || CLO > RW_MAX_CLO) return false;
    if (rw_dyn_lds(NW, NR, CLO, tsize) > RW_MAX_LDS) return false;
    // register budget: accumulators 4 NV NR (fp32 and fp64 alike: 16 bytes per lane and slice) + read buffers + slots
    const int regs = 4 * NV * NR + 32 + 2 * ((NR * (CLO + 1) + 15) / 16) + 24;
    if (regs > (NW == 8 ? 250 : (NW == 12 ? 164 : 124)) || shape_spills(NV, CLO, NHI, NW, NR)) return false;
    bool ok = false;
#define RW_X(W, N) ok |= (NW == W && NR == N);
    if (NV == 1) { RW_SHAPES_NV1(RW_X) } else if (NV == 2) { RW_SHAPES_NV2(RW_X) } else if (NV == 4) { RW_SHAPES_NV4(RW_X) }
#undef RW_X
    return ok;
}

// One flag set PER KERNEL (the template parameter is the kernel itself, not its type: every instantiation of one precision has
// the same function type, and a flag keyed on the type would give only the first shape launched on a device its opt-in to
// more than 64 KiB of dynamic LDS); per-device attribute, serialised (concurrent fits from host threads)
template <auto Kernel>
void set_lds_once(int device) {
    static std::mutex mu;
    static bool done[64] = {};
    std::lock_guard<std::mutex> lk(mu);
    if (!done[device & 63]) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RW_MAX_LDS));
        done[device & 63] = true;
    }
}

template <class T, int NV, int CLO, int NHI, int NW, int NR>
void launch_one(rcppml_hip_ctx* c, const rcppml_rhs_plan* pl, const T* F, T* Bout) {
    constexpr int regs = 4 * NV * NR + 32 + 2 * ((NR * (CLO + 1) + 15) / 16) + 24;
    if constexpr (rw_dyn_lds(NW, NR, CLO, (int)sizeof(T)) > RW_MAX_LDS || regs > (NW == 8 ? 250 : (NW == 12 ? 164 : 124)) || !rate_compiled(CLO, NHI) || shape_spills(NV, CLO, NHI, NW, NR)) {
        throw std::runtime_error("rhs_planned: window shape not compiled");
    } else {
        constexpr auto kern = rhs_win_kernel<T, NV, CLO, NHI, NR, NW>;
        set_lds_once<kern>(c->device);
        const RhsWinGeom& G = pl->WG;
        hipLaunchKernelGGL(kern, dim3((unsigned)(G.P * G.ncb)), dim3(64 * NW), rw_dyn_lds(NW, NR, CLO, (int)sizeof(T)), c->stream,
                           (const char*)pl->svals, F, G, Bout);
        HIPCHK(hipGetLastError());
    }
}
template <class T, int NV, int CLO, int NHI>
void launch_shape(rcppml_hip_ctx* c, const rcppml_rhs_plan* pl, const T* F, T* Bout) {
    const int nr = pl->WG.nr, nw = pl->WG.NW;
#define RW_X(W, N) if (nw == W && nr == N) return launch_one<T, NV, CLO, NHI, W, N>(c, pl, F, Bout);
    if constexpr (NV == 1) { RW_SHAPES_NV1(RW_X) } else if constexpr (NV == 2) { RW_SHAPES_NV2(RW_X) } else { RW_SHAPES_NV4(RW_X) }
#undef RW_X
    throw std::runtime_error("rhs_planned: window shape not compiled");
}
template <class T, int NV, int CLO>
void launch_nhi(rcppml_hip_ctx* c, const rcppml_rhs_plan* pl, const T* F, T* Bout) {
    switch (pl->WG.nhi) {
        case 1: return launch_shape<T, NV, CLO, 1>(c, pl, F, Bout);
        case 2: return launch_shape<T, NV, CLO, 2>(c, pl, F, Bout);
        case 3: return launch_shape<T, NV, CLO, 3>(c, pl, F, Bout);
        case 4: return launch_shape<T, NV, CLO, 4>(c, pl, F, Bout);
        default: throw std::runtime_error("rhs_planned: unsupported slot rate");
    }
}
template <class T, int NV>
void launch_clo(rcppml_hip_ctx* c, const rcppml_rhs_plan* pl, const T* F, T* Bout) {
    switch (pl->WG.clo) {
        case 0: return launch_nhi<T, NV, 0>(c, pl, F, Bout);
        case 1: return launch_nhi<T, NV, 1>(c, pl, F, Bout);
#ifndef RW_DEV
        case 2: return launch_nhi<T, NV, 2>(c, pl, F, Bout);
        case 3: return launch_nhi<T, NV, 3>(c, pl, F, Bout);
        case 4: return launch_nhi<T, NV, 4>(c, pl, F, Bout);
        case 5: return launch_nhi<T, NV, 5>(c, pl, F, Bout);
#endif
        default: throw std::runtime_error("rhs_planned: unsupported slot rate");
    }
}

template <class T>
void set_values(rcppml_hip_ctx* c, rcppml_rhs_plan* pl, const T* vals) {
    if (!pl->dest) throw std::runtime_error("rhs_plan_set_values: the plan was not created without values");
    hipLaunchKernelGGL(rw_values_kernel<T>, dim3((unsigned)((pl->nnz + 255) / 256)), dim3(256), 0, c->stream, vals, (const unsigned*)pl->dest,
                       pl->nnz, (T*)pl->svals, (T*)pl->ovval);
    HIPCHK(hipGetLastError());
    pl->vals = vals;
}

}  // namespace rw_launch
// one translation unit per (precision, row size): ops_rhs_win_f32_nv1.hip, ...
void rcppml_rw_launch_f32_nv1(rcppml_hip_ctx* c, const rcppml_rhs_plan* pl, const float* F, float* Bout);
void rcppml_rw_launch_f32_nv2(rcppml_hip_ctx* c, const rcppml_rhs_plan* pl, const float* F, float* Bout);
void rcppml_rw_launch_f64_nv1(rcppml_hip_ctx* c, const rcppml_rhs_plan* pl, const double* F, double* Bout);
void rcppml_rw_launch_f64_nv2(rcppml_hip_ctx* c, const rcppml_rhs_plan* pl, const double* F, double* Bout);
void rcppml_rw_launch_f64_nv4(rcppml_hip_ctx* c, const rcppml_rhs_plan* pl, const double* F, double* Bout);
namespace rw_launch {

// run_plan<T> (the launch of the tile kernel + the finishing pass): rhs_win_finish.hip.h, one translation unit

// rate code: 4 * clo + nhi (slots per column and phase in quarters); 0 = choose
template <class T>
rcppml_rhs_plan* build_plan(rcppml_hip_ctx* c, int dtype, const int* colptr, const int* rowidx, const T* vals, int64_t ncols,
                            int64_t nrows, int k, int partitions, int rate_code) {
    const int rowb = k * (int)sizeof(T);
    if (rowb != 256 && rowb != 512 && !(rowb == 1024 && sizeof(T) == 8)) return nullptr;
    if (ncols <= 0 || nrows <= 0 || nrows * (int64_t)rowb < 16) return nullptr;
    int nnz_i = 0;
    HIPCHK(hipMemcpyAsync(&nnz_i, colptr + ncols, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (nnz_i <= 0) return nullptr;
    std::unique_ptr<rcppml_rhs_plan> pl(new rcppml_rhs_plan());
    pl->kind = 1; pl->dtype = dtype; pl->k = k; pl->device = c->device; pl->nnz = nnz_i;
    RhsWinGeom& G = pl->WG;
    G.ncols = ncols; G.nrows = nrows; G.rowb = rowb;
    G.R = RW_TB / rowb;
    G.ntiles = (int)((nrows + G.R - 1) / G.R);
    const int NV = rowb / 256;
    const int tsize = (int)sizeof(T);

    // ---- workgroup shape and row partitions.  Model (cycles): a phase costs every SIMD  waves x steps x c_step  + c_phase
    // (barrier skew, pipeline refill, LDS-DMA issue); workgroups run in rounds of num_cu; the partial slabs of P > 1 partitions
    // cost a pass of (P + 1) k ncols elements in the finishing kernel.
    const double lam = (double)pl->nnz / ((double)ncols * (double)G.ntiles);       // nonzeros per (column, tile)
    const double rate_est = rate_code > 0 ? rate_code / 4.0 : std::max(0.5, 1.45 * lam);
    // measured on C2 (rocprofv3, rates 1.0 .. 2.0 at one shape): 9.8 ns per step and SIMD-resident wave, 0.44 us per phase that
    // is not steps, ~8 us of prologue + epilogue per workgroup
    const double c_step = NV == 1 ? 23.5 : (NV == 2 ? 39.0 : 74.0), c_phase = 1050.0;
    double best_t = -1;
    int bNW = 0, bNR = 0, bP = 0;
    const char* eNW = exp_env("RCPPML_RW_NW"); const char* eNR = exp_env("RCPPML_RW_NR");      // -DRCPPML_EXPERIMENTS builds only
    for (int NW = 16; NW >= 8; NW -= 4)
        for (int nr = 1; nr <= 48; ++nr) {
            if (eNW && eNR && (NW != atoi(eNW) || nr != atoi(eNR))) continue;
            const int clo_need = (int)std::ceil(rate_est - 1e-9) - 1;
            if (!shape_ok(NV, std::max(0, clo_need), NW, nr, tsize)) continue;
            const int64_t cap = 4ll * nr * NW;
            const int64_t ncb = (ncols + cap - 1) / cap;
            for (int P = 1; P <= 16; ++P) {
                if (partitions > 0 && P != std::min(partitions, G.ntiles)) continue;
                if (P > G.ntiles) break;
                if (partitions <= 0 && P > 1 && G.ntiles / P < 8) break;
                const double nph = (double)G.ntiles / P;
                const double rounds = std::ceil((double)(ncb * P) / (double)c->num_cu);
                double t_wg = (nph * ((NW / 4) * nr * rate_est * c_step + c_phase) + 19000.0) / 2400.0;      // microseconds
                // XCD locality: workgroup b runs on XCD b % 8 and streams partition b % P, so an XCD sees P / gcd(8, P) partitions =
                // |F| / gcd(8, P) bytes; beyond its 4 MiB L2 the tiles come from the Infinity Cache instead (C2's W side: P = 10 ran
                // 4-8 % slower than P = 8 and doubled the fabric reads)
                int g8 = 1;
                for (int d = 8; d >= 1; d >>= 1) if (P % d == 0) { g8 = d; break; }
                if ((double)nrows * rowb / g8 > 3.5 * 1048576.0 && (double)nrows * rowb > 3.5 * 1048576.0) t_wg *= 1.15;
                const double t_fin = (P > 1 ? (double)(P + 1) * (double)ncols * rowb / 4.0e6 + 4.0 : 0.0);
                const double t = rounds * t_wg + t_fin;
                if (best_t < 0 || t < best_t) { best_t = t; bNW = NW; bNR = nr; bP = P; }
            }
        }
    if (best_t < 0) return nullptr;
    G.NW = bNW; G.nr = bNR; G.P = bP;
    const int64_t cap = 4ll * bNR * bNW;
    G.ncb = (int)((ncols + cap - 1) / cap);
    G.maxph = 0;
    for (int p = 0; p < G.P; ++p) G.maxph = std::max(G.maxph, rw_t0(G, p + 1) - rw_t0(G, p));
    { const char* e = exp_env("RCPPML_RW_DBG"); G.dbg = e ? atoi(e) : 0; }         // -DRCPPML_EXPERIMENTS builds only
    G.maxph = (G.maxph + 3) & ~3;            // the kernel runs whole groups of four phases (the last ones on empty slots)

    // ---- slot rate: survey the overflow of the candidate rates, minimise slots + 8 x overflow
    RwCand cand{};
    auto add_rate = [&](int q) {            // q quarters of a slot per phase; whole rates run as (clo - 1, all four phases "hi")
        int clo = q / 4, nhi = q % 4;
        if (nhi == 0) { clo -= 1; nhi = 4; }
        if (clo < 0 || !rate_compiled(clo, nhi) || !shape_ok(NV, clo, bNW, bNR, tsize, nhi) || cand.n >= RW_MAXCAND) return false;
        cand.clo[cand.n] = clo; cand.nhi[cand.n] = nhi; ++cand.n;
        return true;
    };
    if (rate_code > 0) {
        if (!add_rate(rate_code)) throw std::runtime_error("rhs_plan: slot rate not available for this shape");
    } else {
        for (int q = 2; q <= 4 * (RW_MAX_CLO + 1); ++q) {
            const double r = q / 4.0;
            if (r < 0.95 * lam || r > 2.6 * lam + 1.0) continue;
            if (q > 12 && (q & 1)) continue;                       // quarter steps only below 3 slots per phase
            add_rate(q);
        }
        if (cand.n == 0) return nullptr;
    }
    DevTmp dsv(c, (RW_MAXCAND + 2) * sizeof(unsigned long long));
    HIPCHK(hipMemsetAsync(dsv.p, 0, (RW_MAXCAND + 2) * sizeof(unsigned long long), c->stream));
    unsigned long long* dov = (unsigned long long*)dsv.p;
    int* dflag = (int*)(dov + RW_MAXCAND + 1);
    const int64_t nseg = ncols * (int64_t)G.P;                                      // (column, partition) walks
    const unsigned gseg = (unsigned)((nseg + 255) / 256);
    unsigned long long hov[RW_MAXCAND + 2] = {};
    if (cand.n > 1) {                  // several candidate rates: their overflow on a sample of ~8 192 columns
        const int64_t stride = std::max<int64_t>(1, ncols / 8192);
        const int64_t nsamp = (ncols + stride - 1) / stride;
        hipLaunchKernelGGL(rw_survey_kernel, dim3((unsigned)((nsamp * G.P * cand.n + 255) / 256)), dim3(256), 0, c->stream, colptr, rowidx, ncols,
                           stride, G.R, G.ntiles, G.P, G.nr, G.NW, rw_ub(rowb), cand, dov);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(hov, dsv.p, sizeof(hov), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    const double samp_nnz = std::max<double>(1.0, (double)hov[RW_MAXCAND]);
    int bi = -1;
    double bcost = 0;
    for (int i = 0; i < cand.n; ++i) {
        const double rate = cand.clo[i] + cand.nhi[i] / 4.0;
        const double cost = (double)G.ncb * cap * (double)G.ntiles * rate + 8.0 * (double)hov[i] / samp_nnz * (double)pl->nnz;
        if (bi < 0 || cost < bcost) { bi = i; bcost = cost; }
    }
    G.clo = cand.clo[bi]; G.nhi = cand.nhi[bi];
    if (!shape_ok(NV, G.clo, G.NW, G.nr, tsize, G.nhi)) return nullptr;
    const int chi = G.clo + 1;
    G.sb_lo = (G.nr * G.clo * 4 * (tsize + 2) + 15) & ~15;
    G.sb_hi = (G.nr * chi * 4 * (tsize + 2) + 15) & ~15;
    G.region = (int64_t)G.NW * ((int64_t)G.maxph * G.sb_lo + (int64_t)rw_hi_before(G.maxph, G.nhi) * (G.sb_hi - G.sb_lo));
    const double rate = G.clo + G.nhi / 4.0;
    pl->nslots = (int64_t)((double)G.ncb * cap * (double)G.ntiles * rate);
    const size_t stream_bytes = (size_t)G.ncb * G.P * (size_t)G.region + 4096;     // + slack: the last slot piece may read up to 1 KiB past a block
    pl->stream_bytes = (double)stream_bytes;
    if (rate_code <= 0) {
        // hypersparse inputs must not get a plan: the slot stream holds ncols x ntiles x rate slots whatever nnz is
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
        const bool big = stream_bytes > (size_t)(32u << 20);
        if ((double)stream_bytes > 0.5 * (double)free_b ||
            (big && ((double)pl->nnz / std::max<double>(1.0, (double)pl->nslots) < 0.25 || (double)stream_bytes > 4.0 * (double)pl->nnz * (sizeof(T) + 4.0) ||
                     best_t > (double)pl->nnz * 12.5e-6 * NV + 5.0)))
            return nullptr;
    }

    // exact overflow counts per (column, partition) for the chosen rate + the sortedness check
    DevTmp cnt(c, ((size_t)nseg + 1) * sizeof(int));
    HIPCHK(hipMemsetAsync((char*)cnt.p + (size_t)nseg * sizeof(int), 0, sizeof(int), c->stream));
    hipLaunchKernelGGL(rw_ovcount_kernel, dim3(gseg), dim3(256), 0, c->stream, colptr, rowidx, G, (int*)cnt.p, dflag);
    HIPCHK(hipGetLastError());
    DevTmp ovp_tmp(c, ((size_t)nseg + 1) * sizeof(int));
    exclusive_scan_i32(c, (const int*)cnt.p, (int*)ovp_tmp.p, nseg + 1);
    int h2[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(&h2[0], (int*)ovp_tmp.p + nseg, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(&h2[1], dflag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (h2[1] != 0) return nullptr;                                               // rows not sorted inside a column
    pl->ovnnz = h2[0];
    pl->ov_fraction = (double)pl->ovnnz / (double)pl->nnz;
    if (rate_code <= 0 && pl->ov_fraction > 0.25) return nullptr;                 // too irregular for a fixed rate: the gather kernel is the better tool
    pl->fill = (double)(pl->nnz - pl->ovnnz) / std::max<double>(1.0, (double)pl->nslots);

    // ---- one allocation for everything the plan owns (arena or hipMalloc, never a mixture)
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_ovptr = up(((size_t)nseg + 1) * sizeof(int));
    const size_t b_ovrow = up((size_t)std::max<int64_t>(pl->ovnnz, 1) * sizeof(int));
    const size_t b_ovval = up((size_t)std::max<int64_t>(pl->ovnnz, 1) * sizeof(T));
    const size_t b_slots = up(stream_bytes);
    const size_t b_bp = G.P > 1 ? up((size_t)G.P * (size_t)G.ncb * cap * (size_t)k * sizeof(T)) : 0;
    const bool deferred = vals == nullptr;
    const size_t b_dest = deferred ? up((size_t)pl->nnz * sizeof(unsigned)) : 0;
    if (deferred && (double)stream_bytes / sizeof(T) >= 2147483648.0) return nullptr;      // dest[] addresses 2^31 slot elements
    const size_t total = b_ovptr + b_ovrow + b_ovval + b_slots + b_bp + b_dest + 256;
    char* blk = (char*)c->arena_take(total);
    if (blk) pl->in_arena = true;
    else HIPCHK(hipMalloc((void**)&blk, total));
    pl->block = blk;
    pl->ovptr = (int*)blk; blk += b_ovptr;
    pl->ovrow = (int*)blk; blk += b_ovrow;
    pl->ovval = blk; blk += b_ovval;
    pl->svals = blk; blk += b_slots;
    pl->Bp = G.P > 1 ? blk : nullptr;
    blk += b_bp;
    pl->dest = deferred ? (unsigned*)blk : nullptr;

    // overflow pointers (already scanned), then the scatter
    HIPCHK(hipMemcpyAsync(pl->ovptr, ovp_tmp.p, ((size_t)nseg + 1) * sizeof(int), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemsetAsync(pl->svals, 0, stream_bytes, c->stream));
    hipLaunchKernelGGL(rw_fill_kernel<T>, dim3(gseg), dim3(256), 0, c->stream, colptr, rowidx, vals, G, (char*)pl->svals,
                       (const int*)pl->ovptr, pl->ovrow, (T*)pl->ovval, pl->dest);
    HIPCHK(hipGetLastError());
    pl->colptr = colptr; pl->rowidx = rowidx; pl->vals = vals;
    HIPCHK(hipStreamSynchronize(c->stream));          // temporaries die here
    return pl.release();
}

}  // namespace rw_launch
