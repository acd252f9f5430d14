// rhs_tiled_launch.hip.h -- compiled shapes of rhs_tiled_kernel and their dispatch (included by one translation unit per
// precision so the two sets of instantiations build in parallel).
#pragma once
#include <mutex>
#include "rhs_plan.hip.h"

namespace rt_launch {
using namespace rk;

constexpr int RT_MAX_LDS = 160 * 1024;          // two 64 KiB tiles of F + the two-stage slot ring
inline int rt_dyn_lds(int NW, int NR, int S, int tsize) { return 2 * RT_SLAB_BYTES + 2 * NW * NR * S * 4 * (tsize + 2); }

// Compiled shapes (NV, S, NW, NR).  NW = waves per workgroup: 16 (four per SIMD, 128 VGPRs) or 12 (three per SIMD, 168
// VGPRs: more rounds per wave = more output columns per workgroup = fewer bytes of F streamed per multiply-add and less
// per-tile overhead per step).  Columns per workgroup = 4 NR NW.  A shape must also fit its slot ring into the 32 KiB
// of LDS behind the two tiles of F (rt_dyn_lds).
//   NV = 1 (256-byte rows): NW 16: NR in {2,4,6,8,10,12} (NR >= 10 only for S <= 5);  NW 12: NR in {8,12,16,20} (S <= 6 / S <= 4 for NR >= 16 / 20)
//   NV = 4 (1 KiB rows: fp64 k = 128): NW 16: NR in {1,2,3}; NW 12: NR in {2,4}
//   NV = 2 (512-byte rows): NW 16: NR in {2,4,6};                                     NW 12: NR in {4,6,8,10} (NR 8 / 10 only for S <= 6 / S <= 4)
inline bool shape_ok(int NV, int S, int NW, int NR, int tsize) {
    if (rt_dyn_lds(NW, NR, S, tsize) > RT_MAX_LDS) return false;
    const bool s_ok = S == 2 || S == 3 || S == 4 || S == 5 || S == 6 || S == 8;
    if (!s_ok) return false;
    if (NV == 1 && NW == 16) return NR == 2 || NR == 4 || NR == 6 || NR == 8 || ((NR == 10 || NR == 12) && S <= 5);
    if (NV == 1 && NW == 12) return NR == 8 || NR == 12 || (NR == 16 && S <= 6) || (NR == 20 && S <= 4);
    if (NV == 2 && NW == 16) return NR == 2 || (NR == 4 && S <= 6) || (NR == 6 && S <= 4);   // (beyond: fp32 spills)
    if (NV == 2 && NW == 12) return NR == 4 || NR == 6 || (NR == 8 && S <= 6) || (NR == 10 && S <= 4);
    if (NV == 4 && NW == 16) return NR == 1 || NR == 2 || NR == 3;
    if (NV == 4 && NW == 12) return NR == 2 || NR == 4;
    return false;
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
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RT_MAX_LDS));
        done[device & 63] = true;
    }
}

template <class T, int NV, int S, int NW, int NR>
void launch_tiled(rcppml_hip_ctx* c, const rcppml_rhs_plan* pl, const T* F, const T* Binit, T* Bout) {
    constexpr auto kern = rhs_tiled_kernel<T, NV, S, NR, NW>;
    set_lds_once<kern>(c->device);
    const RhsTiledGeom& G = pl->G;
    if (!shape_ok(NV, S, NW, NR, (int)sizeof(T))) throw std::runtime_error("rhs_planned: shape exceeds the LDS");
    hipLaunchKernelGGL(kern, dim3((unsigned)(G.P * G.ncb)), dim3(64 * NW), rt_dyn_lds(NW, NR, S, (int)sizeof(T)), c->stream,
                       (const T*)pl->svals, (const uint16_t*)pl->soffs, F, G, Binit, Bout);
    HIPCHK(hipGetLastError());
}
template <class T, int NV, int S>
void launch_tiled_nr(rcppml_hip_ctx* c, const rcppml_rhs_plan* pl, const T* F, const T* Binit, T* Bout) {
    const int nr = pl->G.nr, nw = pl->G.NW;
#define RT_SH(W, N) if (nw == W && nr == N) return launch_tiled<T, NV, S, W, N>(c, pl, F, Binit, Bout);
    if constexpr (NV == 1) {
        RT_SH(16, 2) RT_SH(16, 4) RT_SH(16, 6) RT_SH(16, 8) RT_SH(12, 8) RT_SH(12, 12)
        if constexpr (S <= 5) { RT_SH(16, 10) RT_SH(16, 12) }
        if constexpr (S <= 6) { RT_SH(12, 16) }
        if constexpr (S <= 4) { RT_SH(12, 20) }
    } else if constexpr (NV == 2) {
        RT_SH(16, 2) RT_SH(12, 4) RT_SH(12, 6)
        if constexpr (S <= 6) { RT_SH(16, 4) RT_SH(12, 8) }
        if constexpr (S <= 4) { RT_SH(16, 6) }
        if constexpr (S <= 4) { RT_SH(12, 10) }
    } else {
        RT_SH(16, 1) RT_SH(16, 2) RT_SH(16, 3) RT_SH(12, 2) RT_SH(12, 4)
    }
#undef RT_SH
    throw std::runtime_error("rhs_planned: shape not compiled");
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
void launch_tiled_any(rcppml_hip_ctx* c, const rcppml_rhs_plan* pl, const T* F, const T* Binit, T* Bout) {
    const int NV = pl->G.rowb / 256;
    if (NV == 1) launch_tiled_s<T, 1>(c, pl, F, Binit, Bout);
    else if (NV == 2) launch_tiled_s<T, 2>(c, pl, F, Binit, Bout);
    else if (NV == 4) {
        if constexpr (std::is_same<T, double>::value) launch_tiled_s<T, 4>(c, pl, F, Binit, Bout);      // 1 KiB rows: fp64 k = 128
        else throw std::runtime_error("rhs_planned: unsupported row size");
    } else throw std::runtime_error("rhs_planned: unsupported row size");
}
}  // namespace rt_launch

// defined in ops_rhs_tiled.hip (float) and ops_rhs_tiled_f64.hip (double)
void rcppml_rt_launch_f32(rcppml_hip_ctx* c, const rcppml_rhs_plan* pl, const float* F, const float* Binit, float* Bout);
void rcppml_rt_launch_f64(rcppml_hip_ctx* c, const rcppml_rhs_plan* pl, const double* F, const double* Binit, double* Bout);
