// ============================================================================
// kernels_dense.hip.h -- dense-input right-hand sides for gfx950, fp32: "skinny" GEMMs with k <= 128 output rows.
//   B (k x n) = F (k x m) A          dense_rhs_fwd_f32   (reference primitives::rhs on a dense A, fit_cpu.hpp:547-549)
//   B (k x m) = F (k x n) A^T        dense_rhs_bwd_f32   (detail::rhs_transpose, fit_cpu.hpp:783)
// A is column-major m x n and is the only large operand (m*n*4 bytes): it is streamed from HBM exactly once per
// product with 16-byte loads; F (k x rows, a few MB) is re-read through LDS / L2.  The arithmetic runs on
// v_mfma_f32_32x32x2_f32 tiles (M = 32 factor rows, N = 32 output columns, K = 2 reduction indices per instruction):
// lane l supplies A-operand[M = l & 31][K = l >> 5] and B-operand[K = l >> 5][N = l & 31]; D row = (v&3)+8(v>>2)+4(l>>5),
// column = l & 31.  The K slots of one instruction may be ANY two reduction indices as long as both operands agree,
// which is what lets every lane keep the 16 consecutive rows of its own column that one 64-byte line holds.
// Roofline: 2 k flops per 4 bytes of A = 32 flop/B at k = 64, above the f32-MFMA ridge (157 TF / 8 TB/s = 20 flop/B):
// bounded by the matrix cores at k >= 40, by HBM below.
// ============================================================================
#pragma once
#include "kernels.hip.h"

namespace rk {

constexpr int DENSE_KC = 32;          // reduction indices per chunk (forward)
constexpr int DENSE_KC_BWD = 16;      // ... (backward)

// ---- forward: one wavefront = 32 output columns j (lane & 31), the two lane halves take rows i0 + 16 h + s (s = 0..15).
// Block = 4 waves = 128 consecutive columns sharing the F chunk (32 rows x k, contiguous in memory) through LDS; the
// reduction over i is split over gridDim.y slices (partials summed in slice order by dense_reduce_f32) so that several
// waves per SIMD are in flight whatever n is.
template <int RT>                       // k <= 32 RT
__global__ __launch_bounds__(256) void dense_rhs_fwd_f32(const float* __restrict__ A, int64_t m, int64_t n,
                                                         const float* __restrict__ F, int k, int64_t ichunk,
                                                         float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float Fs[2][DENSE_KC * 32 * RT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
    const int64_t j = ((int64_t)blockIdx.x * 4 + wave) * 32 + col;
    const bool jok = j < n;
    const float* acol = A + (jok ? j : 0) * m;
    const bool vec_ok = (m % 4 == 0) && (reinterpret_cast<uintptr_t>(A) % 16 == 0);
    const int64_t ibeg = (int64_t)blockIdx.y * ichunk, iend = min(m, ibeg + ichunk);   // this slice of the reduction
    float* B = part + (int64_t)blockIdx.y * k * n;
    f32x16 acc[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[r][v] = 0.f;
    auto load_a = [&](int64_t i0, float (&a)[16]) {
        const int64_t base = i0 + 16 * half;
        if (jok && vec_ok && base + 16 <= iend) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(acol + base + 4 * q);
                a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int s = 0; s < 16; ++s) a[s] = (jok && base + s < iend) ? acol[base + s] : 0.f;
        }
    };
    // rows i0 .. i0+31 of F^T (32 k contiguous floats, padded to 32 RT per row): fetched into registers while the MFMAs of
    // the current chunk run and parked in LDS after them, so their latency is never waited for
    constexpr int FPT = DENSE_KC * 32 * RT / 256;
    auto fetch_f = [&](int64_t i0, float (&fr)[FPT]) {
#pragma unroll
        for (int q = 0; q < FPT; ++q) {
            const int t = threadIdx.x + 256 * q, row = t / (32 * RT), f = t % (32 * RT);
            fr[q] = (f < k && i0 + row < iend) ? F[(i0 + row) * k + f] : 0.f;
        }
    };
    auto park_f = [&](const float (&fr)[FPT], int buf) {
#pragma unroll
        for (int q = 0; q < FPT; ++q) Fs[buf][threadIdx.x + 256 * q] = fr[q];
    };
    float a[16], an[16], fr[FPT];
    load_a(ibeg, a);
    fetch_f(ibeg, fr);
    park_f(fr, 0);
    __syncthreads();
    int buf = 0;
    for (int64_t i0 = ibeg; i0 < iend; i0 += DENSE_KC) {
        const bool more = i0 + DENSE_KC < iend;
        if (more) { load_a(i0 + DENSE_KC, an); fetch_f(i0 + DENSE_KC, fr); }          // next chunk in flight during the MFMAs
        const float* fs = Fs[buf] + (16 * half) * (32 * RT) + col;
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int r = 0; r < RT; ++r)
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(fs[s * (32 * RT) + 32 * r], a[s], acc[r], 0, 0, 0);
        if (more) park_f(fr, buf ^ 1);
        __syncthreads();
        if (more) {
#pragma unroll
            for (int s = 0; s < 16; ++s) a[s] = an[s];
        }
        buf ^= 1;
    }
    if (jok) {
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int f = 32 * r + (v & 3) + 8 * (v >> 2) + 4 * half;
                if (f < k) B[j * k + f] = acc[r][v];
            }
    }
}

// ---- backward: output columns are ROWS i of A (contiguous in memory), the reduction runs over columns j.  One 16-byte
// load gives a lane 4 consecutive rows i = i0 + 4 c + e of one column j: they feed four interleaved column tiles
// (tile e holds the output columns i0 + 4 c + e, c = lane & 31), the lane halves take j and j + 1.  A block of 4 waves
// covers 512 rows; the reduction is split over gridDim.y slices whose partial (k x m) results go to `part` and are
// summed in slice order by dense_reduce_f32 (deterministic, unlike atomics).
template <int RT>
__global__ __launch_bounds__(256) void dense_rhs_bwd_f32(const float* __restrict__ A, int64_t m, int64_t n,
                                                         const float* __restrict__ F, int k, int64_t jchunk,
                                                         float* __restrict__ part) {
    constexpr int KCB = DENSE_KC_BWD;              // 16 columns j per chunk: 32 + 32 VGPRs of A in flight -> two waves per SIMD
    __shared__ __attribute__((aligned(16))) float Fs[2][KCB * 32 * RT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 31, half = lane >> 5;
    const int64_t i0 = ((int64_t)blockIdx.x * 4 + wave) * 128 + 4 * c;            // this lane's rows i0 .. i0+3
    const int64_t jbeg = (int64_t)blockIdx.y * jchunk, jend = min(n, jbeg + jchunk);
    const bool vec_ok = (m % 4 == 0) && (reinterpret_cast<uintptr_t>(A) % 16 == 0) && i0 + 4 <= m;
    f32x16 acc[RT][4];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[r][e][v] = 0.f;
    auto load_a = [&](int64_t j0, float4 (&a)[KCB / 2]) {            // step s: column j0 + 2 s + half
#pragma unroll
        for (int s = 0; s < KCB / 2; ++s) {
            const int64_t j = j0 + 2 * s + half;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < jend) {
                const float* p = A + j * m + i0;
                if (vec_ok) v = *reinterpret_cast<const float4*>(p);
                else { if (i0 < m) v.x = p[0]; if (i0 + 1 < m) v.y = p[1]; if (i0 + 2 < m) v.z = p[2]; if (i0 + 3 < m) v.w = p[3]; }
            }
            a[s] = v;
        }
    };
    constexpr int FPT = KCB * 32 * RT / 256;
    auto fetch_f = [&](int64_t j0, float (&fr)[FPT]) {
#pragma unroll
        for (int q = 0; q < FPT; ++q) {
            const int t = threadIdx.x + 256 * q, row = t / (32 * RT), f = t % (32 * RT);
            fr[q] = (f < k && j0 + row < jend) ? F[(j0 + row) * k + f] : 0.f;
        }
    };
    auto park_f = [&](const float (&fr)[FPT], int buf) {
#pragma unroll
        for (int q = 0; q < FPT; ++q) Fs[buf][threadIdx.x + 256 * q] = fr[q];
    };
    float4 a[KCB / 2], an[KCB / 2];
    float fr[FPT];
    load_a(jbeg, a);
    fetch_f(jbeg, fr);
    park_f(fr, 0);
    __syncthreads();
    int buf = 0;
    for (int64_t j0 = jbeg; j0 < jend; j0 += KCB) {
        const bool more = j0 + KCB < jend;
        if (more) { load_a(j0 + KCB, an); fetch_f(j0 + KCB, fr); }
        const float* fs = Fs[buf] + half * (32 * RT) + c;
#pragma unroll
        for (int s = 0; s < KCB / 2; ++s)
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const float fv = fs[(2 * s) * (32 * RT) + 32 * r];
                acc[r][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fv, a[s].x, acc[r][0], 0, 0, 0);
                acc[r][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fv, a[s].y, acc[r][1], 0, 0, 0);
                acc[r][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(fv, a[s].z, acc[r][2], 0, 0, 0);
                acc[r][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(fv, a[s].w, acc[r][3], 0, 0, 0);
            }
        if (more) park_f(fr, buf ^ 1);
        __syncthreads();
        if (more) {
#pragma unroll
            for (int s = 0; s < KCB / 2; ++s) a[s] = an[s];
        }
        buf ^= 1;
    }
    float* out = part + (int64_t)blockIdx.y * k * m;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t i = i0 + e;
        if (i < m) {
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int f = 32 * r + (v & 3) + 8 * (v >> 2) + 4 * half;
                    if (f < k) out[i * k + f] = acc[r][e][v];
                }
        }
    }
}

// ============================================================================ fp64 (parity mode)
// Same two products on v_mfma_f64_16x16x4_f64 tiles (M = 16 factor rows, N = 16 output columns, K = 4 reduction indices):
// lane l supplies A-operand[M = l & 15][K = l >> 4] and B-operand[K = l >> 4][N = l & 15]; D row = (l >> 4) + 4 v,
// column = l & 15.  2 k flops per 8 bytes of A = 16 flop/B at k = 64 against a 78.6 TF / 8 TB/s = 10 flop/B ridge.
typedef double f64x4 __attribute__((ext_vector_type(4)));

// forward: 16 output columns per wave; quarter q = lane >> 4 keeps rows i0 + 8 q + s (s = 0..7: one 64-byte line).
template <int RT>                       // k <= 16 RT
__global__ __launch_bounds__(256) void dense_rhs_fwd_f64(const double* __restrict__ A, int64_t m, int64_t n,
                                                         const double* __restrict__ F, int k, int64_t ichunk,
                                                         double* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) double Fs[2][DENSE_KC * 16 * RT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, col = lane & 15, q4 = lane >> 4;
    const int64_t j = ((int64_t)blockIdx.x * 4 + wave) * 16 + col;
    const bool jok = j < n;
    const double* acol = A + (jok ? j : 0) * m;
    const bool vec_ok = (m % 2 == 0) && (reinterpret_cast<uintptr_t>(A) % 16 == 0);
    const int64_t ibeg = (int64_t)blockIdx.y * ichunk, iend = min(m, ibeg + ichunk);
    double* B = part + (int64_t)blockIdx.y * k * n;
    f64x4 acc[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[r][v] = 0.0;
    auto load_a = [&](int64_t i0, double (&a)[8]) {
        const int64_t base = i0 + 8 * q4;
        if (jok && vec_ok && base + 8 <= iend) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double2 v = *reinterpret_cast<const double2*>(acol + base + 2 * t);
                a[2 * t] = v.x; a[2 * t + 1] = v.y;
            }
        } else {
#pragma unroll
            for (int s = 0; s < 8; ++s) a[s] = (jok && base + s < iend) ? acol[base + s] : 0.0;
        }
    };
    constexpr int FPT = DENSE_KC * 16 * RT / 256;
    auto fetch_f = [&](int64_t i0, double (&fr)[FPT]) {
#pragma unroll
        for (int t = 0; t < FPT; ++t) {
            const int e = threadIdx.x + 256 * t, row = e / (16 * RT), f = e % (16 * RT);
            fr[t] = (f < k && i0 + row < iend) ? F[(i0 + row) * k + f] : 0.0;
        }
    };
    auto park_f = [&](const double (&fr)[FPT], int buf) {
#pragma unroll
        for (int t = 0; t < FPT; ++t) Fs[buf][threadIdx.x + 256 * t] = fr[t];
    };
    double a[8], an[8], fr[FPT];
    load_a(ibeg, a);
    fetch_f(ibeg, fr);
    park_f(fr, 0);
    __syncthreads();
    int buf = 0;
    for (int64_t i0 = ibeg; i0 < iend; i0 += DENSE_KC) {
        const bool more = i0 + DENSE_KC < iend;
        if (more) { load_a(i0 + DENSE_KC, an); fetch_f(i0 + DENSE_KC, fr); }
        const double* fs = Fs[buf] + (8 * q4) * (16 * RT) + col;
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int r = 0; r < RT; ++r)
                acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(fs[s * (16 * RT) + 16 * r], a[s], acc[r], 0, 0, 0);
        if (more) park_f(fr, buf ^ 1);
        __syncthreads();
        if (more) {
#pragma unroll
            for (int s = 0; s < 8; ++s) a[s] = an[s];
        }
        buf ^= 1;
    }
    if (jok) {
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int f = 16 * r + q4 + 4 * v;
                if (f < k) B[j * k + f] = acc[r][v];
            }
    }
}

// backward: one 16-byte load = 2 consecutive rows i = i0 + 2 c + e feeding two interleaved column tiles; the lane quarters
// take the columns j0 + 4 s + q.  A wave covers 32 rows, a block 128.
template <int RT>
__global__ __launch_bounds__(256) void dense_rhs_bwd_f64(const double* __restrict__ A, int64_t m, int64_t n,
                                                         const double* __restrict__ F, int k, int64_t jchunk,
                                                         double* __restrict__ part) {
    constexpr int KCB = DENSE_KC_BWD;
    __shared__ __attribute__((aligned(16))) double Fs[2][KCB * 16 * RT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, q4 = lane >> 4;
    const int64_t i0 = ((int64_t)blockIdx.x * 4 + wave) * 32 + 2 * c;
    const int64_t jbeg = (int64_t)blockIdx.y * jchunk, jend = min(n, jbeg + jchunk);
    const bool vec_ok = (m % 2 == 0) && (reinterpret_cast<uintptr_t>(A) % 16 == 0) && i0 + 2 <= m;
    f64x4 acc[RT][2];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[r][e][v] = 0.0;
    auto load_a = [&](int64_t j0, double2 (&a)[KCB / 4]) {
#pragma unroll
        for (int s = 0; s < KCB / 4; ++s) {
            const int64_t j = j0 + 4 * s + q4;
            double2 v = make_double2(0.0, 0.0);
            if (j < jend) {
                const double* p = A + j * m + i0;
                if (vec_ok) v = *reinterpret_cast<const double2*>(p);
                else { if (i0 < m) v.x = p[0]; if (i0 + 1 < m) v.y = p[1]; }
            }
            a[s] = v;
        }
    };
    constexpr int FPT = KCB * 16 * RT / 256;
    auto fetch_f = [&](int64_t j0, double (&fr)[FPT]) {
#pragma unroll
        for (int t = 0; t < FPT; ++t) {
            const int e = threadIdx.x + 256 * t, row = e / (16 * RT), f = e % (16 * RT);
            fr[t] = (f < k && j0 + row < jend) ? F[(j0 + row) * k + f] : 0.0;
        }
    };
    auto park_f = [&](const double (&fr)[FPT], int buf) {
#pragma unroll
        for (int t = 0; t < FPT; ++t) Fs[buf][threadIdx.x + 256 * t] = fr[t];
    };
    double2 a[KCB / 4], an[KCB / 4];
    double fr[FPT];
    load_a(jbeg, a);
    fetch_f(jbeg, fr);
    park_f(fr, 0);
    __syncthreads();
    int buf = 0;
    for (int64_t j0 = jbeg; j0 < jend; j0 += KCB) {
        const bool more = j0 + KCB < jend;
        if (more) { load_a(j0 + KCB, an); fetch_f(j0 + KCB, fr); }
        const double* fs = Fs[buf] + q4 * (16 * RT) + c;
#pragma unroll
        for (int s = 0; s < KCB / 4; ++s)
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const double fv = fs[(4 * s) * (16 * RT) + 16 * r];
                acc[r][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fv, a[s].x, acc[r][0], 0, 0, 0);
                acc[r][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fv, a[s].y, acc[r][1], 0, 0, 0);
            }
        if (more) park_f(fr, buf ^ 1);
        __syncthreads();
        if (more) {
#pragma unroll
            for (int s = 0; s < KCB / 4; ++s) a[s] = an[s];
        }
        buf ^= 1;
    }
    double* out = part + (int64_t)blockIdx.y * k * m;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int64_t i = i0 + e;
        if (i < m) {
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int f = 16 * r + q4 + 4 * v;
                    if (f < k) out[i * k + f] = acc[r][e][v];
                }
        }
    }
}

template <class T>
__global__ __launch_bounds__(256) void dense_reduce(const T* __restrict__ part, int64_t count, int slices, T* __restrict__ B) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= count) return;
    T s = T(0);
    for (int q = 0; q < slices; ++q) s += part[(int64_t)q * count + t];
    B[t] = s;
}

__global__ __launch_bounds__(256) void dense_reduce_f32(const float* __restrict__ part, int64_t count, int slices, float* __restrict__ B) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= count) return;
    float s = 0.f;
    for (int q = 0; q < slices; ++q) s += part[(int64_t)q * count + t];
    B[t] = s;
}

}  // namespace rk
