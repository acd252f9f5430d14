// ops_spz.hip -- StreamPress / SparsePress **v2** `.spz` reader for gfx950: file bytes -> device-resident
// CSC (int32 col_ptr, int32 row_idx, double values), the arrays rcppml_gpu_nmf_zerocopy_double consumes.
// Replaces reference src/sp_gpu_bridge.cu:41-157 (rcppml_sp_read_gpu / rcppml_sp_free_gpu, bound by R/sp_gpu.R); the format
// is the one streampress/sparsepress_v2.hpp:897-1090 decodes (256..2048-column chunks, per chunk one gap stream and one
// value stream, each byte-renormalised rANS with a 14-bit probability scale + varint escapes, floats byte-shuffled).
//
// Split of the work:
//   host    header, chunk index, per-column nonzero counts (varints, n of them) -> col_ptr; a table of rANS streams
//           ("jobs": where the frequency table, the encoded bytes and the escape bytes of each stream lie in the file)
//   device  the whole file is uploaded once; spz_rans_kernel decodes one stream per wavefront (rANS is a serial
//           recurrence on a 32-bit state: no parallelism inside a stream, all of it across the 2..9 streams of each
//           chunk); the 16384-slot decode table {symbol, freq, slot - cum} is built by the wave in LDS (128 KB, one
//           ds_read_b64 per symbol); symbols are scattered straight into their final place (gaps, doubles, or the byte
//           plane of a shuffled float).  spz_rows_kernel then turns gaps into row indices (wave prefix sums per
//           column) and spz_float_kernel widens shuffled fp32 / fp16 / fp64 payloads to double.
// Nothing here falls back to a host decoder: the entropy decoding itself runs on the GPU.
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <vector>
#include "common.hip.h"
#include "spz_parse.hpp"
using namespace rcppml_spz;



namespace {

constexpr int SPZ_PROB_BITS = 14;
constexpr uint32_t SPZ_SLOTS = 1u << SPZ_PROB_BITS;
constexpr uint32_t SPZ_L = 1u << 23;

// ---------------------------------------------------------------------------------------------------------------- kernels
// quant8 dequantisation offset + scale * q in fp32 as the reference's host code evaluates it: two roundings, never an fma
// (HIP's __fmul_rn / __fadd_rn are plain operators and contract under the default -ffp-contract=fast)
__device__ __forceinline__ float spz_dequant(float off, float scale, uint32_t q) {
#pragma clang fp contract(off)
    const float prod = scale * (float)q;
    return off + prod;
}
// One wavefront per rANS stream.  LDS: the 16384 x u64 decode table {symbol (8) | freq (15) << 8 | slot - cum (14) << 23},
// a block of SPZ_B decoded symbols and the input bytes that block can consume.
//
// rANS is a serial recurrence, so one lane decodes; the other 63 do the memory traffic around it, block by block:
//   all lanes   copy the next <= 2*SPZ_B + 4 encoded bytes (the most SPZ_B symbols can consume) global -> LDS
//   lane 0      decodes SPZ_B symbols.  Dependent chain per symbol: mask -> ds_read_b64 (table) -> unpack -> 24-bit
//               multiply-add -> at most two renormalisation bytes (after a step x >= freq * 2^9 >= 2^9) taken from an
//               8-byte big-endian register window whose successor word is already on its way from LDS.  No global
//               access sits on the chain (on gfx9 loads and stores share vmcnt: a global store per symbol made every
//               window refill wait for the newest store), escapes excepted (rare varints read from the file).
//   all lanes   flush the block to its final place, coalesced: gaps (u32), values (double, optionally de-quantised),
//               or one byte plane of a shuffled float.
constexpr int SPZ_B = 2048;
constexpr int SPZ_IN_WORDS = (2 * SPZ_B + 4 + 7) / 8 + 3;

__global__ __launch_bounds__(64) void spz_rans_kernel(const uint8_t* __restrict__ file, uint64_t file_padded,
                                                      const SpzJob* __restrict__ jobs, uint32_t* __restrict__ gaps,
                                                      double* __restrict__ values, uint8_t* __restrict__ raw) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    uint64_t* tab = reinterpret_cast<uint64_t*>(smem_raw);
    uint32_t* obuf = reinterpret_cast<uint32_t*>(smem_raw + (size_t)SPZ_SLOTS * 8);
    uint64_t* ibuf = reinterpret_cast<uint64_t*>(smem_raw + (size_t)SPZ_SLOTS * 8 + (size_t)SPZ_B * 4);
    __shared__ uint32_t cum[257];
    const SpzJob job = jobs[blockIdx.x];
    const int lane = threadIdx.x;
    const uint8_t* tb = file + job.table_off;
    const uint32_t ns = tb[0] | (tb[1] << 8);
    // cumulative frequencies (rans.hpp:157-163): lanes load, lane 0 accumulates (<= 256 terms)
    for (uint32_t s = lane; s < ns; s += 64) cum[s + 1] = tb[2 + 2 * s] | (tb[3 + 2 * s] << 8);
    __syncthreads();
    if (lane == 0) {
        uint32_t c = 0;
        for (uint32_t s = 0; s < ns; ++s) { const uint32_t f = cum[s + 1]; cum[s] = c; c += f; }
        cum[ns] = c;
    }
    __syncthreads();
    // slot -> symbol by binary search over the cumulative table (rans.hpp:128-136 build_lookup); slots past the total stay 0
    const uint32_t total = cum[ns];
    for (uint32_t slot = lane; slot < SPZ_SLOTS; slot += 64) {
        uint64_t e = 0;
        if (slot < total) {
            uint32_t lo = 0, hi = ns;                   // last s with cum[s] <= slot (zero-frequency symbols share a cum: take the last)
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (cum[mid] <= slot) lo = mid; else hi = mid; }
            const uint32_t f = cum[lo + 1] - cum[lo];
            e = (uint64_t)lo | ((uint64_t)f << 8) | ((uint64_t)(slot - cum[lo]) << 23);
        }
        tab[slot] = e;
    }
    const uint8_t* enc = file + job.enc_off;
    const uint64_t* fend = reinterpret_cast<const uint64_t*>(file + file_padded);
    const bool has_ov = job.kind != JOB_PLANE && job.ov_size > 0;
    // decoder state: UNIFORM -- every lane runs the same scalar program on the same data, so the recurrence lives in
    // SGPRs on the scalar ALU (v_readfirstlane after each LDS read tells the compiler so) and its branches are plain
    // s_cbranch_scc, not exec-mask sequences
    auto rfl = [](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    uint32_t x = 0, left = job.enc_size;                // left: bytes not yet consumed (rans.hpp:240: ptr_ < end_)
    const uint8_t* ov = file + job.ov_off;
    const uint8_t* const ov_end = ov + job.ov_size;
    for (uint32_t base = 0; base < job.count; base += SPZ_B) {
        const uint32_t cnt = min((uint32_t)SPZ_B, job.count - base);
        // ---- stage the input this block can consume, already byte-swapped to big-endian words
        const uint8_t* pos = enc + (job.enc_size - left);
        const uint64_t* gw = reinterpret_cast<const uint64_t*>(reinterpret_cast<uintptr_t>(pos) & ~uintptr_t(7));
        const int skip = (int)(reinterpret_cast<uintptr_t>(pos) & 7);
        __syncthreads();                                // the previous flush has read obuf
        for (int w = lane; w < SPZ_IN_WORDS; w += 64) ibuf[w] = (gw + w < fend) ? __builtin_bswap64(gw[w]) : 0ull;
        __syncthreads();
        {
            auto lds_word = [&](int w) -> uint64_t { const uint64_t v = ibuf[w]; return (uint64_t)rfl((uint32_t)v) | ((uint64_t)rfl((uint32_t)(v >> 32)) << 32); };
            uint64_t cur = lds_word(0) << (8 * skip);
            uint64_t nxt = lds_word(1);
            int nav = 8 - skip, widx = 1;
            auto next_byte = [&]() -> uint32_t {
                if (nav == 0) { cur = nxt; ++widx; nxt = lds_word(widx); nav = 8; }
                const uint32_t b = (uint32_t)(cur >> 56);
                cur <<= 8; --nav; --left;
                return b;
            };
            if (base == 0) {
#pragma unroll
                for (int b = 0; b < 4; ++b) x = (x << 8) | next_byte();
            }
            for (uint32_t i = 0; i < cnt; ++i) {
                const uint64_t e = tab[x & (SPZ_SLOTS - 1)];
                const uint32_t elo = rfl((uint32_t)e), ehi = rfl((uint32_t)(e >> 32));
                uint32_t sym = elo & 255u;
                x = ((elo >> 8) & 0x7FFFu) * (x >> SPZ_PROB_BITS) + ((elo >> 23) | (ehi << 9));
                if (x < SPZ_L && left > 0) {
                    x = (x << 8) | next_byte();
                    if (x < SPZ_L && left > 0) x = (x << 8) | next_byte();
                }
                if (has_ov && sym == 255) {                                  // sparsepress_v2.hpp:430-436: varint escape
                    uint32_t v = 0, sh = 0, byte;
                    do {                                                     // never past the escape section (malformed files)
                        byte = ov < ov_end ? rfl((uint32_t)*ov++) : 0u;
                        if (sh < 32) v |= (byte & 0x7F) << sh;
                        sh += 7;
                    } while (byte & 0x80);
                    sym = v;
                }
                obuf[i] = sym;                                               // all lanes, same value, same address
            }
        }
        __syncthreads();
        // ---- flush, coalesced
        const uint64_t o0 = job.out_off + base;
        if (job.kind == JOB_GAPS) { for (uint32_t i = lane; i < cnt; i += 64) gaps[o0 + i] = obuf[i]; }
        else if (job.kind == JOB_INT) { for (uint32_t i = lane; i < cnt; i += 64) values[o0 + i] = (double)obuf[i]; }                    // :1042-1044
        else if (job.kind == JOB_QUANT) { for (uint32_t i = lane; i < cnt; i += 64) values[o0 + i] = (double)spz_dequant(job.qoff, job.qscale, obuf[i]); }   // :1066-1068
        else { for (uint32_t i = lane; i < cnt; i += 64) raw[(o0 + i) * job.bpv + job.plane] = (uint8_t)obuf[i]; }                          // :471-473
    }
}

// gaps -> row indices, one wavefront per column: row_t = sum_{u <= t} gap_u + t (prev = row + 1; sparsepress_v2.hpp:1017-1025)
__global__ __launch_bounds__(256) void spz_rows_kernel(const int* __restrict__ colptr, int64_t ncols, int* __restrict__ rows) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * 4 + wave;
    if (j >= ncols) return;
    const int s = colptr[j], e = colptr[j + 1];
    uint32_t carry = 0;
    for (int t0 = s; t0 < e; t0 += 64) {
        const int t = t0 + lane;
        uint32_t v = t < e ? (uint32_t)rows[t] + (t > s ? 1u : 0u) : 0u;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(v, off, 64);
            if (lane >= off) v += o;
        }
        v += carry;
        if (t < e) rows[t] = (int)v;
        carry = __shfl(v, 63, 64);
    }
}

__device__ __forceinline__ float spz_half_to_float(uint32_t h) {             // format/header_v2.hpp:661-686 (exact widening)
    const uint32_t sign = (h >> 15) & 1;
    uint32_t exp = (h >> 10) & 0x1F, frac = h & 0x3FF, bits;
    if (exp == 0) {
        if (frac == 0) bits = sign << 31;
        else { exp = 1; while (!(frac & 0x400)) { frac <<= 1; exp--; } frac &= 0x3FF; bits = (sign << 31) | ((exp + 112) << 23) | (frac << 13); }
    } else if (exp == 31) bits = (sign << 31) | 0x7F800000u | (frac << 13);
    else bits = (sign << 31) | ((exp + 112) << 23) | (frac << 13);
    return __uint_as_float(bits);
}
// de-shuffled payload -> double (sparsepress_v2.hpp:1046-1078)
__global__ __launch_bounds__(256) void spz_float_kernel(const uint8_t* __restrict__ raw, int64_t nnz, int bpv, double* __restrict__ values) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nnz) return;
    if (bpv == 4) { uint32_t b; memcpy(&b, raw + t * 4, 4); values[t] = (double)__uint_as_float(b); }
    else if (bpv == 2) { const uint32_t b = raw[t * 2] | (raw[t * 2 + 1] << 8); values[t] = (double)spz_half_to_float(b); }
    else { double d; memcpy(&d, raw + t * 8, 8); values[t] = d; }
}

// rows of a decoded matrix must lie in [0, m): a malformed gap stream would otherwise send the transpose's histogram
// (atomicAdd on counts[row]) out of bounds in the zero-copy fit
// undo the row sort of a row-sorted file: i[k] = perm[i[k]] for i[k] < perm_len (sparsepress_v2.hpp:1098-1102)
__global__ __launch_bounds__(256) void spz_unpermute_rows_kernel(int* __restrict__ rows, int64_t nnz, const uint32_t* __restrict__ perm,
                                                                 uint32_t perm_len) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < nnz) {
        const uint32_t r = (uint32_t)rows[t];
        if (r < perm_len) rows[t] = (int)perm[r];
    }
}
__global__ __launch_bounds__(256) void spz_check_rows_kernel(const int* __restrict__ rows, int64_t nnz, int m, int* __restrict__ bad) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < nnz) {
        const int r = rows[t];
        if (r < 0 || r >= m) *bad = 1;
    }
}

void decode_to_device(rcppml_hip_ctx* c, const uint8_t* data, uint64_t size, const SpzParsed& P, int* d_col_ptr, int* d_row_idx,
                      double* d_values) {
    const SpzHeader& h = P.h;
    hipStream_t s = c->stream;
    HIPCHK(hipMemcpyAsync(d_col_ptr, P.col_ptr.data(), ((size_t)h.n + 1) * sizeof(int), hipMemcpyHostToDevice, s));
    if (h.nnz == 0) { HIPCHK(hipStreamSynchronize(s)); return; }
    // scratch: file bytes (+16 so the 8-byte window may read past the last stream), jobs, de-shuffle staging
    const size_t file_bytes = (size + 16 + 255) / 256 * 256;
    const size_t job_bytes = (P.jobs.size() * sizeof(SpzJob) + 255) / 256 * 256;
    const size_t raw_bytes = ((size_t)h.nnz * P.bpv + 255) / 256 * 256;
    const bool own_seg = P.seg_ptr != P.col_ptr;
    const size_t seg_bytes = own_seg ? ((size_t)h.n + 1) * sizeof(int) : 0;
    char* buf = static_cast<char*>(c->scratch(WS_GRAPH, file_bytes + job_bytes + raw_bytes + seg_bytes + 256));
    uint8_t* d_file = reinterpret_cast<uint8_t*>(buf);
    SpzJob* d_jobs = reinterpret_cast<SpzJob*>(buf + file_bytes);
    uint8_t* d_raw = reinterpret_cast<uint8_t*>(buf + file_bytes + job_bytes);
    int* d_seg = own_seg ? reinterpret_cast<int*>(buf + file_bytes + job_bytes + raw_bytes) : d_col_ptr;
    if (own_seg) HIPCHK(hipMemcpyAsync(d_seg, P.seg_ptr.data(), seg_bytes, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(d_file + size, 0, file_bytes - size, s));
    HIPCHK(hipMemcpyAsync(d_file, data, size, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(d_row_idx, 0, (size_t)h.nnz * sizeof(int), s));     // streams the file omits decode to zeros
    HIPCHK(hipMemsetAsync(d_values, 0, (size_t)h.nnz * sizeof(double), s));
    if (raw_bytes) HIPCHK(hipMemsetAsync(d_raw, 0, raw_bytes, s));
    if (!P.jobs.empty()) {
        HIPCHK(hipMemcpyAsync(d_jobs, P.jobs.data(), P.jobs.size() * sizeof(SpzJob), hipMemcpyHostToDevice, s));
        const size_t smem = (size_t)SPZ_SLOTS * sizeof(uint64_t) + (size_t)SPZ_B * 4 + (size_t)SPZ_IN_WORDS * 8;
        static DynSmemOnce once;
        once.ensure(reinterpret_cast<const void*>(spz_rans_kernel), smem, c->device);
        hipLaunchKernelGGL(spz_rans_kernel, dim3((unsigned)P.jobs.size()), dim3(64), smem, s, d_file, (uint64_t)file_bytes, d_jobs,
                           reinterpret_cast<uint32_t*>(d_row_idx), d_values, d_raw);
        HIPCHK(hipGetLastError());
    }
    hipLaunchKernelGGL(spz_rows_kernel, dim3((unsigned)(((int64_t)h.n + 3) / 4)), dim3(256), 0, s, d_seg, (int64_t)h.n, d_row_idx);
    HIPCHK(hipGetLastError());
    if (P.bpv) {
        hipLaunchKernelGGL(spz_float_kernel, dim3((unsigned)((h.nnz + 255) / 256)), dim3(256), 0, s, d_raw, (int64_t)h.nnz, (int)P.bpv, d_values);
        HIPCHK(hipGetLastError());
    }
    if (!P.row_perm.empty()) {                // row-sorted file: map the rows back (the file bytes on the device are done with)
        uint32_t* d_perm = reinterpret_cast<uint32_t*>(d_file);
        if (P.row_perm.size() * 4 > file_bytes) throw std::runtime_error("spz: row permutation larger than the file");   // cannot happen: it was read from it
        HIPCHK(hipMemcpyAsync(d_perm, P.row_perm.data(), P.row_perm.size() * 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(spz_unpermute_rows_kernel, dim3((unsigned)((h.nnz + 255) / 256)), dim3(256), 0, s, d_row_idx, (int64_t)h.nnz,
                           d_perm, (uint32_t)P.row_perm.size());
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(s));          // the host buffers (jobs, col_ptr) die with the caller's frame
}

}  // namespace

extern "C" int rcppml_hip_spz_info(const void* file_bytes, uint64_t size, int* m, int* n, int64_t* nnz, int* value_type) {
    try {
        SpzHeader h;
        const int st = check_header(static_cast<const uint8_t*>(file_bytes), size, h);
        if (st) { rcppml_err() = "spz_info: not a v2 .spz file"; return st; }
        *m = (int)h.m; *n = (int)h.n; *nnz = (int64_t)h.nnz; *value_type = h.value_type;
        return 0;
    }
    RCPPML_CATCH_RET
}

extern "C" int rcppml_hip_spz_decode(rcppml_hip_ctx* c, const void* file_bytes, uint64_t size, int* d_col_ptr, int* d_row_idx,
                                     double* d_values) {
    try {
        HIPCHK(hipSetDevice(c->device));
        const uint8_t* data = static_cast<const uint8_t*>(file_bytes);
        SpzParsed P;
        try { P = parse_file(data, size); }
        catch (const ParseError& e) { rcppml_err() = std::string("spz_decode: ") + e.what; return e.status; }
        decode_to_device(c, data, size, P, d_col_ptr, d_row_idx, d_values);
        return 0;
    }
    RCPPML_CATCH_RET
}

// reference src/sp_gpu_bridge.cu:41-123.  out_status: 0 ok, 1 cannot open, 2 short read, 3 too small, 4 not v2, 5 decode error.
extern "C" void rcppml_sp_read_gpu(const char** path_ptr, int* device_id, double* out_col_ptr_addr, double* out_row_idx_addr,
                                   double* out_values_addr, int* out_m, int* out_n, double* out_nnz, int* out_status) {
    *out_status = -1;
    void *dp = nullptr, *di = nullptr, *dx = nullptr;
    try {
        struct File { FILE* f; ~File() { if (f) std::fclose(f); } } file{std::fopen(*path_ptr, "rb")};      // closed on every path
        if (!file.f) { *out_status = 1; rcppml_err() = "sp_read_gpu: cannot open file"; return; }
        long fsz = -1;
        if (std::fseek(file.f, 0, SEEK_END) == 0) fsz = std::ftell(file.f);
        if (fsz < 0 || std::fseek(file.f, 0, SEEK_SET) != 0) { *out_status = 1; rcppml_err() = "sp_read_gpu: cannot determine the file size"; return; }
        const size_t size = (size_t)fsz;
        std::vector<uint8_t> bytes(size);
        const size_t got = size ? std::fread(bytes.data(), 1, size, file.f) : 0;
        if (got != size) { *out_status = 2; rcppml_err() = "sp_read_gpu: short read"; return; }
        SpzParsed P;
        try { P = parse_file(bytes.data(), size); }
        catch (const ParseError& e) { *out_status = e.status; rcppml_err() = std::string("sp_read_gpu: ") + e.what; return; }
        HIPCHK(hipSetDevice(*device_id));
        hipStream_t s = nullptr;
        HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        rcppml_hip_ctx* c = nullptr;
        if (rcppml_hip_ctx_create(&c, *device_id, s) != 0) { (void)hipStreamDestroy(s); *out_status = 5; return; }
        try {
            const size_t nnz = (size_t)P.h.nnz;
            HIPCHK(hipMalloc(&dp, ((size_t)P.h.n + 1) * sizeof(int)));
            HIPCHK(hipMalloc(&di, (nnz ? nnz : 1) * sizeof(int)));
            HIPCHK(hipMalloc(&dx, (nnz ? nnz : 1) * sizeof(double)));
            decode_to_device(c, bytes.data(), size, P, (int*)dp, (int*)di, (double*)dx);
            // what leaves this entry feeds the zero-copy fit: insist on a well-formed CSC (the low-level decode op keeps
            // the reference decoder's output bit for bit, garbage column pointers of nonzero-free chunks included)
            bool ok = P.col_ptr[0] == 0 && (uint64_t)(uint32_t)P.col_ptr[P.h.n] == P.h.nnz;
            for (size_t j = 0; ok && j < P.h.n; ++j) ok = P.col_ptr[j] >= 0 && P.col_ptr[j] <= P.col_ptr[j + 1];
            if (ok && nnz) {
                int* dbad = static_cast<int*>(c->scratch(WS_RED, sizeof(int)));
                HIPCHK(hipMemsetAsync(dbad, 0, sizeof(int), s));
                hipLaunchKernelGGL(spz_check_rows_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, s, (const int*)di, (int64_t)nnz, (int)P.h.m, dbad);
                HIPCHK(hipGetLastError());
                int bad = 0;
                HIPCHK(hipMemcpyAsync(&bad, dbad, sizeof(int), hipMemcpyDeviceToHost, s));
                HIPCHK(hipStreamSynchronize(s));
                ok = bad == 0;
            }
            if (!ok) throw std::runtime_error("decoded arrays are not a well-formed CSC (column pointers not monotone, or a row index outside the matrix)");
        } catch (...) {
            rcppml_hip_ctx_destroy(c); (void)hipStreamDestroy(s);
            throw;
        }
        rcppml_hip_ctx_destroy(c);
        (void)hipStreamDestroy(s);
        *out_m = (int)P.h.m; *out_n = (int)P.h.n; *out_nnz = (double)P.h.nnz;
        // device pointers travel as doubles: R has no 64-bit integer (sp_gpu_bridge.cu:101-105)
        *out_col_ptr_addr = (double)reinterpret_cast<uintptr_t>(dp);
        *out_row_idx_addr = (double)reinterpret_cast<uintptr_t>(di);
        *out_values_addr = (double)reinterpret_cast<uintptr_t>(dx);
        *out_status = 0;
    } catch (const std::exception& e) {
        if (dp) (void)hipFree(dp);
        if (di) (void)hipFree(di);
        if (dx) (void)hipFree(dx);
        rcppml_err() = std::string("sp_read_gpu: ") + e.what();
        *out_status = 5;
    }
}

// reference src/sp_gpu_bridge.cu:132-155
extern "C" void rcppml_sp_free_gpu(double* col_ptr_addr, double* row_idx_addr, double* values_addr, int* out_status) {
    *out_status = 0;
    for (double* a : {col_ptr_addr, row_idx_addr, values_addr}) {
        if (*a != 0.0) (void)hipFree(reinterpret_cast<void*>(static_cast<uintptr_t>(*a)));
        *a = 0.0;
    }
}
