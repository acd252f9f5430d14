// oracle/spz_oracle.cpp -- TEST INFRASTRUCTURE (never linked or loaded by the product).
// CPU restatement of the reference's StreamPress / SparsePress **v2 decoder** (SURVEY.md 8f N4), the format the
// reference's GPU reader accepts (src/sp_gpu_bridge.cu:76-84: "Only v2 .spz files supported for GPU decode").
// Parity: PINNED against the reference's own decoder and encoder compiled in place (oracle/_ref/libref_spz.so,
// tests/test_oracle_ref.py) on the bundled inst/extdata/pbmc3k.spz and on files of every value type.
//   header / chunk index      streampress/format/header_v2.hpp:118-154, :229-247
//   varint                    streampress/codec/varint.hpp:43-52
//   rANS table + byte decoder streampress/codec/rans.hpp:152-166 (deserialize), :128-136 (lookup), :216-247 (decoder)
//   rANS + escape block       streampress/sparsepress_v2.hpp:404-439
//   byte-shuffled floats      streampress/sparsepress_v2.hpp:442-476
//   chunk loop                streampress/sparsepress_v2.hpp:897-1090 (no partial reads; row permutation NOT restated:
//                             row-sorted files: the stored permutation is applied to the decoded rows)
#include <cstdint>
#include <cstring>
#include <vector>

#define SPZ_API extern "C" __attribute__((visibility("default")))

namespace {
struct Hdr {                         // header_v2.hpp:118-154 (128 bytes, little endian, natural alignment)
    uint8_t magic[4]; uint16_t version, header_size; uint32_t m, n; uint64_t nnz; uint32_t chunk_cols, num_chunks,
        num_tables, table_log; uint8_t value_type, compression_level, row_sorted, col_sorted; uint32_t most_common_value;
    uint64_t chunk_index_offset, tables_offset, data_offset, transpose_offset, metadata_offset; uint32_t max_value;
    float density; uint8_t reserved[32];
};
static_assert(sizeof(Hdr) == 128, "v2 header is 128 bytes");
struct Chunk {                       // header_v2.hpp:229-247 (48 bytes)
    uint32_t col_start, num_cols, nnz, stream_offset[2], stream_size[2], decoded_gap_bytes, decoded_value_bytes;
    float quant_scale, quant_offset; uint32_t reserved;
};
static_assert(sizeof(Chunk) == 48, "v2 chunk descriptor is 48 bytes");

inline uint32_t rd32(const uint8_t* p) { uint32_t v; std::memcpy(&v, p, 4); return v; }
inline uint64_t varint(const uint8_t*& p) {                    // varint.hpp:43-52
    uint64_t v = 0; int sh = 0;
    do { v |= (uint64_t)(*p & 0x7F) << sh; sh += 7; } while (*p++ & 0x80);
    return v;
}
struct Table { std::vector<uint16_t> freq, cum, slot2sym; };
inline Table read_table(const uint8_t*& p) {                   // rans.hpp:152-166, :128-136
    Table t; const uint32_t ns = p[0] | (p[1] << 8); p += 2;
    t.freq.resize(ns); t.cum.resize(ns); t.slot2sym.assign(1u << 14, 0);
    uint16_t c = 0;
    for (uint32_t s = 0; s < ns; ++s) { t.freq[s] = p[0] | (p[1] << 8); t.cum[s] = c; c += t.freq[s]; p += 2; }
    for (uint32_t s = 0; s < ns; ++s) for (uint32_t k = 0; k < t.freq[s]; ++k) t.slot2sym[t.cum[s] + k] = (uint16_t)s;
    return t;
}
inline void rans_decode(const uint8_t* data, size_t size, uint32_t* out, size_t count, const Table& t) {   // rans.hpp:216-247
    const uint8_t* p = data; const uint8_t* end = data + size;
    uint32_t x = 0;
    for (int b = 0; b < 4; ++b) x = (x << 8) | *p++;
    for (size_t i = 0; i < count; ++i) {
        const uint32_t slot = x & ((1u << 14) - 1), s = t.slot2sym[slot];
        x = t.freq[s] * (x >> 14) + slot - t.cum[s];
        while (x < (1u << 23) && p < end) x = (x << 8) | *p++;
        out[i] = s;
    }
}
inline std::vector<uint32_t> rans_escape(const uint8_t* data, size_t size, uint32_t count) {   // sparsepress_v2.hpp:404-439
    if (size == 0 || count == 0) return std::vector<uint32_t>(count, 0);
    const uint8_t* tp = data; Table t = read_table(tp);
    size_t off = tp - data;
    if (off + 4 > size) return {};
    const uint32_t enc = rd32(data + off); off += 4;
    std::vector<uint32_t> sym(count);
    rans_decode(data + off, enc, sym.data(), count, t);
    off += enc;
    if (off + 4 > size) return sym;
    const uint32_t ov = rd32(data + off); off += 4;
    if (ov > 0) { const uint8_t* o = data + off; for (uint32_t i = 0; i < count; ++i) if (sym[i] == 255) sym[i] = (uint32_t)varint(o); }
    return sym;
}
inline void byteshuffle(const uint8_t* data, size_t size, uint32_t count, uint8_t* out, uint32_t bpv) {   // :442-476
    if (size == 0 || count == 0) return;
    size_t off = 0; const uint8_t ns = data[off++];
    std::vector<uint32_t> st(count);
    for (int s = 0; s < ns; ++s) {
        if (off + 4 > size) return;
        const uint32_t tsz = rd32(data + off); off += 4;
        const uint8_t* tp = data + off; Table t = read_table(tp); off += tsz;
        if (off + 4 > size) return;
        const uint32_t enc = rd32(data + off); off += 4;
        rans_decode(data + off, enc, st.data(), count, t); off += enc;
        for (uint32_t k = 0; k < count; ++k) out[(size_t)k * bpv + s] = (uint8_t)st[k];
    }
}
inline float half_to_float(uint16_t h) {   // IEEE 754 binary16 -> binary32 (format/header_v2.hpp:661-686; exact)
    const uint32_t sign = (uint32_t)(h & 0x8000) << 16, exp = (h >> 10) & 0x1F, man = h & 0x3FF;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else { int e = -1; uint32_t mm = man; do { ++e; mm <<= 1; } while (!(mm & 0x400)); bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((mm & 0x3FF) << 13); }
    } else if (exp == 31) bits = sign | 0x7F800000u | (man << 13);
    else bits = sign | ((exp + 112) << 23) | (man << 13);
    float f; std::memcpy(&f, &bits, 4); return f;
}
}  // namespace

// 0 = ok; 1 = too small / bad magic / not v2; 3 = truncated
SPZ_API int oracle_spz_info(const uint8_t* data, uint64_t size, uint32_t* m, uint32_t* n, uint64_t* nnz, int* value_type) {
    if (size < 128 || std::memcmp(data, "SPRZ", 4) != 0) return 1;
    Hdr h; std::memcpy(&h, data, 128);
    if (h.version != 2) return 1;
    *m = h.m; *n = h.n; *nnz = h.nnz; *value_type = h.value_type;
    return 0;
}
SPZ_API int oracle_spz_decode(const uint8_t* data, uint64_t size, uint32_t* P, uint32_t* I, double* X) {
    uint32_t m, n; uint64_t nnz; int vt;
    const int st = oracle_spz_info(data, size, &m, &n, &nnz, &vt);
    if (st) return st;
    Hdr h; std::memcpy(&h, data, 128);
    if (size < h.chunk_index_offset + (uint64_t)h.num_chunks * 48) return 3;
    std::vector<Chunk> ch(h.num_chunks);
    std::memcpy(ch.data(), data + h.chunk_index_offset, (size_t)h.num_chunks * 48);
    uint64_t out = 0;
    for (uint32_t c = 0; c < h.num_chunks; ++c) {                                            // :976-1084
        const Chunk& d = ch[c];
        const uint8_t* gp = data + h.data_offset + d.stream_offset[0];
        const size_t gs = d.stream_size[0];
        if (gs >= 4) {
            const uint32_t cc = rd32(gp);
            const uint8_t* ccp = gp + 4; const uint8_t* rp = ccp + cc; const size_t rs = gs - 4 - cc;
            std::vector<uint32_t> cnt(d.num_cols);
            for (uint32_t j = 0; j < d.num_cols; ++j) cnt[j] = (uint32_t)varint(ccp);
            uint32_t run = (uint32_t)out;
            for (uint32_t j = 0; j < d.num_cols; ++j) { P[d.col_start + j] = run; run += cnt[j]; }
            if (d.nnz > 0) {
                std::vector<uint32_t> gaps = rans_escape(rp, rs, d.nnz);
                uint32_t g = 0;
                for (uint32_t j = 0; j < d.num_cols; ++j) {
                    uint32_t prev = 0;
                    for (uint32_t q = 0; q < cnt[j]; ++q) { const uint32_t row = prev + gaps[g]; I[out + g] = row; prev = row + 1; ++g; }
                }
            }
        } else {
            for (uint32_t j = 0; j < d.num_cols; ++j) P[d.col_start + j] = (uint32_t)out;
        }
        const uint8_t* vp = data + h.data_offset + d.stream_offset[1];
        const size_t vs = d.stream_size[1];
        switch (vt) {                                                                          // :1039-1079
            case 0: case 1: case 2: { auto v = rans_escape(vp, vs, d.nnz); for (uint32_t q = 0; q < d.nnz; ++q) X[out + q] = (double)v[q]; break; }
            case 3: { std::vector<float> f(d.nnz); byteshuffle(vp, vs, d.nnz, (uint8_t*)f.data(), 4); for (uint32_t q = 0; q < d.nnz; ++q) X[out + q] = (double)f[q]; break; }
            case 4: { std::vector<uint16_t> f(d.nnz); byteshuffle(vp, vs, d.nnz, (uint8_t*)f.data(), 2); for (uint32_t q = 0; q < d.nnz; ++q) X[out + q] = (double)half_to_float(f[q]); break; }
            case 5: { auto v = rans_escape(vp, vs, d.nnz); for (uint32_t q = 0; q < d.nnz; ++q) X[out + q] = (double)(d.quant_offset + d.quant_scale * v[q]); break; }
            case 6: { std::vector<double> f(d.nnz); byteshuffle(vp, vs, d.nnz, (uint8_t*)f.data(), 8); for (uint32_t q = 0; q < d.nnz; ++q) X[out + q] = f[q]; break; }
            default: return 1;
        }
        out += d.nnz;
    }
    P[n] = (uint32_t)nnz;
    // Row-sorted files: undo the permutation (sparsepress_v2.hpp:1089-1103).  Metadata section (header_v2.hpp:405-457): u32
    // entry count, then {u8 key, u32 length, bytes} per entry, ending 16 bytes (the footer) before the end of the file;
    // key 2 = ROW_PERMUTATION (u32 array); every decoded row r < perm.size() becomes perm[r].
    if (h.row_sorted && h.metadata_offset > 0 && h.metadata_offset < size && size >= 16 + h.metadata_offset) {
        const uint8_t* p = data + h.metadata_offset;
        const uint8_t* end = data + size - 16;
        if (end - p >= 4) {
            const uint32_t ne = rd32(p); p += 4;
            for (uint32_t e = 0; e < ne && p < end; ++e) {
                const uint8_t key = *p++;
                if (end - p < 4) break;
                const uint32_t len = rd32(p); p += 4;
                if ((uint64_t)(end - p) < len) break;
                if (key == 2) {
                    const uint32_t cnt = len / 4;
                    for (uint64_t q = 0; q < nnz; ++q)
                        if (I[q] < cnt) I[q] = rd32(p + 4 * (size_t)I[q]);
                    break;
                }
                p += len;
            }
        }
    }
    return 0;
}
