// spz_parse.hpp -- host-side parser of a StreamPress / SparsePress v2 `.spz` file: header, chunk index, per-column counts
// (varints) -> col_ptr, and the table of rANS streams ("jobs") the device decoder (ops_spz.hip) works through.  Plain C++17, no
// HIP: ops_spz.hip includes it, and tools/sanitize/spz_parse_fuzz.cpp compiles the same code for the CPU under ASan + UBSan and
// feeds it the fixtures and a corpus of truncated / corrupted files.
// Format references: streampress/format/header_v2.hpp, streampress/sparsepress_v2.hpp:897-1103 (decoder), rans.hpp:139-166.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

namespace rcppml_spz {

struct SpzHeader {                   // streampress/format/header_v2.hpp:118-154 (128 bytes, little endian)
    uint8_t magic[4]; uint16_t version, header_size; uint32_t m, n; uint64_t nnz; uint32_t chunk_cols, num_chunks,
        num_tables, table_log; uint8_t value_type, compression_level, row_sorted, col_sorted; uint32_t most_common_value;
    uint64_t chunk_index_offset, tables_offset, data_offset, transpose_offset, metadata_offset; uint32_t max_value;
    float density; uint8_t reserved[32];
};
static_assert(sizeof(SpzHeader) == 128, "v2 header is 128 bytes");
struct SpzChunk {                    // header_v2.hpp:229-247 (48 bytes)
    uint32_t col_start, num_cols, nnz, stream_offset[2], stream_size[2], decoded_gap_bytes, decoded_value_bytes;
    float quant_scale, quant_offset; uint32_t reserved;
};
static_assert(sizeof(SpzChunk) == 48, "v2 chunk descriptor is 48 bytes");

enum { JOB_GAPS = 0, JOB_INT = 1, JOB_QUANT = 2, JOB_PLANE = 3 };
struct SpzJob {                      // one rANS stream
    uint64_t table_off;              // serialized table: u16 n_symbols, then n_symbols u16 frequencies (rans.hpp:139-166)
    uint64_t enc_off;                // encoded bytes
    uint64_t ov_off;                 // varint escapes (valid when ov_size > 0)
    uint64_t out_off;                // first output element
    uint32_t enc_size, ov_size, count, kind, plane, bpv;
    float qscale, qoff;
};

struct ParseError { int status; const char* what; };

inline uint32_t rd32(const uint8_t* p) { uint32_t v; std::memcpy(&v, p, 4); return v; }

// rANS + escape block (sparsepress_v2.hpp:404-439): [table][enc_sz u32][enc bytes][ov_sz u32][varints]
// `in_file(off, len)`: the byte range [off, off + len) lies inside the file -- written so that it cannot wrap
inline bool in_file(uint64_t size, uint64_t off, uint64_t len) { return off <= size && len <= size - off; }

inline void add_escape_job(const uint8_t* data, uint64_t size, uint64_t off, uint64_t len, uint32_t count, int kind, uint64_t out_off,
                    float qs, float qo, std::vector<SpzJob>& jobs) {
    if (len == 0 || count == 0) return;                                    // :407-408 -> zeros (outputs are pre-zeroed)
    if (!in_file(size, off, len)) throw ParseError{5, "rANS block beyond end of file"};
    if (len < 2) throw ParseError{5, "truncated rANS block"};
    const uint32_t ns = data[off] | (data[off + 1] << 8);
    if (ns > 256) throw ParseError{5, "rANS table with more than 256 symbols"};
    uint64_t o = 2 + 2ull * ns;
    if (o + 4 > len) throw ParseError{5, "truncated rANS block"};          // the reference returns an empty vector and then indexes it
    SpzJob j{};
    j.table_off = off; j.count = count; j.kind = kind; j.out_off = out_off; j.qscale = qs; j.qoff = qo;
    j.enc_size = rd32(data + off + o); o += 4;
    j.enc_off = off + o;
    if (j.enc_size < 4 || j.enc_size > len - o) throw ParseError{5, "rANS payload beyond its block"};
    o += j.enc_size;
    if (o + 4 <= len) {
        j.ov_size = rd32(data + off + o); o += 4; j.ov_off = off + o;
        if (j.ov_size > len - o) throw ParseError{5, "escape bytes beyond their block"};
    }
    jobs.push_back(j);
}
// byte-shuffled floats (sparsepress_v2.hpp:442-476): [n_streams u8] then per plane [table_sz u32][table][enc_sz u32][enc]
inline void add_plane_jobs(const uint8_t* data, uint64_t size, uint64_t off, uint64_t len, uint32_t count, uint32_t bpv, uint64_t out_off,
                    std::vector<SpzJob>& jobs) {
    if (len == 0 || count == 0) return;
    if (!in_file(size, off, len)) throw ParseError{5, "byte-plane block beyond end of file"};
    uint64_t o = 0;
    const uint32_t ns = data[off + o++];
    if (ns > bpv) throw ParseError{5, "more byte planes than bytes per value"};
    for (uint32_t s = 0; s < ns; ++s) {
        if (o + 4 > len) return;                                            // :453 (remaining planes stay zero)
        const uint32_t tsz = rd32(data + off + o); o += 4;
        SpzJob j{};
        j.table_off = off + o; j.count = count; j.kind = JOB_PLANE; j.plane = s; j.bpv = bpv; j.out_off = out_off;
        if (tsz < 2 || tsz > len - o) throw ParseError{5, "bad byte-plane table"};
        const uint32_t nsym = data[j.table_off] | (data[j.table_off + 1] << 8);
        if (nsym > 256 || 2ull + 2ull * nsym > tsz) throw ParseError{5, "bad byte-plane table"};
        o += tsz;
        if (o + 4 > len) return;                                            // :460
        j.enc_size = rd32(data + off + o); o += 4;
        j.enc_off = off + o;
        if (j.enc_size < 4 || j.enc_size > len - o) throw ParseError{5, "byte-plane payload beyond its block"};
        o += j.enc_size;
        jobs.push_back(j);
    }
}

// col_ptr: what the reference decoder returns; seg_ptr: where each column's gaps actually lie in the chunk's output range
// (identical for well-formed files; they differ only when a chunk's count section is garbage -- the reference encoder
// omits its size prefix for chunks without nonzeros, sparsepress_v2.hpp:94 vs :988-991 -- and then seg_ptr keeps the
// prefix-sum kernel inside the chunk)
// row_perm: the stored row permutation of a row-sorted file (metadata entry ROW_PERMUTATION, header_v2.hpp:107-112,
// :347-359); the reference decoder maps every decoded row through it, i[k] = perm[i[k]] where i[k] < perm.size()
// (sparsepress_v2.hpp:1089-1103) -- rows then need not ascend inside a column any more
struct SpzParsed { SpzHeader h; std::vector<int> col_ptr, seg_ptr; std::vector<SpzJob> jobs; uint32_t bpv = 0; std::vector<uint32_t> row_perm; };

// Metadata section: u32 entry count, then per entry {u8 key, u32 length, bytes} (header_v2.hpp:405-457); it ends 16 bytes
// (the footer) before the end of the file (sparsepress_v2.hpp:1089-1091).  Returns the FIRST row permutation, as the
// reference's getter does; a file that is flagged row_sorted but carries none decodes unpermuted (:1094).
inline std::vector<uint32_t> read_row_permutation(const uint8_t* data, uint64_t size, const SpzHeader& h) {
    std::vector<uint32_t> perm;
    if (!(h.metadata_offset > 0 && h.metadata_offset < size)) return perm;
    if (size < 16 + h.metadata_offset) return perm;                       // (the reference's size_t difference would wrap here)
    const uint8_t* p = data + h.metadata_offset;
    const uint8_t* end = data + size - 16;
    if (end - p < 4) return perm;
    const uint32_t n = rd32(p); p += 4;
    for (uint32_t e = 0; e < n && p < end; ++e) {
        const uint8_t key = *p++;
        if (end - p < 4) break;
        const uint32_t len = rd32(p); p += 4;
        if ((uint64_t)(end - p) < len) break;
        if (key == 2) { perm.resize(len / 4); if (len / 4) std::memcpy(perm.data(), p, (size_t)(len / 4) * 4); return perm; }
        p += len;
    }
    return perm;
}

inline int check_header(const uint8_t* data, uint64_t size, SpzHeader& h) {
    if (size < 6) return 3;                                                  // sp_gpu_bridge.cu:69-73
    uint16_t version; std::memcpy(&version, data + 4, 2);
    if (version != 2) return 4;                                              // :75-81
    if (size < 128 || std::memcmp(data, "SPRZ", 4) != 0) return 5;
    std::memcpy(&h, data, 128);
    return 0;
}

inline SpzParsed parse_file(const uint8_t* data, uint64_t size) {
    SpzParsed P;
    const int st = check_header(data, size, P.h);
    if (st) throw ParseError{st, "not a v2 .spz file"};
    const SpzHeader& h = P.h;
    if (h.row_sorted) P.row_perm = read_row_permutation(data, size, h);
    if (h.nnz > 0x7FFFFFFFull || h.n > 0x7FFFFFFEu || h.m > 0x7FFFFFFFu) throw ParseError{5, "matrix too large for int32 CSC indices"};
    // every column costs at least one varint byte in its chunk's count section and every 2048 columns a 48-byte descriptor: a
    // header whose n cannot fit the file is damaged -- refuse it before sizing col_ptr by it (found by tools/sanitize: a flipped
    // bit in n made the parser allocate 16 GB)
    if ((uint64_t)h.n > 16 * size + 4096 || (uint64_t)h.num_chunks * 48 > size) throw ParseError{5, "column / chunk count cannot fit the file"};
    if (!in_file(size, h.chunk_index_offset, (uint64_t)h.num_chunks * 48)) throw ParseError{5, "truncated chunk index"};   // sparsepress_v2.hpp:913-914
    if (h.data_offset > size) throw ParseError{5, "data section beyond end of file"};
    std::vector<SpzChunk> ch(h.num_chunks);
    if (h.num_chunks) std::memcpy(ch.data(), data + h.chunk_index_offset, (size_t)h.num_chunks * 48);
    P.col_ptr.assign((size_t)h.n + 1, 0);
    P.seg_ptr.assign((size_t)h.n + 1, 0);
    const int vt = h.value_type;
    P.bpv = vt == 3 ? 4 : vt == 4 ? 2 : vt == 6 ? 8 : 0;
    if (vt > 6) throw ParseError{5, "unknown value type"};
    uint64_t out = 0;
    for (uint32_t c = 0; c < h.num_chunks; ++c) {                            // sparsepress_v2.hpp:976-1084
        const SpzChunk& d = ch[c];
        if ((uint64_t)d.col_start + d.num_cols > h.n || out + d.nnz > h.nnz) throw ParseError{5, "chunk outside the matrix"};
        if (d.stream_offset[0] > size - h.data_offset || d.stream_offset[1] > size - h.data_offset) throw ParseError{5, "stream offset beyond end of file"};
        const uint64_t gp = h.data_offset + d.stream_offset[0], gs = d.stream_size[0];
        if (gs >= 4) {
            if (!in_file(size, gp, gs)) throw ParseError{5, "gap stream beyond end of file"};
            const uint32_t cc = rd32(data + gp);
            const uint8_t* ccp = data + gp + 4;
            uint64_t run = out;
            for (uint32_t j = 0; j < d.num_cols; ++j) {                      // varint column counts (:994-997)
                uint64_t v = 0; int sh = 0; uint8_t byte;
                do {
                    // (a chunk without nonzeros has no count section: the reference decoder reads whatever bytes follow --
                    //  replicated, but never past the end of the file)
                    if (ccp >= (d.nnz > 0 ? data + gp + gs : data + size)) throw ParseError{5, "column counts beyond their stream"};
                    byte = *ccp++; if (sh < 64) v |= (uint64_t)(byte & 0x7F) << sh; sh += 7;
                } while (byte & 0x80);
                if (d.nnz > 0 && (v > d.nnz || run - out > d.nnz - v)) throw ParseError{5, "column counts exceed the chunk's nonzeros"};
                P.col_ptr[d.col_start + j] = (int)(uint32_t)run;
                P.seg_ptr[d.col_start + j] = (int)(d.nnz > 0 ? std::min<uint64_t>(run, out + d.nnz) : out);
                run += (uint32_t)v;
            }
            if (d.nnz > 0) {
                if (run - out != d.nnz) throw ParseError{5, "column counts do not add up to the chunk's nonzeros"};
                if (gs < 4ull + cc) throw ParseError{5, "column-count section longer than the gap stream"};
                add_escape_job(data, size, gp + 4 + cc, gs - 4 - cc, d.nnz, JOB_GAPS, out, 0.f, 0.f, P.jobs);
            }
        } else {
            for (uint32_t j = 0; j < d.num_cols; ++j) P.col_ptr[d.col_start + j] = P.seg_ptr[d.col_start + j] = (int)(uint32_t)out;
        }
        const uint64_t vp = h.data_offset + d.stream_offset[1], vs = d.stream_size[1];
        if (vs && !in_file(size, vp, vs)) throw ParseError{5, "value stream beyond end of file"};
        if (vt <= 2) add_escape_job(data, size, vp, vs, d.nnz, JOB_INT, out, 0.f, 0.f, P.jobs);
        else if (vt == 5) add_escape_job(data, size, vp, vs, d.nnz, JOB_QUANT, out, d.quant_scale, d.quant_offset, P.jobs);
        else add_plane_jobs(data, size, vp, vs, d.nnz, P.bpv, out, P.jobs);
        out += d.nnz;
    }
    P.col_ptr[h.n] = P.seg_ptr[h.n] = (int)(uint32_t)h.nnz;                    // :1087
    for (size_t j = h.n; j-- > 0;) P.seg_ptr[j] = std::min(P.seg_ptr[j], P.seg_ptr[j + 1]);   // monotone
    return P;
}


}  // namespace rcppml_spz
