// oracle/ref_pieces_spz.cpp -- TEST INFRASTRUCTURE.  C-ABI harness around the REFERENCE'S OWN StreamPress v2 codec
// (/root/reference/inst/include/streampress/sparsepress_v2.hpp and what it includes: plain C++, Eigen- and R-free),
// compiled from where it lies.  Output: oracle/_ref/libref_spz.so.  Pins the oracle's restatement of the v2 DECODER
// (oracle/spz_oracle.cpp) and produces the .spz fixtures of tests/golden/make_spz_vectors.py with
// the reference's ENCODER (the product never encodes).
#include <streampress/sparsepress_v2.hpp>
#include <cstring>

using namespace streampress;

extern "C" {
// decode: returns 0 and fills p (n+1), i (nnz), x (nnz); m / n / nnz must have been obtained from ref_spz_info
__attribute__((visibility("default"))) int ref_spz_info(const uint8_t* data, uint64_t size, uint32_t* m, uint32_t* n,
                                                        uint64_t* nnz, int* value_type, int* row_sorted, uint32_t* num_chunks) {
    try {
        if (size < 128) return 1;
        v2::FileHeader_v2 h = v2::FileHeader_v2::deserialize(data);
        *m = h.m; *n = h.n; *nnz = h.nnz; *value_type = h.value_type; *row_sorted = h.row_sorted; *num_chunks = h.num_chunks;
        return 0;
    } catch (...) { return 2; }
}
__attribute__((visibility("default"))) int ref_spz_decode(const uint8_t* data, uint64_t size, int threads, uint32_t* p,
                                                          uint32_t* i, double* x) {
    try {
        v2::DecompressConfig_v2 cfg; cfg.num_threads = threads;
        CSCMatrix M = v2::decompress_v2(data, size, cfg);
        std::memcpy(p, M.p.data(), sizeof(uint32_t) * M.p.size());
        std::memcpy(i, M.i.data(), sizeof(uint32_t) * M.i.size());
        std::memcpy(x, M.x.data(), sizeof(double) * M.x.size());
        return 0;
    } catch (...) { return 2; }
}
// encode with the reference compressor; returns the byte count (0 on failure); call with out = nullptr to size
__attribute__((visibility("default"))) uint64_t ref_spz_encode(uint32_t m, uint32_t n, uint64_t nnz, const uint32_t* p,
                                                               const uint32_t* i, const double* x, const char* precision,
                                                               uint32_t chunk_cols, uint8_t* out, uint64_t cap) {
    try {
        CSCMatrix M(m, n, nnz);
        std::memcpy(M.p.data(), p, sizeof(uint32_t) * (n + 1));
        std::memcpy(M.i.data(), i, sizeof(uint32_t) * nnz);
        std::memcpy(M.x.data(), x, sizeof(double) * nnz);
        v2::CompressConfig_v2 cfg; cfg.precision = precision; cfg.chunk_cols = chunk_cols;
        std::vector<uint8_t> buf = v2::compress_v2(M, cfg);
        if (out && buf.size() <= cap) std::memcpy(out, buf.data(), buf.size());
        return buf.size();
    } catch (...) { return 0; }
}
// the same with the encoder's row_sort option (rows renumbered by descending nonzero count, permutation in the metadata)
__attribute__((visibility("default"))) uint64_t ref_spz_encode_rowsort(uint32_t m, uint32_t n, uint64_t nnz, const uint32_t* p,
                                                                       const uint32_t* i, const double* x, const char* precision,
                                                                       uint32_t chunk_cols, uint8_t* out, uint64_t cap) {
    try {
        CSCMatrix M(m, n, nnz);
        std::memcpy(M.p.data(), p, sizeof(uint32_t) * (n + 1));
        std::memcpy(M.i.data(), i, sizeof(uint32_t) * nnz);
        std::memcpy(M.x.data(), x, sizeof(double) * nnz);
        v2::CompressConfig_v2 cfg; cfg.precision = precision; cfg.chunk_cols = chunk_cols; cfg.row_sort = true;
        std::vector<uint8_t> buf = v2::compress_v2(M, cfg);
        if (out && buf.size() <= cap) std::memcpy(out, buf.data(), buf.size());
        return buf.size();
    } catch (...) { return 0; }
}
}
