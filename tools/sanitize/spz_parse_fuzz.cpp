// spz_parse_fuzz.cpp -- the plugin's host-side `.spz` parser (rcppml_amd/csrc/spz_parse.hpp, the code ops_spz.hip runs before
// it launches the device decoder) compiled for the CPU with -fsanitize=address,undefined and driven with well-formed files and
// a deterministic corpus of damaged ones: truncations at every interesting length, byte flips, header / chunk-descriptor
// fields overwritten with extreme values.  A damaged file must either parse (to whatever it now says) or be refused with a
// ParseError; every table the device decoder will index (jobs: offsets and sizes; col_ptr; row permutation) must lie inside the
// file / the matrix.  Any out-of-bounds read, signed overflow or misaligned access aborts through the sanitizers.
// Usage: spz_parse_fuzz file.spz [file.spz ...]      (built and run by tools/sanitize/run.sh and tests/test_sanitize_cpu.py)
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <string>
#include "../../rcppml_amd/csrc/spz_parse.hpp"
using namespace rcppml_spz;

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint64_t rnd() { rng_state += 0x9e3779b97f4a7c15ull; uint64_t z = rng_state; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
                        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }

// what the device decoder relies on after a successful parse
static void check_parsed(const SpzParsed& P, uint64_t size) {
    const SpzHeader& h = P.h;
    if (P.col_ptr.size() != (size_t)h.n + 1 || P.seg_ptr.size() != (size_t)h.n + 1) { std::fprintf(stderr, "col_ptr size\n"); std::abort(); }
    for (size_t j = 0; j + 1 < P.seg_ptr.size(); ++j)
        if (P.seg_ptr[j] > P.seg_ptr[j + 1] || P.seg_ptr[j] < 0 || (uint64_t)P.seg_ptr[j + 1] > h.nnz) { std::fprintf(stderr, "seg_ptr not monotone inside nnz\n"); std::abort(); }
    for (const SpzJob& j : P.jobs) {
        const bool ok = in_file(size, j.table_off, 2) && in_file(size, j.enc_off, j.enc_size) && (j.ov_size == 0 || in_file(size, j.ov_off, j.ov_size)) &&
                        j.out_off <= h.nnz && j.count <= h.nnz - j.out_off;
        if (!ok) { std::fprintf(stderr, "job outside the file / the matrix\n"); std::abort(); }
    }
}

static int try_parse(const std::vector<uint8_t>& f, bool must_succeed) {
    // exact-size heap copy: ASan then sees every read past the end
    uint8_t* buf = static_cast<uint8_t*>(std::malloc(f.size() ? f.size() : 1));
    if (!f.empty()) std::memcpy(buf, f.data(), f.size());
    int rc = 0;
    try {
        SpzParsed P = parse_file(buf, f.size());
        check_parsed(P, f.size());
    } catch (const ParseError& e) {
        rc = e.status;
        if (must_succeed) { std::fprintf(stderr, "well-formed file refused: %s\n", e.what); std::abort(); }
    }
    std::free(buf);
    return rc;
}

int main(int argc, char** argv) {
    long parsed = 0, refused = 0;
    for (int a = 1; a < argc; ++a) {
        std::ifstream in(argv[a], std::ios::binary);
        std::vector<uint8_t> f((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        if (f.empty()) { std::fprintf(stderr, "cannot read %s\n", argv[a]); return 2; }
        try_parse(f, true);
        auto tally = [&](const std::vector<uint8_t>& g) { (try_parse(g, false) == 0 ? parsed : refused) += 1; };
        // truncations: every length up to 400 bytes, then 200 random ones
        for (size_t len = 0; len < std::min<size_t>(400, f.size()); ++len) tally(std::vector<uint8_t>(f.begin(), f.begin() + len));
        const int scale = (int)std::max<uint64_t>(1, std::min<uint64_t>(30, (100ull << 20) / (30 * f.size())));   // ~100 MB of copies per file
        for (int t = 0; t < 7 * scale; ++t) tally(std::vector<uint8_t>(f.begin(), f.begin() + rnd() % f.size()));
        // single and multiple byte flips, biased towards the header, the chunk index and the metadata section
        SpzHeader h; std::memcpy(&h, f.data(), 128);
        for (int t = 0; t < 100 * scale; ++t) {
            std::vector<uint8_t> g = f;
            const int nflip = 1 + (int)(rnd() % 4);
            for (int q = 0; q < nflip; ++q) {
                size_t pos;
                switch (rnd() % 4) {
                    case 0: pos = rnd() % 128; break;
                    case 1: pos = h.chunk_index_offset < f.size() ? h.chunk_index_offset + rnd() % std::min<uint64_t>(48ull * std::max(1u, h.num_chunks), f.size() - h.chunk_index_offset) : rnd() % f.size(); break;
                    case 2: pos = h.metadata_offset && h.metadata_offset < f.size() ? h.metadata_offset + rnd() % (f.size() - h.metadata_offset) : rnd() % f.size(); break;
                    default: pos = rnd() % f.size();
                }
                g[pos] = (rnd() & 1) ? (uint8_t)rnd() : (uint8_t)(g[pos] ^ (1u << (rnd() % 8)));
            }
            tally(g);
        }
        // extreme values in every 4-byte word of the header and of the first chunk descriptors
        const uint32_t extremes[6] = {0u, 1u, 0x7fffffffu, 0x80000000u, 0xfffffffeu, 0xffffffffu};
        const size_t words = (128 + std::min<size_t>(4, h.num_chunks) * 48) / 4;
        for (size_t wd = 1; wd < words; ++wd)          // (word 0 is the magic)
            for (uint32_t v : extremes) {
                std::vector<uint8_t> g = f;
                const size_t pos = wd < 32 ? wd * 4 : (size_t)h.chunk_index_offset + (wd - 32) * 4;
                if (pos + 4 > g.size()) continue;
                std::memcpy(g.data() + pos, &v, 4);
                tally(g);
            }
    }
    std::printf("spz_parse_fuzz: %d well-formed file(s) parsed; damaged corpus: %ld parsed, %ld refused; no sanitizer report\n", argc - 1, parsed, refused);
    return 0;
}
