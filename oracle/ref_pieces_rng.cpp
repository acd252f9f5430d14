// oracle/ref_pieces_rng.cpp -- TEST INFRASTRUCTURE.  C-ABI harness around the REFERENCE'S OWN rng header, compiled from
// where it lies (/root/reference/inst/include/FactorNet/rng/rng.hpp; nothing is copied into this repo).  That header is
// the one piece of the hot path that builds without Eigen/Rcpp: its host-only Eigen overload sits behind
// `#ifndef __CUDACC__`, so it is compiled with the image's HIP compiler in host-only mode with the reference's own
// device-compilation branch selected (-D__CUDACC__).  Output: oracle/_ref/libref_rng.so (git-ignored, travels to the
// GPU box).  Used by tests/test_oracle_ref.py to pin the oracle's (and rcppml_amd.data's) SplitMix64 restatement --
// factor initialisation, SURVEY.md row a1 -- against the reference itself, bit for bit.
#include <FactorNet/rng/rng.hpp>
#include <cstdint>

extern "C" {
__attribute__((visibility("default"))) void ref_fill_uniform_f64(uint64_t seed, double* out, int rows, int cols) {
    FactorNet::rng::SplitMix64 r(seed);
    r.fill_uniform(out, rows, cols);
}
__attribute__((visibility("default"))) void ref_fill_uniform_f32(uint64_t seed, float* out, int rows, int cols) {
    FactorNet::rng::SplitMix64 r(seed);
    r.fill_uniform(out, rows, cols);
}
// the no-W_init path of initialize_factors: ONE stream fills W_T (k x m) and continues into H (k x n)
__attribute__((visibility("default"))) void ref_init_factors_f64(uint64_t seed, int k, int m, int n, double* W_T, double* H) {
    FactorNet::rng::SplitMix64 r(seed);
    r.fill_uniform(W_T, k, m);
    r.fill_uniform(H, k, n);
}
__attribute__((visibility("default"))) void ref_init_factors_f32(uint64_t seed, int k, int m, int n, float* W_T, float* H) {
    FactorNet::rng::SplitMix64 r(seed);
    r.fill_uniform(W_T, k, m);
    r.fill_uniform(H, k, n);
}
__attribute__((visibility("default"))) uint64_t ref_next(uint64_t* state) {
    FactorNet::rng::SplitMix64 r(1);
    r.set_state(*state);
    const uint64_t v = r.next();
    *state = r.state();
    return v;
}
__attribute__((visibility("default"))) uint64_t ref_hash(uint64_t seed, uint32_t i, uint32_t j) {
    return FactorNet::rng::SplitMix64::hash(seed, i, j);
}
__attribute__((visibility("default"))) int ref_is_holdout(uint64_t seed, uint32_t i, uint32_t j, uint64_t inv_prob) {
    return FactorNet::rng::SplitMix64::is_holdout(seed, i, j, inv_prob) ? 1 : 0;
}
}
