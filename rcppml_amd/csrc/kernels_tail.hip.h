// kernels_tail.hip.h -- the iteration's tail in fewer launches (round 5); included after kernels.hip.h by ops_misc.hip / ops_gram.hip.
//
// Between two solves an MSE iteration runs row sums -> their final sum -> scaling -> Gram partials -> Gram sum (-> cross partials ->
// loss) plus the two kernels of the next solve's work order: nine launches on the W side, seven on the H side, ~10 us of HBM time
// in total but 4.5-9 us of launch, drain and cache write-back apiece (profiles/r05_summary.md).  INDEPENDENT kernels share a launch
// here, told apart by the block index: the work-order histogram beside the row sums, its scatter beside the scaling, the cross-term
// partials beside the Gram's final sum.  The bodies are the ones of the separate kernels (kernels.hip.h: same partial layouts, same
// summation orders), so every result is bit-identical to the separate launches.
//
// Measured and NOT kept (profiles/r05_tail_ab.txt): running the one-block final sums (row sums, loss) in the last block of the
// producing kernel to finish (device-scope acq_rel ticket per block).  A device-scope release / acquire on this part is an L2
// write-back + invalidate of the block's XCD: one per block made the iteration 3 % SLOWER than the separate launches (one per
// thread: 11 % slower); the final sums stay their own small launches.
#pragma once
#include "kernels.hip.h"

namespace rk {

// blocks [0, nb_order): order_hist of the solve that just ran; blocks [nb_order, nb_order + nb_norm): row-sum partials of X
// (VEC = 0: the scalar body)
template <class T, int VEC>
__global__ __launch_bounds__(256) void tail_norm_hist_kernel(const T* __restrict__ X, int k, int64_t ncols, int norm_type,
                                                              T* __restrict__ partial, unsigned nb_norm, const int* __restrict__ sweeps,
                                                              unsigned int* __restrict__ part, unsigned nb_order) {
    if (blockIdx.x < nb_order) { order_hist_body(sweeps, ncols, part, blockIdx.x, nb_order); return; }
    const unsigned bid = blockIdx.x - nb_order;
    if constexpr (VEC > 0) row_norm_partial_vec_body<T, VEC>(X, k, ncols, norm_type, partial, bid, nb_norm);
    else row_norm_partial_body<T>(X, k, ncols, norm_type, partial, bid, nb_norm);
}
// blocks [0, nb_order): order_scatter; the rest: scale_rows_from_sums (VEC = 0: scalar body)
template <class T, int VEC>
__global__ __launch_bounds__(256) void tail_scale_scatter_kernel(T* __restrict__ X, int k, int64_t total, const T* __restrict__ sums,
                                                                  int norm_type, T* __restrict__ d, unsigned nb_scale,
                                                                  const int* __restrict__ sweeps, int64_t ncols,
                                                                  const unsigned int* __restrict__ part, int* __restrict__ order,
                                                                  unsigned nb_order) {
    if (blockIdx.x < nb_order) { order_scatter_body(sweeps, ncols, part, order, blockIdx.x, nb_order); return; }
    const unsigned bid = blockIdx.x - nb_order;
    if constexpr (VEC > 0) scale_rows_from_sums_vec_body<T, VEC>(X, k, total, sums, norm_type, d, bid, nb_scale);
    else scale_rows_from_sums_body<T>(X, k, total, sums, norm_type, d, bid, nb_scale);
}
// fp32, k = 64: extract_scaling's second pass INSIDE the Gram's partial-tile kernel (gram_partial_f32_k64_body<STEPS, true>: the one
// lane that loads an element divides it by its row's d, stores it back and feeds the scaled value to the matrix cores) -- one pass
// over the factor and one launch less; blocks [0, nb_order): order_scatter; block nb_order also stores d
template <int STEPS>
__global__ __launch_bounds__(256) void tail_scale_gram_k64_kernel(float* __restrict__ X, int k, int64_t ncols, const float* __restrict__ sums,
                                                                   int norm_type, float* __restrict__ d, float* __restrict__ gpartial,
                                                                   unsigned nb_gram, const int* __restrict__ sweeps,
                                                                   const unsigned int* __restrict__ part, int* __restrict__ order,
                                                                   unsigned nb_order) {
    if (blockIdx.x < nb_order) { order_scatter_body(sweeps, ncols, part, order, blockIdx.x, nb_order); return; }
    const unsigned bid = blockIdx.x - nb_order;
    if (bid == 0)
        for (int f = threadIdx.x; f < k; f += blockDim.x) {
            float s = sums[f];
            if (norm_type == 1) s = sqrtf(s);
            d[f] = s + 1e-15f;
        }
    gram_partial_f32_k64_body<STEPS, true>(X, k, ncols, gpartial, bid, nb_gram, sums, norm_type);
}
// blocks [0, nb_cross): cross_partial; the rest: gram_finalize into G
template <class T>
__global__ __launch_bounds__(256) void tail_gramfin_cross_kernel(const T* __restrict__ gpartial, int nblk_g, int KP, int k, T eps, T l2,
                                                                  T* __restrict__ G, const T* __restrict__ W, const T* __restrict__ Bw,
                                                                  const T* __restrict__ d, int64_t total, double* __restrict__ cpartial,
                                                                  unsigned nb_cross) {
    __shared__ double sh[4];
    if (blockIdx.x < nb_cross) cross_partial_body<T>(W, Bw, d, k, total, cpartial, blockIdx.x, nb_cross, sh);
    else gram_finalize_body<T>(gpartial, nblk_g, KP, k, eps, l2, G, blockIdx.x - nb_cross);
}

}  // namespace rk
