// ops_rhs_win_f32_nv1.hip -- tile-loop kernels of the window rhs for float, rows of 256 bytes (own unit: parallel build)
#include "rhs_win_impl.hip.h"
void rcppml_rw_launch_f32_nv1(rcppml_hip_ctx* c, const rcppml_rhs_plan* pl, const float* F, float* Bout) {
    rw_launch::launch_clo<float, 1>(c, pl, F, Bout);
}
