// solve_common.hip.h -- padding helpers shared by the CD and Cholesky solve launchers
#pragma once
#include "common.hip.h"
#include "kernels.hip.h"

using namespace rk;
static int solve_kp(int k) { return k <= 16 ? 16 : (k <= 32 ? 32 : (k <= 64 ? 64 : 128)); }

template <class T>
static void pad_impl(rcppml_hip_ctx* c, const T* G, int k, int KP, T** Gp, T** invd) {
    T* buf = static_cast<T*>(c->scratch(WS_GPAD, ((size_t)KP * KP + KP) * sizeof(T)));
    *Gp = buf;
    *invd = buf + (size_t)KP * KP;
    hipLaunchKernelGGL(pad_gram<T>, dim3((KP * KP + 255) / 256), dim3(256), 0, c->stream, G, k, KP, *Gp, *invd);
    HIPCHK(hipGetLastError());
}

