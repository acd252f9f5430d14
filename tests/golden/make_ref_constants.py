"""Generates tests/golden/ref_constants.npz: the numerical constants and algorithm defaults of the reference's
inst/include/FactorNet/core/constants.hpp, read from the header ITSELF compiled in place (oracle/_ref/libref_loss.so, built by
`make -C oracle ref` from /root/reference; no source is copied).  Run in the build container; the GPU box uses the committed file."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = ["tiny_num_f64", "tiny_num_f32", "kl_epsilon", "CD_TOL", "CD_MAXIT", "CD_ABS_TOL", "NMF_TOL", "NMF_MAXIT", "NMF_PATIENCE",
         "DEFAULT_L1", "DEFAULT_L2", "DEFAULT_L21", "DEFAULT_GRAPH_LAMBDA", "DEFAULT_HUBER_DELTA"]


def read_live():
    L = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref", "libref_loss.so"))
    buf = (C.c_double * 32)()
    n = L.ref_constants(buf, 32)
    assert n == len(NAMES), n
    return {k: float(buf[i]) for i, k in enumerate(NAMES)}


if __name__ == "__main__":
    vals = read_live()
    np.savez(os.path.join(HERE, "ref_constants.npz"), **{k: np.float64(v) for k, v in vals.items()})
    print(vals)
