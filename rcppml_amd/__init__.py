"""rcppml_amd -- MI355X (gfx950) backend for RcppML's alternating-NNLS NMF hot path.

Layout (only what the path needs):
  csrc/      hand-written HIP kernels + the C-ABI (include/rcppml_gpu.h) -> lib/RcppML_gpu.so
  _abi.py    ctypes binding of that C-ABI (no CPU fallback: fails loudly if the library is missing)
  nmf.py     host-side mirror of the reference R surface: nmf() / nnls() / predict() / evaluate()
  als.py     one-process-per-GPU column-sharded ALS loop over torch.distributed (RCCL): Comm, ShardedALS, HipOps
  data.py    synthetic inputs (restatement of R/simulateNMF.R) and CSC helpers
"""
__version__ = "0.1.0"
