"""Calls the per-phase profile entry (rcppml_gpu_nmf_profile_double, reference src/gpu_bridge_utils.cu:48) on a synthetic
matrix of BASELINE configs[1]'s shape (20 000 x 100 000, 1 %, k = 64) and prints its eleven phase times as one JSON line."""
import argparse
import json

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=20000)
    ap.add_argument("--n", type=int, default=100000)
    ap.add_argument("--k", type=int, default=64)
    ap.add_argument("--density", type=float, default=0.0115)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--cd-maxit", type=int, default=10)
    args = ap.parse_args()
    import torch
    from rcppml_amd import _abi, data
    A, _, _ = data.simulate_nmf_sparse(args.m, args.n, args.k, args.density, seed=123, device=torch.device("cuda", 0))
    res = _abi.nmf_profile_double(A.p.astype(np.int32), A.i.astype(np.int32), A.x.astype(np.float64), args.m, args.n, args.k,
                                  max_iter=args.iters, tol=0.0, cd_maxit=args.cd_maxit, seed=42)
    print(json.dumps(dict(entry="rcppml_gpu_nmf_profile_double", dtype="f64", m=args.m, n=args.n, k=args.k, nnz=int(A.x.shape[0]),
                          cd_maxit=args.cd_maxit, iters=res["iters"], per_iter_ms={p: round(v, 4) for p, v in res["per_iter_ms"].items()})))


if __name__ == "__main__":
    main()
