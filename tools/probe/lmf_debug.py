"""Element-wise comparison of the lane = column CD kernel with the 32-column MFMA kernel at a fixed sweep count."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rcppml_amd import _abi
ctx = _abi.Context(0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for k, n, lg in ((7, 70000, 1), (7, 300, 1), (32, 300, 1), (20, 300, 1), (64, 300, 1), (7, 300, 4)):
    rs = np.random.default_rng(100 * lg + k)
    Fm = rs.uniform(size=(4 * k + 5, k))
    G = (Fm.T @ Fm).astype(np.float32)
    G[np.diag_indices(k)] += np.float32(1e-15)
    B = (rs.standard_normal((n, k)) * 3 + 1).astype(np.float32)
    X0 = rs.uniform(size=(n, k)).astype(np.float32)
    ctx.set_option(_abi.OPT_CD_LMF_LANE_GROUPS, lg)
    ctx.set_option(_abi.OPT_CD_LMF_WAVES_PER_SIMD, 1)
    for kw in (dict(warm=1), dict(warm=0), dict(zero_init=1)):
        for maxit in (1, 2, 6):
            outs = []
            for var in (_abi.CD_MFMA, _abi.CD_LMF, _abi.CD_GROUP):
                dX = dev(X0.copy())
                ctx.solve_cd(_abi.F32, dev(G), dev(B), dX, k, n, maxit=maxit, tol=0.0, variant=var, **kw)
                outs.append(dX.cpu().numpy())
            d = outs[0] != outs[1]
            d2 = outs[2] != outs[1]
            print("k=%d n=%d lg=%d %s maxit=%d: mfma32 vs lmf differing elements %d (cols %d) max|d| %.3e | group vs lmf %d max|d| %.3e | group vs mfma32 %d"
                  % (k, n, lg, kw, maxit, int(d.sum()), int(d.any(axis=1).sum()), np.abs(outs[0] - outs[1]).max(), int(d2.sum()),
                     np.abs(outs[2] - outs[1]).max(), int((outs[2] != outs[0]).sum())))
            if d.any() and n <= 300 and maxit == 1:
                jj, rr = np.nonzero(d)
                print("   first diffs (col,row,mfma32,lmf,group):", [(int(a), int(b), float(outs[0][a, b]), float(outs[1][a, b]), float(outs[2][a, b])) for a, b in list(zip(jj, rr))[:6]])
