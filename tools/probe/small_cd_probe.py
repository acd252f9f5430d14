import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from rcppml_amd import _abi
ctx = _abi.Context(0)
rs = np.random.default_rng(1)
for n, k in ((610, 32), (1183, 10), (1500, 64), (3867, 32)):
    F = rs.uniform(size=(4 * k + 5, k)); G = (F.T @ F).astype(np.float32); G[np.diag_indices(k)] += 1e-15
    B = (rs.standard_normal((n, k)) * 3 + 1).astype(np.float32)
    X0 = rs.uniform(size=(n, k)).astype(np.float32)
    dG, dB = torch.from_numpy(G).cuda(), torch.from_numpy(B).cuda()
    outs = {}
    for name, var in (("auto", _abi.CD_AUTO), ("mfma32", _abi.CD_MFMA), ("mfma16", _abi.CD_MFMA16)):
        sw = torch.zeros(n, dtype=torch.int32, device="cuda")
        ts = []
        for rep in range(6):
            dX = torch.from_numpy(X0.copy()).cuda()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ctx.solve_cd(_abi.F32, dG, dB, dX, k, n, warm=1, maxit=100, tol=1e-8, variant=var, sweeps_out=sw)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        outs[name] = dX.cpu().numpy()
        s = sw.cpu().numpy()
        print("n %5d k %2d %-7s %.4f ms  sweeps mean %.1f max %d" % (n, k, name, float(np.median(ts[1:])), s.mean(), s.max()))
    print("   max |auto - mfma32| %.3e" % np.abs(outs["auto"] - outs["mfma32"]).max())
