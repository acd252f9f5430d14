"""Idle-slot fraction of the MFMA CD tiles on C2: a wave sweeps until its slowest column has converged."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from rcppml_amd import als, data
m, n, k = 20000, 100000, 64
A, _, _ = data.simulate_nmf_sparse(m, n, k, 0.0115, seed=123, device=torch.device("cuda", 0))
At = A.transpose()
W0, H0 = data.init_factors(42, k, m, n, np.float32)
ops = als.HipOps(0, "f32")
st = als.ShardedALS(ops, als.Comm(None), A, At, W0, H0, als.AlsConfig(k=k, max_iter=30, tol=0.0))
for it in range(12):
    st.step()
    if it in (3, 7, 11):
        for side, tile in (("H", 32), ("W", 16)):
            o = ops._order[side]
            sw = o["sweeps"].cpu().numpy().astype(np.int64)
            order = o["order"].cpu().numpy() if sw.shape[0] >= als.ORDER_MIN_COLUMNS else np.arange(sw.shape[0])
            s = sw[order]
            pad = (-len(s)) % tile
            t = np.concatenate([s, np.zeros(pad, np.int64)]).reshape(-1, tile)
            busy = t.sum(); slots = (t.max(axis=1) * tile).sum()
            # what a perfect sort by THIS iteration's sweeps would give
            ss = np.sort(sw); t2 = np.concatenate([ss, np.zeros(pad, np.int64)]).reshape(-1, tile)
            slots2 = (t2.max(axis=1) * tile).sum()
            print("iter %2d side %s: mean sweeps %.1f  max %d  idle fraction %.3f  (oracle-sorted %.3f)  tiles %d  hist10/50/90 %s" % (
                it, side, sw.mean(), sw.max(), 1 - busy / slots, 1 - busy / slots2, t.shape[0], np.percentile(sw, [10, 50, 90])))
