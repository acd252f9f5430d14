"""How well do earlier iterations' sweep counts predict this iteration's (the sort key of the MFMA CD tiles)?"""
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
hist = {"H": [], "W": []}
for it in range(14):
    st.step()
    for side in ("H", "W"):
        hist[side].append(ops._order[side]["sweeps"].cpu().numpy().astype(np.int64).copy())
def idle(sw, key, tile):
    order = np.argsort(-key, kind="stable")
    s = sw[order]; pad = (-len(s)) % tile
    t = np.concatenate([s, np.zeros(pad, np.int64)]).reshape(-1, tile)
    return 1 - t.sum() / (t.max(axis=1) * tile).sum()
for side, tile in (("H", 32), ("W", 16)):
    h = hist[side]
    for it in (9, 13):
        cur, p1, p2, p3 = h[it], h[it - 1], h[it - 2], h[it - 3]
        print(side, "iter", it, "corr(prev,cur) %.3f" % np.corrcoef(p1, cur)[0, 1],
              "idle: natural %.3f prev %.3f mean2 %.3f mean3 %.3f max2 %.3f 2*prev-prev2 %.3f oracle %.3f" % (
                  idle(cur, -np.arange(len(cur)), tile), idle(cur, p1, tile), idle(cur, p1 + p2, tile), idle(cur, p1 + p2 + p3, tile),
                  idle(cur, np.maximum(p1, p2), tile), idle(cur, 2 * p1 - p2, tile), idle(cur, cur, tile)))
