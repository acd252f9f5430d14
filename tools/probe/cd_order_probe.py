"""Which column order makes the MFMA CD kernels fastest?  Captures the inputs of the H- and W-side solves at a steady-state
iteration of C2 and replays rcppml_hip_solve_cd with different orders (results per column do not depend on the order)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from rcppml_amd import als, data
m, n, k = 20000, 100000, 64
A, _, _ = data.simulate_nmf_sparse(m, n, k, 0.0115, seed=123, device=torch.device("cuda", 0))
At = A.transpose()
W0, H0 = data.init_factors(42, k, m, n, np.float32)
ops = als.HipOps(0, "f32")
cap = {}
orig = ops.solve
def spy(G, B, X, cfg, side, warm, tag="solve"):
    if cap.get("on"):
        cap[side] = dict(G=G.clone(), B=B.clone(), X=X.clone(), warm=warm, cfg=cfg, prev=ops._order[side]["sweeps"].clone())
    orig(G, B, X, cfg, side, warm, tag)
    if cap.get("on"):
        cap[side]["cur"] = ops._order[side]["sweeps"].clone()
ops.solve = spy
st = als.ShardedALS(ops, als.Comm(None), A, At, W0, H0, als.AlsConfig(k=k, max_iter=30, tol=0.0))
for it in range(10):
    cap["on"] = it == 9
    st.step()
def timeit(c, order):
    X = c["X"].clone(); sw = torch.zeros(X.shape[0], dtype=torch.int32, device="cuda")
    def run():
        X.copy_(c["X"])
        ops.ctx.solve_cd(ops.dt, c["G"], c["B"], X, k, X.shape[0], l1_pre=0.0, warm=int(c["warm"]), zero_init=0, nonneg=1,
                         maxit=100, tol=1e-8, ub_post=0.0, variant=0, sweeps_out=sw, col_order=order)
    run(); run(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(7):
        X.copy_(c["X"]); s.record()
        ops.ctx.solve_cd(ops.dt, c["G"], c["B"], X, k, X.shape[0], l1_pre=0.0, warm=int(c["warm"]), zero_init=0, nonneg=1,
                         maxit=100, tol=1e-8, ub_post=0.0, variant=0, sweeps_out=sw, col_order=order)
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return float(np.median(ts)) * 1e3
def blocks_perm(order, per_block, fn):
    nb = (len(order) + per_block - 1) // per_block
    pad = nb * per_block - len(order)
    o = np.concatenate([order, np.full(pad, -1, order.dtype)]).reshape(nb, per_block)
    o = o[fn(nb)]
    o = o.reshape(-1)
    return o[o >= 0]
def serp(period):
    def f(nb):
        idx = np.arange(nb); g = idx // period; pos = idx % period
        base = g * period
        size = np.minimum(period, nb - base)
        return np.where(g % 2 == 1, base + size - 1 - pos, idx)
    return f
for side, tile in (("H", 32), ("W", 16)):
    c = cap[side]
    prev, cur = c["prev"].cpu().numpy().astype(np.int64), c["cur"].cpu().numpy().astype(np.int64)
    per_block = 4 * tile
    o_prev = np.argsort(-prev, kind="stable").astype(np.int32)
    o_cur = np.argsort(-cur, kind="stable").astype(np.int32)
    variants = {"natural": None, "sorted(prev)": o_prev, "sorted(cur) = oracle": o_cur,
                "sorted(prev) ascending": o_prev[::-1].copy()}
    for p in (8, 16, 24, 32, 48, 64, 128, 256, 512):
        variants["serpentine/%d blocks (prev)" % p] = blocks_perm(o_prev, per_block, serp(p))
    for p in (32, 64, 128, 256):
        variants["serpentine/%d tiles (prev)" % p] = blocks_perm(o_prev, tile, serp(p))
    for p in (16, 32, 64):
        variants["serpentine/%d x 128 slots (prev)" % p] = blocks_perm(o_prev, 128, serp(p))
    variants["serpentine/256 (oracle)"] = blocks_perm(o_cur, per_block, serp(256))
    rs = np.random.default_rng(0)
    variants["sorted(prev), blocks shuffled"] = blocks_perm(o_prev, per_block, lambda nb: rs.permutation(nb))
    print("side", side, "columns", len(cur), "mean sweeps %.1f max %d" % (cur.mean(), cur.max()))
    for name, o in variants.items():
        od = None if o is None else torch.from_numpy(np.ascontiguousarray(o.astype(np.int32))).cuda()
        print("   %-36s %.1f us" % (name, timeit(c, od)))
