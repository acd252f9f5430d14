import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from rcppml_amd import als, data
m, n, k = 20000, 100000, 64
A, _, _ = data.simulate_nmf_sparse(m, n, k, 0.0115, seed=123, device=torch.device("cuda", 0))
At = A.transpose()
W0, H0 = data.init_factors(42, k, m, n, np.float32)
ops = als.HipOps(0, "f32")
cfg = als.AlsConfig(k=k, max_iter=4, tol=0.0)
st = als.ShardedALS(ops, als.Comm(None), A, At, W0, H0, cfg)
print('H plan', st.A['plans'][k].info()); print('W plan', st.At['plans'][k].info())
for name, csc in (('H', st.A), ('W', st.At)):
    p2 = ops.ctx.rhs_plan(ops.dt, csc['p'], csc['i'], csc['x'], csc['cols'], csc['rows'], k)
    print(name, 'rebuilt', p2.info())
    F = st.W_T if name == 'H' else st.H
    B1 = ops.empty((csc['cols'], k)); B2 = ops.empty((csc['cols'], k))
    ops.ctx.rhs_planned(p2, F, B1); ops.ctx.rhs(ops.dt, csc['p'], csc['i'], csc['x'], csc['cols'], F, k, B2)
    print(name, 'rebuilt plan rel diff', ((B1 - B2).abs().max() / B2.abs().max()).item(), 'p dtype', csc['p'].dtype, csc['i'].dtype, csc['x'].dtype, 'F', F.dtype, F.shape, F.is_contiguous(), F.data_ptr() % 256)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for it in range(3):
    for name, csc, F in (("H", st.A, st.W_T), ("W", st.At, st.H)):
        plan = csc["plans"][k]
        B1 = ops.empty((csc["cols"], k)); B2 = ops.empty((csc["cols"], k))
        t1 = t(lambda: ops.ctx.rhs_planned(plan, F, B1))
        t2 = t(lambda: ops.ctx.rhs(ops.dt, csc["p"], csc["i"], csc["x"], csc["cols"], F, k, B2))
        d = (B1 - B2).abs().max().item() / B2.abs().max().item()
        nz = (F == 0).float().mean().item()
        print("iter", it, name, "planned %.3f ms gather %.3f ms  rel diff %.2e  F zeros %.3f  F absmax %.3e min nonzero %.3e" % (t1, t2, d, nz, F.abs().max().item(), F[F > 0].min().item()))
    st.step()
