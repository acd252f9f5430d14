"""N > 1 path with the real HIP kernels: two ranks share cuda:0 under `gloo` (RCCL refuses two ranks on one device; the
GPU box has one GPU), so the column-sharded loop of rcppml_amd/als.py -- row-block W solve on sliced device buffers,
fused [G | B] all-reduce, W_T all-gather -- runs end to end through the C-ABI and is checked against the
single-process oracle fit.  The RCCL transport itself is exercised only by the driver's multi-GPU bench."""
import os
import socket

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, dtype, k, cfg_kw, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from rcppml_amd import als, data
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        A, _, _ = data.simulate_nmf_sparse(203, 340, 5, 0.2, seed=23)
        bounds = als.partition_columns_by_nnz(A.p, world)
        c0, c1 = bounds[rank], bounds[rank + 1]
        A_loc = A.col_slice(c0, c1)
        W0, H0 = data.init_factors(9, k, A.rows, A.cols)
        cfg = als.AlsConfig(k=k, **cfg_kw)
        ops = als.HipOps(0, dtype)
        st = als.ShardedALS(ops, als.Comm(dist), A_loc, A_loc.transpose(), W0, H0[c0:c1], cfg)
        res = st.fit()
        W_T, d, H = st.factors()
        q.put((rank, c0, c1, res, W_T, d, H))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dtype,k,tol", [("f64", 7, 1e-6), ("f64", 16, 1e-6), ("f32", 4, 2e-3)])
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_als_hip_two_ranks_one_gpu(dtype, k, tol, world):
    import torch.multiprocessing as mp
    from rcppml_amd import data
    cfg_kw = dict(max_iter=6, tol=0.0, L1_H=2e-6, L2_W=1e-3)       # penalties sized to the data (values ~1e-4): no dead factors
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, dtype, k, cfg_kw, q)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    import time
    outs, t0 = [], time.time()
    while len(outs) < world:
        try:
            outs.append(q.get(timeout=2))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() - t0 > 240:
                for p in procs:
                    if p.is_alive():
                        p.terminate()
                pytest.fail("worker failed or timed out: exit codes %s" % [p.exitcode for p in procs])
    outs.sort(key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    A, _, _ = data.simulate_nmf_sparse(203, 340, 5, 0.2, seed=23)
    Ao = O.Csc(A.shape, A.p, A.i, A.x)
    W0, H0 = data.init_factors(9, k, A.rows, A.cols)
    nd = np.float64 if dtype == "f64" else np.float32
    ref = O.nmf_fit(Ao, W0, H0, nd, max_iter=6, tol=0.0, L1=(0.0, 2e-6), L2=(1e-3, 0.0), sort_model=False)
    assert ref.d.min() > 1e-3 and (ref.H > 0).mean() > 0.2            # a live fit, not the all-zero fixed point
    for o in outs:   # W_T and d are replicated: bitwise identical on every rank (identical reduced inputs, gathered blocks)
        assert np.array_equal(o[4], outs[0][4]) and np.array_equal(o[5], outs[0][5])
    H_full = np.concatenate([o[6] for o in outs], axis=0)
    hist = np.array(outs[0][3]["loss_history"])
    # fp64: loss to 1e-9; factors to 1e-6 (summation order of the reduced Gram / RHS moves the cd_tol early exit)
    assert np.abs(hist - ref.loss_history).max() / ref.loss_history.max() < (1e-9 if dtype == "f64" else tol)
    assert np.abs(outs[0][4] - ref.W_T).max() < tol
    assert np.abs(outs[0][5] - ref.d).max() / ref.d.max() < tol
    assert np.abs(H_full - ref.H).max() < tol
