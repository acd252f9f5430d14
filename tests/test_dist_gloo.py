"""N > 1 path on CPU: the column-sharded ALS loop (rcppml_amd/als.py) under torch.distributed `gloo` with
world_size 2 and 3, compute supplied by the oracle-backed ops of tests/oracle_ops.py.  Checks the sharded run
reproduces the single-process oracle fit (nmf_fit<CPU> restatement) -- same loss trajectory and factors up to
summation-order rounding -- and that nnz-balanced partitioning covers every column exactly once."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as O
from rcppml_amd import als, data
from tests.oracle_ops import OracleOps


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _matrix(kind):
    """"even": 90 x 140, 20 % dense.  "skewed": the same rows, 61 columns of which a handful hold most of the nonzeros (dense blocks
    next to nearly empty columns: nnz-balanced shards of very different widths).  "tiny": 5 columns for 8 ranks -- empty shards."""
    if kind == "even":
        return data.simulate_nmf_sparse(90, 140, 5, 0.2, seed=17)[0]
    rs = np.random.default_rng(5)
    cols = 61 if kind == "skewed" else 5
    dens = np.where(rs.uniform(size=cols) < 0.15, 0.95, 0.02) if kind == "skewed" else np.full(cols, 0.5)
    mask = rs.uniform(size=(90, cols)) < dens[None, :]
    mask[0, :] = True                                     # no empty column
    D = np.where(mask, rs.uniform(0.1, 1.0, size=(90, cols)), 0.0)
    import scipy.sparse as sp
    M = sp.csc_matrix(D)
    M.sort_indices()
    return data.CSC((90, cols), M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data.astype(np.float64))


def _worker(rank, world, port, cfg_kw, q, kind="even"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        A = _matrix(kind)
        k = 6
        bounds = als.partition_columns_by_nnz(A.p, world)
        c0, c1 = bounds[rank], bounds[rank + 1]
        A_loc = A.col_slice(c0, c1)
        W0, H0 = data.init_factors(9, k, A.rows, A.cols)
        cfg = als.AlsConfig(k=k, **cfg_kw)
        st = als.ShardedALS(OracleOps("f64"), als.Comm(dist), A_loc, A_loc.transpose(), W0, H0[c0:c1], cfg)
        res = st.fit()
        W_T, d, H = st.factors()
        q.put((rank, c0, c1, res, W_T, d, H))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,kind,cfg_kw", [
    (2, "even", dict(max_iter=8, tol=0.0, L1_H=2e-6, L2_W=1e-3)),
    (3, "even", dict(max_iter=8, tol=0.0, L1_H=2e-6, L2_W=1e-3)),
    (2, "even", dict(max_iter=6, tol=0.0, solver_mode=1, norm_type=1)),
    (3, "even", dict(max_iter=6, tol=0.0, solver_mode=1, norm_type=1)),
    (3, "even", dict(max_iter=6, tol=0.0, w_solve="replicated")),
    # the 8-rank leg of BASELINE configs[3] in miniature: nnz-balanced shards of uneven width, and more ranks than columns
    (8, "skewed", dict(max_iter=6, tol=0.0, L1_H=2e-6)),
    (8, "skewed", dict(max_iter=5, tol=0.0, w_solve="replicated", solver_mode=1)),
    (8, "tiny", dict(max_iter=5, tol=0.0)),
])
def test_sharded_als_matches_single_process(world, kind, cfg_kw):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, cfg_kw, q, kind)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference on the full matrix (unsorted, to compare factor-by-factor)
    A = _matrix(kind)
    if kind != "even":          # the shards really are uneven / empty
        widths = [o[2] - o[1] for o in outs]
        assert (min(widths) == 0) if kind == "tiny" else (max(widths) >= 2 * max(1, min(widths)))
    Ao = O.Csc(A.shape, A.p, A.i, A.x)
    W0, H0 = data.init_factors(9, 6, A.rows, A.cols)
    ref = O.nmf_fit(Ao, W0, H0, max_iter=cfg_kw["max_iter"], tol=0.0, L1=(cfg_kw.get("L1_W", 0.0), cfg_kw.get("L1_H", 0.0)),
                    L2=(cfg_kw.get("L2_W", 0.0), cfg_kw.get("L2_H", 0.0)), solver_mode=cfg_kw.get("solver_mode", 0),
                    norm_type=cfg_kw.get("norm_type", 0), sort_model=False)
    if kind == "even":
        assert ref.d.min() > 1e-3 and (ref.H > 0).mean() > 0.2        # a live fit (penalties sized to data ~1e-4)
    H_full = np.concatenate([o[6] for o in outs], axis=0)
    assert outs[0][1] == 0 and outs[-1][2] == A.cols and all(outs[i][2] == outs[i + 1][1] for i in range(world - 1))
    for o in outs:   # W_T, d and the loss are replicated: identical on every rank
        assert np.array_equal(o[4], outs[0][4]) and np.array_equal(o[5], outs[0][5])
        assert o[3]["loss_history"] == outs[0][3]["loss_history"]
    hist = np.array(outs[0][3]["loss_history"])
    assert np.abs(hist - ref.loss_history).max() / ref.loss_history.max() < 1e-9
    assert np.abs(outs[0][4] - ref.W_T).max() < 1e-9
    assert np.abs(outs[0][5] - ref.d).max() / ref.d.max() < 1e-9
    assert np.abs(H_full - ref.H).max() < 1e-9


def test_partition_by_nnz():
    A, _, _ = data.simulate_nmf_sparse(300, 1000, 4, 0.05, seed=2)
    for world in (1, 2, 4, 8):
        b = als.partition_columns_by_nnz(A.p, world)
        assert b[0] == 0 and b[-1] == A.cols and len(b) == world + 1 and all(b[i] <= b[i + 1] for i in range(world))
        per = [int(A.p[b[i + 1]] - A.p[b[i]]) for i in range(world)]
        assert max(per) - min(per) <= 2 * int(np.diff(A.p).max())
    # degenerate: more ranks than columns
    b = als.partition_columns_by_nnz(np.array([0, 3, 5]), 4)
    assert b[0] == 0 and b[-1] == 2 and all(b[i] <= b[i + 1] for i in range(4))


def _one_rank_worker(port, cfg_kw, force, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        A = _matrix("even")
        k = 6
        W0, H0 = data.init_factors(9, k, A.rows, A.cols)
        comm = als.Comm(dist, force=force)
        st = als.ShardedALS(OracleOps("f64"), comm, A, A.transpose(), W0, H0, als.AlsConfig(k=k, **cfg_kw))
        res = st.fit()
        W_T, d, H = st.factors()
        q.put((comm.sharded, st.rows_per, res, W_T, d, H))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("w_solve", ["block", "replicated"])
def test_forced_one_rank_group_takes_the_sharded_branch(w_solve):
    """Comm(force=True) with a process group of ONE rank (what bench.py RCPPML_BENCH_FORCE_DIST=1 builds on the GPU box with the nccl
    backend): the loop runs its sharded branch -- partial Gram / right-hand side / row sums into the fused buffer, the all-reduce and
    the all-gather really issued, D^-1 applied after the sum -- and lands on the unforced fit (scaling before the Gram) to rounding;
    without `force` a one-rank group stays on the single-rank branch."""
    ctx = mp.get_context("spawn")
    out = {}
    for force in (True, False):
        q = ctx.Queue()
        p = ctx.Process(target=_one_rank_worker, args=(_free_port(), dict(max_iter=6, tol=0.0, L1_H=2e-6, w_solve=w_solve), force, q))
        p.start()
        out[force] = q.get(timeout=240)
        p.join(timeout=60)
        assert p.exitcode == 0
    sharded, rows_per, res, W_T, d, H = out[True]
    sharded0, rows_per0, res0, W_T0, d0, H0 = out[False]
    assert sharded and not sharded0
    assert rows_per % 4 == 0 and rows_per >= 90 and rows_per0 == 90            # forced: the padded block length of the all-gather
    assert res["iter"] == res0["iter"]
    assert abs(res["loss"] - res0["loss"]) <= 1e-9 * abs(res0["loss"])
    assert np.abs(W_T - W_T0).max() < 1e-8 and np.abs(H - H0).max() < 1e-8 and np.abs(d - d0).max() <= 1e-8 * np.abs(d0).max()
