"""BASELINE configs[3] at its FULL stated extent on one MI355X: pbmc3k-shaped 30 000 x 1 300 000, 3 %-dense CSC
(1.17e9 nonzeros: nnz * 4 bytes > 2^32, 96 nnz + 96 k (m + n) ~ 128 GB of the 288 GB), k = 128, fp32, CD.

Round-5 verdict, missing item 1: only one 162 500-column shard had ever run.  Here the WHOLE matrix goes
  (a) through the plugin's 73-pointer boundary on one device (rcppml_gpu_nmf_ex = rcppml_gpu_nmf_unified_float's loop + loss history),
  (b) through the same call with RCPPML_GPU_DEVICES=8 RCPPML_GPU_DEVICES_SHARE=1 (eight column shards, the multi-device loop of
      plugin_multi.hip with its collectives replaced by the shared-device sum: everything but the xGMI transfer), both W-solve forms,
  (c) through the device-level loop (rcppml_amd/als.py, A^T built on the device), op by op,
  (d) through the same boundary in fp64 (parity mode): fp32 loss within 1e-6 of the fp64 fit's at this size.
Checked: (a) = (b) = (c) to fp32 tolerances (the shards change the summation order of [G | B] only), W_T replicas bitwise
equal across the eight shards (asserted inside the plugin: status 0), rows of H and columns of W sum to 1, loss finite and
non-increasing, non-negativity, the linearity checksum of the SpMM-like kernel on both sides (planned and gather form), and 128 sampled
columns of each half-update against the oracle's fused RHS + CD from the device's own inputs.

The file sorts last (zz): it needs ~55 GB of device memory and a few minutes; the module-scoped matrix (14 GB of host arrays) is
built once.  Everything the reference does at this point of the path: nmf/fit_cpu.hpp:239-253 (the transpose memory pre-check is
the plugin's arena estimate), :486-893, :1729-1753."""
import os
import time

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

M, N, K, SHARDS = 30000, 1300000, 128, 8


class _Env:
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        for k, v in self.kw.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.fixture(scope="module")
def full():
    import torch
    from rcppml_amd import data
    if torch.cuda.get_device_properties(0).total_memory < 200 * 2 ** 30:
        pytest.skip("needs a 288 GB device")
    t0 = time.perf_counter()
    A = data.simulate_nmf_sparse_shards(M, N, K, 0.031, SHARDS, seed=11, device=torch.device("cuda", 0), round_f32=True)
    W0, H0 = data.init_factors(42, K, M, N, np.float32)
    W0, H0 = W0.astype(np.float64), H0.astype(np.float64)
    torch.cuda.empty_cache()
    print("c4full: generated %d x %d, nnz %d (%.3f %%) in %.1f s" % (M, N, A.nnz, 100.0 * A.nnz / (M * float(N)), time.perf_counter() - t0))
    assert abs(A.nnz / (M * float(N)) - 0.03) < 0.003 and A.nnz * 4 > 2 ** 32
    return A, W0, H0


def _plugin(A, W0, H0, ndev=1, w_solve=None, iters=2, verbose=0, precision=None):
    from rcppml_amd import _abi
    W, H = W0.copy(), H0.copy()
    with _Env(RCPPML_GPU_DEVICES=ndev if ndev > 1 else None, RCPPML_GPU_DEVICES_SHARE=1 if ndev > 1 else None,
              RCPPML_GPU_W_SOLVE=w_solve):
        t0 = time.perf_counter()
        res = _abi.nmf_unified(A.p, A.i, A.x, A.rows, A.cols, K, W, H, entry="ex", precision=_abi.F32 if precision is None else precision,
                               max_iter=iters, tol=0.0, solver_mode=0, sort_model=0, want_history=True, verbose=verbose)
        res["seconds"] = time.perf_counter() - t0
    assert res["status"] == 0, res.get("error")
    res["W_T"], res["H"] = W, H
    return res


def _props(res):
    hist = res["loss_history"]
    assert res["iter"] == len(hist) and np.all(np.isfinite(hist)) and hist[-1] <= hist[0] * (1 + 1e-5)
    W, H, d = res["W_T"], res["H"], res["d"]
    assert W.min() >= 0 and H.min() >= 0 and np.all(d > 0)
    assert np.allclose(W.sum(axis=0), 1.0, atol=3e-4)          # columns of W (L1 scaling, fp32 sums over 30 000 rows)
    assert np.allclose(H.sum(axis=0), 1.0, atol=3e-4)          # rows of H (sums over 1.3 M columns)


def test_full_extent_plugin_one_device_and_eight_shards(full, capfd):
    A, W0, H0 = full
    one = _plugin(A, W0, H0)
    _props(one)
    print("c4full plugin, one device: %.1f s for 2 iterations incl. upload / transpose / plans / download; loss %s" % (one["seconds"], one["loss_history"]))
    capfd.readouterr()
    rep = _plugin(A, W0, H0, ndev=SHARDS, verbose=2)
    err = capfd.readouterr().err
    shard_lines = [ln for ln in err.splitlines() if "shard" in ln and "ms/iteration" in ln]
    assert len(shard_lines) == SHARDS and "W_T replicas bitwise equal" in err, err[-2000:]
    print("\n".join(ln for ln in err.splitlines() if ln.startswith("[rcppml_gpu x")))
    _props(rep)
    blk = _plugin(A, W0, H0, ndev=SHARDS, w_solve="block")
    _props(blk)
    print("c4full plugin, 8 shards on one device: replicated %.1f s, block %.1f s" % (rep["seconds"], blk["seconds"]))
    for other, what in ((rep, "replicated"), (blk, "block")):
        assert other["iter"] == one["iter"]
        dl = np.abs(other["loss_history"] - one["loss_history"]).max() / one["loss_history"].max()
        dw = np.abs(other["W_T"] - one["W_T"]).max() / one["W_T"].max()
        dh = np.abs(other["H"] - one["H"]).max() / one["H"].max()
        dd = np.abs(other["d"] - one["d"]).max() / one["d"].max()
        print("c4full 8 shards (%s) vs one device: loss %.2e  W %.2e  H %.2e  d %.2e" % (what, dl, dw, dh, dd))
        # fp32: the shards change the summation order of the all-reduced [G | B | row sums]; same bars as the fp32 whole-fit tests
        assert dl < 1e-5 and dw < 5e-3 and dh < 5e-3 and dd < 1e-4
    # the two W-solve forms start from the same all-reduced (G, B): the blocks are solved by the same kernels
    assert np.abs(blk["W_T"] - rep["W_T"]).max() / rep["W_T"].max() < 1e-4
    # parity mode at full extent (1024-byte rows in the window kernel, eight row tiles per wave in the fp64 MFMA solve): the fp32 fit's
    # loss stays within the north star's 1e-6 of the fp64 fit's, the factors within the fp32 bar
    from rcppml_amd import _abi
    f64 = _plugin(A, W0, H0, precision=_abi.F64)
    _props(f64)
    dl = np.abs(f64["loss_history"] - one["loss_history"]).max() / f64["loss_history"].max()
    dl_last = abs(f64["loss_history"][-1] - one["loss_history"][-1]) / f64["loss_history"][-1]
    dw = np.abs(f64["W_T"] - one["W_T"]).max() / f64["W_T"].max()
    dh = np.abs(f64["H"] - one["H"]).max() / f64["H"].max()
    print("c4full fp32 vs fp64 (one device, %.1f s): loss history %.2e, last loss %.2e, W %.2e, H %.2e" % (f64["seconds"], dl, dl_last, dw, dh))
    assert dl < 1e-6 and dl_last < 1e-6 and dw < 5e-3 and dh < 5e-3
    full_one = one
    test_full_extent_plugin_one_device_and_eight_shards.one = full_one


def test_full_extent_device_loop_properties_and_sampled_columns(full):
    import torch
    from rcppml_amd import als
    A, W0, H0 = full
    ops = als.HipOps(0, "f32")
    cfg = als.AlsConfig(k=K, max_iter=3, tol=0.0)
    t0 = time.perf_counter()
    st = als.ShardedALS(ops, als.Comm(None), A, None, W0, H0, cfg)          # A^T on the device
    ops.sync()
    print("c4full device loop: upload + device transpose + plans %.1f s" % (time.perf_counter() - t0))
    assert st.A.get("plans", {}).get(K) is not None and st.At.get("plans", {}).get(K) is not None      # the window kernel runs
    for side, csc in (("H", st.A), ("W", st.At)):
        print("c4full plan", side, csc["plans"][K].info())
    # the device transpose at this size: row pointers = row counts, column indices ascending inside every row, values carried along
    Ai, Ax = st.A["i"], st.A["x"]
    cnt = torch.bincount(Ai.long(), minlength=M)
    tp = st.At["p"].long()
    assert int(tp[-1]) == A.nnz and torch.equal(tp[1:] - tp[:-1], cnt)
    ti = st.At["i"]
    asc = ti[1:] > ti[:-1]
    asc[(tp[1:-1] - 1).clamp(min=0, max=A.nnz - 2)] = True          # row boundaries
    assert bool(asc.all())
    assert abs(float(st.At["x"].double().sum()) - float(Ax.double().sum())) <= 1e-9 * float(Ax.double().sum())
    del cnt, asc
    losses = []
    for it in range(2):
        t1 = time.perf_counter()
        losses.append(float(st.step()[0].item()))
        print("c4full device loop: iteration %d %.3f s (first call of a kernel includes its code upload), loss %.9g" % (it, time.perf_counter() - t1, losses[-1]))
        rs = ops.row_norms(st.H, 0).cpu().numpy()
        assert np.allclose(rs, 1.0, atol=3e-4)
    assert np.all(np.isfinite(losses)) and losses[1] <= losses[0] * (1 + 1e-5)
    one = getattr(test_full_extent_plugin_one_device_and_eight_shards, "one", None)
    if one is not None:          # the plugin's loop issues the same device ops on the same inputs
        assert np.abs(np.array(losses) - one["loss_history"]).max() / one["loss_history"].max() < 1e-6
    W_T, d, H = st.factors()
    assert W_T.min() >= 0 and H.min() >= 0 and np.all(d > 0)
    assert np.allclose(W_T.sum(axis=0), 1.0, atol=3e-4)
    # linearity checksums (fp32 kernels, fp64 reference on the device): sum_j B(:, j) = F^T (A 1), planned and gather form
    rowsum = torch.zeros(M, dtype=torch.float64, device=Ai.device).index_add_(0, Ai.long(), Ax.double())
    colsum = torch.zeros(N, dtype=torch.float64, device=Ai.device).index_add_(0, st.At["i"].long(), st.At["x"].double())
    for csc, F, w in ((st.A, st.W_T, rowsum), (st.At, st.H, colsum)):
        ref = (F.double() * w[:, None]).sum(dim=0).cpu().numpy()
        B = ops.rhs(csc, F)
        s = B.double().sum(dim=0).cpu().numpy()
        assert np.abs(s - ref).max() / np.abs(s).max() < 2e-4
        B2 = ops.empty(tuple(B.shape))
        ops.ctx.rhs(ops.dt, csc["p"], csc["i"], csc["x"], csc["cols"], F, K, B2)
        assert float((B - B2).abs().max() / B2.abs().max()) < 5e-5          # planned == gather kernel up to summation order
        del B, B2
    del rowsum, colsum
    # sampled columns of both half-updates vs the oracle (third iteration, driven op by op)
    sums, dd = ops.empty((K,)), ops.empty((K,))
    for side in ("H", "W"):
        F, X, csc = (st.W_T, st.H, st.A) if side == "H" else (st.H, st.W_T, st.At)
        G = ops.gram(F, 1e-15, 0.0)
        B = ops.rhs(csc, F)
        cols = np.sort(np.random.default_rng(5 + (side == "W")).choice(csc["cols"], size=128, replace=False))
        ct = torch.from_numpy(cols).to(X.device)
        X_prev = X[ct].cpu().numpy()
        F_host, G_host = F.cpu().numpy(), G.cpu().numpy()
        ops.solve(G, B, X, cfg, side, True)
        X_new = X[ct].cpu().numpy()
        p_host = csc["p"].cpu().numpy().astype(np.int64)
        ii = np.concatenate([csc["i"][p_host[c]:p_host[c + 1]].cpu().numpy() for c in cols])
        xx = np.concatenate([csc["x"][p_host[c]:p_host[c + 1]].cpu().numpy() for c in cols]).astype(np.float64)
        pick = O.Csc((csc["rows"], len(cols)), np.concatenate([[0], np.cumsum(np.diff(p_host)[cols])]).astype(np.int32), ii, xx)
        ref = O.fused_cd(pick, F_host, G_host, X_prev, maxit=100, tol=1e-8, warm=True)
        err = np.abs(X_new - ref).max() / np.abs(ref).max()
        print("c4full sampled columns, %s side: worst relative deviation %.3e" % (side, err))
        assert err < 2e-2, (side, err)
        assert float(X.min()) >= 0
        ops.row_norms(X, 0, out=sums)
        ops.apply_scaling(X, sums, 0, dd)
