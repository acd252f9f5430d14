"""GPU tests of the multi-device ALS loop behind the plugin boundary (rcppml_amd/csrc/plugin_multi.hip):
RCPPML_GPU_DEVICES=n shards the columns of the 73-pointer call's matrix over n devices of the process.  The GPU box has one
GPU and RCCL refuses two ranks on one device, so the shards are mapped onto cuda:0 (RCPPML_GPU_DEVICES_SHARE=1: same
partitioning, same kernels, same loop; the two all-reduces become a local sum kernel).  With distinct devices the only
difference is who adds the partial buffers.  Checked: the sharded fit equals the one-device fit and the CPU oracle
(fp64: loss 1e-9 / factors 1e-8 -- the shards change only the summation order of [G | B] and of the k row sums), same
iteration count and convergence flag; W is solved redundantly per shard from identical inputs, so the run is also
deterministic bit for bit; configurations the sharded loop does not cover fall back to the single-device loop."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.util import lowrank_csc, random_csc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def abi():
    from rcppml_amd import _abi
    assert _abi.detect()
    return _abi


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


def _fit(abi, A, W0, H0, ndev, w_solve=None, **kw):
    W, H = W0.copy(), H0.copy()
    with _Env(RCPPML_GPU_DEVICES=ndev if ndev > 1 else None, RCPPML_GPU_DEVICES_SHARE=1 if ndev > 1 else None,
              RCPPML_GPU_W_SOLVE=w_solve):
        res = abi.nmf_unified(A.p, A.i, A.x, A.rows, A.cols, W.shape[1], W, H, entry="ex", **kw)
    assert res["status"] == 0, res.get("error")
    res["W_T"], res["H"] = W, H
    return res


@pytest.mark.parametrize("ndev", [2, 3])
@pytest.mark.parametrize("solver", [0, 1])
def test_sharded_plugin_fit_matches_single_device_and_oracle(abi, ndev, solver):
    A = lowrank_csc(300, 1100, 6, 0.07, seed=17 + ndev)
    k = 12
    W0, H0 = O.init_factors(5, k, A.rows, A.cols, np.float64)
    kw = dict(max_iter=9, tol=0.0, solver_mode=solver, precision=1, L1_W=0.01, L1_H=0.02, L2_W=0.0, L2_H=0.03, want_history=True)
    one = _fit(abi, A, W0, H0, 1, **kw)
    many = _fit(abi, A, W0, H0, ndev, **kw)
    again = _fit(abi, A, W0, H0, ndev, **kw)
    ref = O.nmf_fit(A, W0, H0, np.float64, max_iter=9, tol=0.0, solver_mode=solver, L1=(0.01, 0.02), L2=(0.0, 0.03))
    assert many["iter"] == one["iter"] == ref.iter
    assert abs(many["loss"] - one["loss"]) / abs(one["loss"]) < 1e-9
    assert np.abs(many["loss_history"] - one["loss_history"]).max() / one["loss_history"].max() < 1e-9
    for key in ("W_T", "H"):
        assert np.abs(many[key] - one[key]).max() < 1e-8
    assert np.abs(many["d"] - one["d"]).max() / np.abs(one["d"]).max() < 1e-8
    assert abs(many["loss"] - ref.loss) / abs(ref.loss) < 1e-6
    assert np.abs(many["W_T"] - ref.W_T).max() < 1e-6 and np.abs(many["H"] - ref.H).max() < 1e-6
    # deterministic run to run
    assert np.array_equal(many["W_T"], again["W_T"]) and np.array_equal(many["H"], again["H"]) and many["loss"] == again["loss"]


@pytest.mark.parametrize("ndev", [2, 3, 8])
@pytest.mark.parametrize("solver", [0, 1])
def test_sharded_plugin_block_w_solve_equals_replicated(abi, ndev, solver):
    """RCPPML_GPU_W_SOLVE=block: device r solves its block of W's rows and one all-gather replicates them -- the same numbers as
    the default (every device solves all of W): the blocks are solved from identical (G, B) by the same kernels.  203 rows
    over 8 devices: blocks of 28 rows, the last one short (7 rows), m not a multiple of the block."""
    A = lowrank_csc(203, 900, 5, 0.08, seed=23 + ndev)
    k = 8
    W0, H0 = O.init_factors(3, k, A.rows, A.cols, np.float64)
    kw = dict(max_iter=7, tol=0.0, solver_mode=solver, precision=1, L1_W=0.01, L2_H=0.02, want_history=True)
    rep = _fit(abi, A, W0, H0, ndev, **kw)
    blk = _fit(abi, A, W0, H0, ndev, w_solve="block", **kw)
    assert blk["iter"] == rep["iter"]
    # (small blocks may take another CD kernel form than the whole side: same iterates up to rounding and the cd_tol exit)
    assert np.abs(blk["loss_history"] - rep["loss_history"]).max() / rep["loss_history"].max() < 1e-9
    assert np.abs(blk["W_T"] - rep["W_T"]).max() < 1e-8 and np.abs(blk["H"] - rep["H"]).max() < 1e-8
    one = _fit(abi, A, W0, H0, 1, **kw)
    assert abs(blk["loss"] - one["loss"]) / abs(one["loss"]) < 1e-9


def _rccl_mapped():
    with open("/proc/self/maps") as f:
        return any("librccl" in line for line in f)


@pytest.mark.parametrize("w_solve", [None, "block"])
@pytest.mark.parametrize("precision", [1, 0])
def test_one_rank_rccl_communicator_executes(abi, w_solve, precision):
    """RCPPML_GPU_DEVICES=1 + RCPPML_GPU_DEVICES_FORCE=1 WITHOUT the shared-device stand-in: the sharded loop with a real one-rank
    RCCL communicator -- dlopen of librccl, the seven dlsym's, ncclCommInitAll(comms, 1, devs), the grouped ncclAllReduce on
    [G | B | row sums] and (w_solve = block) ncclAllGather on W_T's row blocks all execute on this one-GPU box.  A one-rank sum
    is the identity, so the fit must equal, BIT FOR BIT, the same loop with the local-sum stand-in, and agree with the
    single-device loop (which scales H before forming its Gram, the sharded loop after the sum) to rounding."""
    A = lowrank_csc(300, 1100, 6, 0.07, seed=23)
    k = 12
    W0, H0 = O.init_factors(5, k, A.rows, A.cols, np.float64)
    kw = dict(max_iter=7, tol=0.0, solver_mode=0, precision=precision, L1_H=0.02, want_history=True)

    def run(share):
        W, H = W0.copy(), H0.copy()
        with _Env(RCPPML_GPU_DEVICES=1, RCPPML_GPU_DEVICES_FORCE=1, RCPPML_GPU_DEVICES_SHARE=1 if share else None, RCPPML_GPU_W_SOLVE=w_solve):
            res = abi.nmf_unified(A.p, A.i, A.x, A.rows, A.cols, k, W, H, entry="ex", **kw)
        assert res["status"] == 0, res.get("error")
        res["W_T"], res["H"] = W, H
        return res

    real = run(False)
    assert _rccl_mapped(), "librccl was not loaded: the RCCL path did not run"
    local = run(True)
    assert real["loss"] == local["loss"] and np.array_equal(real["loss_history"], local["loss_history"])
    assert np.array_equal(real["W_T"], local["W_T"]) and np.array_equal(real["H"], local["H"]) and np.array_equal(real["d"], local["d"])
    one = _fit(abi, A, W0, H0, 1, **kw)
    tol = 1e-9 if precision == 1 else 2e-4
    assert real["iter"] == one["iter"]
    assert abs(real["loss"] - one["loss"]) / abs(one["loss"]) < tol


def test_sharded_plugin_fit_fp32_convergence_and_norms(abi):
    A = lowrank_csc(200, 900, 5, 0.1, seed=3)
    k = 8
    W0, H0 = O.init_factors(11, k, A.rows, A.cols, np.float64)
    for norm_type in (0, 1):
        kw = dict(max_iter=40, tol=1e-4, solver_mode=0, precision=0, norm_type=norm_type)
        one = _fit(abi, A, W0, H0, 1, **kw)
        many = _fit(abi, A, W0, H0, 2, **kw)
        assert many["converged"] == one["converged"] and abs(many["iter"] - one["iter"]) <= 1
        assert abs(many["loss"] - one["loss"]) / abs(one["loss"]) < 2e-4
        assert np.all(np.diff(many["d"]) <= 0)                       # sorted by descending d


def test_sharded_plugin_uneven_and_empty_shards(abi):
    # all nonzeros in the first 40 of 400 columns: the nnz-balanced cut gives later shards few or no columns with entries
    A0 = random_csc(60, 40, 0.5, seed=4)
    p = np.concatenate([A0.p, np.full(360, A0.p[-1], np.int32)])
    A = O.Csc((60, 400), p, A0.i, A0.x)
    k = 4
    W0, H0 = O.init_factors(2, k, A.rows, A.cols, np.float64)
    kw = dict(max_iter=5, tol=0.0, solver_mode=0, precision=1)
    one = _fit(abi, A, W0, H0, 1, **kw)
    many = _fit(abi, A, W0, H0, 3, **kw)
    assert abs(many["loss"] - one["loss"]) / abs(one["loss"]) < 1e-9
    assert np.abs(many["H"] - one["H"]).max() < 1e-8 and np.abs(many["W_T"] - one["W_T"]).max() < 1e-8


def test_unsharded_configurations_fall_back_and_errors_are_loud(abi):
    A = lowrank_csc(120, 160, 4, 0.15, seed=11)
    M = random_csc(120, 160, 0.05, seed=12)
    k = 6
    W0, H0 = O.init_factors(9, k, A.rows, A.cols, np.float64)
    kw = dict(max_iter=4, tol=0.0, solver_mode=0, precision=1, mask=(M.p, M.i))
    one = _fit(abi, A, W0, H0, 1, **kw)
    many = _fit(abi, A, W0, H0, 2, **kw)          # explicit mask: the single-device loop runs
    assert many["loss"] == one["loss"] and np.array_equal(many["H"], one["H"])
    # two real devices requested on a one-GPU box: refused with a message, never silently single-device
    import torch
    if torch.cuda.device_count() < 2:
        W, H = W0.copy(), H0.copy()
        with _Env(RCPPML_GPU_DEVICES=2, RCPPML_GPU_DEVICES_SHARE=None):
            res = abi.nmf_unified(A.p, A.i, A.x, A.rows, A.cols, k, W, H, entry="ex", max_iter=2, tol=0.0, precision=1)
        assert res["status"] != 0 and "device" in res["error"]
