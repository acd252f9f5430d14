"""The one-kernel fit of small sparse matrices (rcppml_amd/csrc/kernels_small.hip.h, rcppml_hip_als_small_fit): the whole ALS loop --
fused right-hand side + solve per column, scaling, Gram, loss, convergence rule -- as one persistent kernel on one XCD, the path the
plugin takes for plain sparse MSE fits with k <= 16, m + n <= 3072 and nnz <= 2^17 (hawaiibirds, BASELINE configs[0]; the rule is
measured: profiles/r06_small_threshold.txt).  Ranks 17 .. 32 run the kernel's one-column-per-wavefront form, which the plugin never takes
(it loses to the multi-launch loop at every size) but the device-level op accepts: held to the oracle here all the same.

Checked against the CPU oracle's nmf_fit (same bars as the multi-launch loop's plugin tests: fp64 loss 1e-6 / factors 1e-6 and in
practice ~1e-12, fp32 2e-4 / 2e-3) AND against the multi-launch loop on the same inputs (RCPPML_GPU_NO_SMALL=1: the two paths differ
only in summation order), over both solvers, every option the kernel takes, ranks 1 .. 32, degenerate shapes, and run to run (bitwise).
The rest of the GPU suite runs with RCPPML_GPU_NO_SMALL=1 (tests/conftest.py) so that its small matrices keep exercising the
multi-launch kernels; this file is where the one-kernel path is held to the oracle."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.util import load_fixture, lowrank_csc, random_csc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def abi():
    from rcppml_amd import _abi
    assert _abi.detect()
    return _abi


class _Small:
    """Environment for one plugin call: the one-kernel path on (default of the library) or off."""

    def __init__(self, on):
        self.on = on

    def __enter__(self):
        self.old = os.environ.get("RCPPML_GPU_NO_SMALL")
        if self.on:
            os.environ.pop("RCPPML_GPU_NO_SMALL", None)
        else:
            os.environ["RCPPML_GPU_NO_SMALL"] = "1"

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("RCPPML_GPU_NO_SMALL", None)
        else:
            os.environ["RCPPML_GPU_NO_SMALL"] = self.old


def _fit(abi, A, W0, H0, small, precision=1, **kw):
    W, H = W0.astype(np.float64).copy(), H0.astype(np.float64).copy()
    with _Small(small):
        res = abi.nmf_unified(A.p, A.i, A.x, A.rows, A.cols, W.shape[1], W, H, entry="ex", precision=precision, want_history=True,
                              verbose=1 if small else 0, **kw)
    assert res["status"] == 0, res.get("error")
    res["W_T"], res["H"] = W, H
    return res


def _same(a, b, tol_loss, tol_fac):
    assert a["iter"] == b["iter"] and a["converged"] == b["converged"]
    assert np.abs(a["loss_history"] - b["loss_history"]).max() <= tol_loss * np.abs(b["loss_history"]).max()
    assert abs(a["loss"] - b["loss"]) <= tol_loss * abs(b["loss"])
    assert np.abs(a["W_T"] - b["W_T"]).max() < tol_fac and np.abs(a["H"] - b["H"]).max() < tol_fac
    assert np.abs(a["d"] - b["d"]).max() <= tol_fac * np.abs(b["d"]).max()


def _vs_oracle(res, ref, tol_loss, tol_fac):
    assert res["iter"] == ref.iter and res["converged"] == ref.converged
    assert abs(res["loss"] - ref.loss) <= tol_loss * abs(ref.loss)
    assert np.abs(res["loss_history"] - ref.loss_history).max() <= tol_loss * np.abs(ref.loss_history).max()
    assert np.abs(res["W_T"] - ref.W_T).max() < tol_fac and np.abs(res["H"] - ref.H).max() < tol_fac
    assert np.abs(res["d"] - ref.d).max() <= tol_fac * np.abs(ref.d).max()


def test_eligibility(abi):
    hb, ml = load_fixture("hawaiibirds"), load_fixture("movielens")
    assert abi.small_eligible(hb.rows, hb.cols, hb.nnz, 10)
    assert not abi.small_eligible(ml.rows, ml.cols, ml.nnz, 32)          # 4 477 columns x 32^2: the whole-chip kernels win there
    assert not abi.small_eligible(20000, 100000, 20000000, 64)
    assert not abi.small_eligible(100, 100, 500, 17)                      # one column per wavefront: loses at every size (profiles/r06_small_threshold.txt)
    assert not abi.small_eligible(800, 3200, 205276, 10) and abi.small_eligible(400, 2400, 76059, 10)


@pytest.mark.parametrize("solver", [1, 0])
def test_hawaiibirds_one_kernel_fit_fp64(abi, solver, capfd):
    """BASELINE configs[0] through the plugin boundary: oracle parity, the same fit as the multi-launch loop, and the convergence rule
    evaluated on the device (tol = 1e-4: same iteration count and flag)."""
    A = load_fixture("hawaiibirds")
    W0, H0 = O.init_factors(42, 10, A.rows, A.cols, np.float64)
    for kw in (dict(max_iter=30, tol=1e-4), dict(max_iter=12, tol=0.0)):
        ref = O.nmf_fit(A, W0, H0, np.float64, solver_mode=solver, **kw)
        capfd.readouterr()
        one = _fit(abi, A, W0, H0, True, solver_mode=solver, **kw)
        assert "one-kernel fit" in capfd.readouterr().err          # the path under test really ran
        multi = _fit(abi, A, W0, H0, False, solver_mode=solver, **kw)
        _vs_oracle(one, ref, 1e-6, 1e-6)
        _same(one, multi, 1e-9, 1e-7)
        again = _fit(abi, A, W0, H0, True, solver_mode=solver, **kw)
        assert np.array_equal(one["W_T"], again["W_T"]) and np.array_equal(one["H"], again["H"]) and one["loss"] == again["loss"]


@pytest.mark.parametrize("solver", [1, 0])
def test_hawaiibirds_one_kernel_fit_fp32(abi, solver):
    A = load_fixture("hawaiibirds")
    W0, H0 = O.init_factors(42, 10, A.rows, A.cols, np.float32)
    ref = O.nmf_fit(A, W0, H0, np.float32, max_iter=20, tol=0.0, solver_mode=solver)
    one = _fit(abi, A, W0, H0, True, precision=0, max_iter=20, tol=0.0, solver_mode=solver)
    multi = _fit(abi, A, W0, H0, False, precision=0, max_iter=20, tol=0.0, solver_mode=solver)
    assert one["iter"] == ref.iter
    assert abs(one["loss"] - ref.loss) <= 2e-4 * abs(ref.loss)
    assert np.abs(one["W_T"] - ref.W_T).max() < 2e-3 and np.abs(one["H"] - ref.H).max() < 2e-3
    _same(one, multi, 2e-4, 2e-3)


@pytest.mark.parametrize("k", [1, 2, 7, 16, 17, 24, 32])
@pytest.mark.parametrize("solver", [0, 1])
def test_ranks(abi, k, solver):
    """Every padded size (KP = 16 up to k = 16: the form the plugin takes; KP = 32 above: device-level op only), ranks that are not
    multiples of anything: the device-level op on caller-owned memory against the oracle's unsorted fit, and for k <= 16 the plugin."""
    import torch
    from rcppml_amd import als
    from rcppml_amd.data import CSC
    Ao = lowrank_csc(120, 190, 6, 0.12, seed=100 + k)
    assert abi.small_eligible(Ao.rows, Ao.cols, Ao.nnz, k) == (k <= 16)
    A = CSC((Ao.rows, Ao.cols), Ao.p, Ao.i, Ao.x)
    W0, H0 = O.init_factors(3 + k, k, A.rows, A.cols, np.float64)
    ref = O.nmf_fit(Ao, W0, H0, np.float64, max_iter=8, tol=0.0, solver_mode=solver, sort_model=False)
    ops = als.HipOps(0, "f64")
    a, at = ops.upload_csc(A), ops.upload_csc(A.transpose())
    W, H, d = ops.to_device(W0), ops.to_device(H0), ops.zeros((k,)) + 1
    res, hist = torch.zeros(8, dtype=torch.float64, device="cuda"), torch.zeros(8, dtype=torch.float64, device="cuda")
    ops.ctx.als_small_fit(ops.dt, a, at, A.rows, A.cols, k, W, H, d, ops.sumsq(a["x"]), solver_mode=solver, max_iter=8, tol=0.0,
                          loss_history=hist, result8=res)
    r = res.cpu().numpy()
    assert r[4] == 1.0 and int(r[0]) == ref.iter
    assert abs(r[2] - ref.loss) <= 1e-6 * abs(ref.loss) and np.abs(hist.cpu().numpy() - ref.loss_history).max() <= 1e-6 * ref.loss_history.max()
    assert np.abs(W.cpu().numpy() - ref.W_T).max() < 1e-6 and np.abs(H.cpu().numpy() - ref.H).max() < 1e-6
    assert np.abs(d.cpu().numpy() - ref.d).max() <= 1e-6 * ref.d.max()
    if k <= 16:
        refs = O.nmf_fit(Ao, W0, H0, np.float64, max_iter=8, tol=0.0, solver_mode=solver)
        one = _fit(abi, Ao, W0, H0, True, max_iter=8, tol=0.0, solver_mode=solver)
        _vs_oracle(one, refs, 1e-6, 1e-6)
        assert np.all(np.diff(one["d"]) <= 0)                      # sorted on the way out, like every fit


OPTIONS = [
    dict(L1=(0.0, 0.1)), dict(L1=(0.05, 0.02)), dict(L2=(0.01, 0.03)), dict(L1=(0.02, 0.0), L2=(0.0, 0.05)),
    dict(upper_bound=(0.01, 0.02)), dict(norm="L2"), dict(norm="none"), dict(nonneg=(False, True)), dict(nonneg=(True, False)),
    dict(cd_maxit=3), dict(cd_tol=1e-3),
]


@pytest.mark.parametrize("shape", [(150, 260), (330, 420)], ids=["one_side_four_abreast", "both_sides_four_abreast"])
@pytest.mark.parametrize("solver", [0, 1])
@pytest.mark.parametrize("opt", OPTIONS, ids=lambda o: ",".join("%s=%s" % kv for kv in o.items()))
def test_options(abi, opt, solver, shape):
    """L1 / L2 / upper bounds / scaling norm / non-negativity switches / CD limits, per solver, fp64 against the oracle and the multi-launch
    loop.  Sides with more columns than the kernel has wavefronts (256) run four columns per wavefront (k <= 16), the others one: the two
    shapes put the W side on either form."""
    if solver == 1 and ("cd_maxit" in opt or "cd_tol" in opt):
        pytest.skip("CD parameters")
    A = lowrank_csc(shape[0], shape[1], 5, 0.1, seed=9)
    k = 9
    W0, H0 = O.init_factors(11, k, A.rows, A.cols, np.float64)
    L1, L2, ub = opt.get("L1", (0.0, 0.0)), opt.get("L2", (0.0, 0.0)), opt.get("upper_bound", (0.0, 0.0))
    nn = opt.get("nonneg", (True, True))
    norm = {"L1": 0, "L2": 1, "none": 2}[opt.get("norm", "L1")]
    okw = dict(max_iter=7, tol=0.0, solver_mode=solver, L1=L1, L2=L2, ub=ub, nonneg=nn, norm_type=norm, cd_maxit=opt.get("cd_maxit", 100),
               cd_tol=opt.get("cd_tol", 1e-8))
    ref = O.nmf_fit(A, W0, H0, np.float64, **okw)
    gkw = dict(max_iter=7, tol=0.0, solver_mode=solver, L1_W=L1[0], L1_H=L1[1], L2_W=L2[0], L2_H=L2[1], ub_W=ub[0], ub_H=ub[1],
               nonneg_W=int(nn[0]), nonneg_H=int(nn[1]), norm_type=norm, cd_maxit=opt.get("cd_maxit", 100), cd_tol=opt.get("cd_tol", 1e-8))
    one = _fit(abi, A, W0, H0, True, **gkw)
    multi = _fit(abi, A, W0, H0, False, **gkw)
    # (without non-negativity / with a dead factor the fit is decided by rounding in BOTH implementations: compare loosely there)
    # (likewise bounds on both factors: DESIGN 7 -- the clamp decides by the last bit which entries sit at the bound)
    loose = not all(nn) or norm == 2 or "upper_bound" in opt
    _vs_oracle(one, ref, 1e-5 if loose else 1e-6, 1e-4 if loose else 1e-6)
    _same(one, multi, 1e-5 if loose else 1e-9, 1e-4 if loose else 1e-7)


@pytest.mark.parametrize("seed", range(8))
def test_random_shapes_and_degenerate_columns(abi, seed):
    """Random small matrices with empty columns and rows, k drawn from 1 .. 16, both solvers alternating, fp32 and fp64."""
    rng = np.random.default_rng(500 + seed)
    m, n = int(rng.integers(3, 200)), int(rng.integers(3, 300))
    k = int(rng.integers(1, min(16, m, n) + 1))
    A = random_csc(m, n, float(rng.uniform(0.02, 0.3)), seed=seed)
    if A.nnz == 0 or not abi.small_eligible(m, n, A.nnz, k):
        pytest.skip("degenerate draw")
    solver = seed % 2
    W0, H0 = O.init_factors(seed + 1, k, m, n, np.float64)
    ref = O.nmf_fit(A, W0, H0, np.float64, max_iter=6, tol=0.0, solver_mode=solver)
    one = _fit(abi, A, W0, H0, True, max_iter=6, tol=0.0, solver_mode=solver)
    multi = _fit(abi, A, W0, H0, False, max_iter=6, tol=0.0, solver_mode=solver)
    assert one["iter"] == ref.iter == multi["iter"]
    if np.isfinite(ref.loss) and ref.d.min() > 1e-9:              # (a dead factor is decided by rounding: DESIGN 7)
        _vs_oracle(one, ref, 1e-6, 1e-5)
        _same(one, multi, 1e-8, 1e-6)
    W32, H32 = W0.astype(np.float32), H0.astype(np.float32)
    r32 = O.nmf_fit(A, W32, H32, np.float32, max_iter=6, tol=0.0, solver_mode=solver)
    o32 = _fit(abi, A, W32, H32, True, precision=0, max_iter=6, tol=0.0, solver_mode=solver)
    if np.isfinite(r32.loss) and r32.d.min() > 1e-6:
        assert abs(o32["loss"] - r32.loss) <= 5e-4 * abs(r32.loss)


def test_device_level_op_and_early_stop(abi):
    """rcppml_hip_als_small_fit on caller-owned device memory (no plugin around it): result record, loss history, convergence rule with
    patience -- the reference's rule restated on the device against the oracle's iteration count."""
    import torch
    from rcppml_amd import als
    from rcppml_amd.data import CSC
    Ao = load_fixture("hawaiibirds")
    A = CSC((Ao.rows, Ao.cols), Ao.p, Ao.i, Ao.x)
    k = 10
    W0, H0 = O.init_factors(42, k, A.rows, A.cols, np.float64)
    ops = als.HipOps(0, "f64")
    a, at = ops.upload_csc(A), ops.upload_csc(A.transpose())
    for tol, patience in ((1e-3, 2), (1e-4, 5), (1e-2, 1)):
        W, H, d = ops.to_device(W0), ops.to_device(H0), ops.zeros((k,)) + 1
        tr = ops.sumsq(a["x"])
        hist = torch.zeros(60, dtype=torch.float64, device="cuda")
        res = torch.zeros(8, dtype=torch.float64, device="cuda")
        ops.ctx.als_small_fit(ops.dt, a, at, A.rows, A.cols, k, W, H, d, tr, solver_mode=0, max_iter=60, tol=tol, patience=patience,
                              loss_history=hist, result8=res)
        r = res.cpu().numpy()
        ref = O.nmf_fit(Ao, W0, H0, np.float64, max_iter=60, tol=tol, patience=patience, solver_mode=0, sort_model=False)
        assert r[4] == 1.0 and int(r[0]) == ref.iter and bool(r[1]) == ref.converged
        assert abs(r[2] - ref.loss) <= 1e-8 * abs(ref.loss)
        h = hist.cpu().numpy()[:ref.iter]
        assert np.abs(h - ref.loss_history).max() <= 1e-8 * ref.loss_history.max()
        assert np.abs(W.cpu().numpy() - ref.W_T).max() < 1e-7 and np.abs(H.cpu().numpy() - ref.H).max() < 1e-7
        assert np.abs(d.cpu().numpy() - ref.d).max() <= 1e-7 * ref.d.max()


def test_r_surface_takes_the_one_kernel_path():
    """nmf(hawaiibirds, k = 10) -- what a user of the R surface gets with the GPU backend on -- equals the multi-launch fit."""
    from rcppml_amd import data, nmf
    Ao = load_fixture("hawaiibirds")
    A = data.CSC((Ao.rows, Ao.cols), Ao.p, Ao.i, Ao.x)
    with _Small(True):
        a = nmf.nmf(A, 10, seed=42, tol=1e-5, maxit=40, precision="fp64")
    with _Small(False):
        b = nmf.nmf(A, 10, seed=42, tol=1e-5, maxit=40, precision="fp64")
    assert a.misc["iter"] == b.misc["iter"] and abs(a.misc["loss"] - b.misc["loss"]) <= 1e-9 * abs(b.misc["loss"])
    assert np.abs(a.w - b.w).max() < 1e-7 and np.abs(a.h - b.h).max() < 1e-7


def test_give_up_path_restarts_on_the_multi_launch_loop(abi, capfd):
    """The kernel's barrier gives up when its workgroups are not all resident on one XCD (another persistent fit, another partition mode).
    Forced here by presetting the abort flag (RCPPML_OPT_SMALL_GIVE_UP / RCPPML_GPU_SMALL_GIVE_UP_TEST): the device op reports it in
    result8[4] and leaves promptly, and the plugin restarts the fit from the caller's factors on the multi-launch loop -- same result as
    with the path switched off, bit for bit."""
    import torch
    from rcppml_amd import als
    from rcppml_amd.data import CSC
    Ao = load_fixture("hawaiibirds")
    k = 10
    W0, H0 = O.init_factors(42, k, Ao.rows, Ao.cols, np.float64)
    A = CSC((Ao.rows, Ao.cols), Ao.p, Ao.i, Ao.x)
    ops = als.HipOps(0, "f64")
    a, at = ops.upload_csc(A), ops.upload_csc(A.transpose())
    W, H, d = ops.to_device(W0), ops.to_device(H0), ops.zeros((k,)) + 1
    res = torch.zeros(8, dtype=torch.float64, device="cuda")
    ops.ctx.set_option(abi.OPT_SMALL_GIVE_UP, 1)
    ops.ctx.als_small_fit(ops.dt, a, at, A.rows, A.cols, k, W, H, d, ops.sumsq(a["x"]), max_iter=5, tol=0.0, result8=res)
    assert float(res[4].item()) != 1.0                      # gave up: the caller must not use W / H
    ops.ctx.set_option(abi.OPT_SMALL_GIVE_UP, 0)
    ops.ctx.als_small_fit(ops.dt, a, at, A.rows, A.cols, k, ops.to_device(W0), ops.to_device(H0), d, ops.sumsq(a["x"]), max_iter=5, tol=0.0, result8=res)
    assert float(res[4].item()) == 1.0
    off = _fit(abi, Ao, W0, H0, False, max_iter=9, tol=0.0, solver_mode=0)
    os.environ["RCPPML_GPU_SMALL_GIVE_UP_TEST"] = "1"
    try:
        capfd.readouterr()
        gave_up = _fit(abi, Ao, W0, H0, True, max_iter=9, tol=0.0, solver_mode=0)
        err = capfd.readouterr().err
    finally:
        os.environ.pop("RCPPML_GPU_SMALL_GIVE_UP_TEST", None)
    assert "gave up at its barrier" in err and "one-kernel fit)" not in err
    assert np.array_equal(gave_up["W_T"], off["W_T"]) and np.array_equal(gave_up["H"], off["H"]) and gave_up["loss"] == off["loss"]
    assert np.array_equal(gave_up["loss_history"], off["loss_history"])
