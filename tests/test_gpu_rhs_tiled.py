"""GPU parity tests of the LDS row-tiled right-hand side (rcppml_hip_rhs_plan_create / rcppml_hip_rhs_planned,
rcppml_amd/csrc/kernels_rhs_tiled.hip.h) against the CPU oracle's rhs (reference primitives/cpu/rhs.hpp:52-70).

What is exercised: both row sizes the kernel is compiled for (256 and 512 bytes: fp32 k = 64 / 128, fp64 k = 32 / 64), every
slot count (forced), automatic slot choice, one and several row partitions, spills (segments longer than the slot
count), empty columns, a last tile shorter than 64 KiB, more workgroups than one, ineligible shapes (plan is None) and
run-to-run bitwise determinism.  Tolerances: fp64 1e-11, fp32 5e-5 relative to the largest entry (summation order
differs from the oracle's sequential loop: spilled nonzeros first, then tile by tile)."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import random_csc, rel_err

pytestmark = pytest.mark.gpu

TOL = {np.float32: 5e-5, np.float64: 1e-11}


@pytest.fixture(scope="module")
def env():
    import torch
    from rcppml_amd import _abi
    return torch, _abi, _abi.Context(0)


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _run(env, A, k, dtype, partitions=0, slots=0, expect_plan=True):
    torch, _abi, ctx = env
    dt = _abi.F32 if dtype == np.float32 else _abi.F64
    tt = torch.float32 if dtype == np.float32 else torch.float64
    F = np.random.default_rng(k + A.rows).standard_normal((A.rows, k)).astype(dtype)
    dp, di, dx = _dev(torch, A.p), _dev(torch, A.i), _dev(torch, A.values(dtype))
    plan = ctx.rhs_plan(dt, dp, di, dx, A.cols, A.rows, k, partitions, slots)
    if not expect_plan:
        assert plan is None
        return None
    assert plan is not None
    info = plan.info()
    dF = _dev(torch, F)
    dB = torch.full((A.cols, k), 7.0, dtype=tt, device="cuda")
    ctx.rhs_planned(plan, dF, dB)
    B = dB.cpu().numpy()
    assert rel_err(B, O.rhs(A, F, dtype)) < TOL[dtype], info
    dB2 = torch.full((A.cols, k), -3.0, dtype=tt, device="cuda")
    ctx.rhs_planned(plan, dF, dB2)
    assert np.array_equal(B, dB2.cpu().numpy()), "planned rhs must be bitwise reproducible"
    return info


@pytest.mark.parametrize("dtype,k", [(np.float32, 64), (np.float32, 128), (np.float64, 32), (np.float64, 64), (np.float64, 128)])
@pytest.mark.parametrize("slots", [0, 2, 3, 4, 5, 6, 8])
def test_planned_rhs_slots(env, dtype, k, slots):
    # 700 rows: 2.7 tiles of 256 rows (k*s = 256 B) / 5.5 tiles of 128 rows; the last tile is short
    A = random_csc(700, 1500, 0.012, seed=slots + k)
    info = _run(env, A, k, dtype, slots=slots)        # 0: the window plan (round 4); 2..8: the slab plan with that many slots
    if slots:
        assert info["slots"] == slots


# window plans (kernels_rhs_win.hip.h): slots = 100 + 4 x (slots per column and 32 KiB phase); quarter steps up to 2, halves above
WIN_RATES = [101, 102, 103, 104, 105, 106, 107, 108, 110, 112, 114, 116, 120, 124]


@pytest.mark.parametrize("dtype,k", [(np.float32, 64), (np.float32, 128), (np.float64, 32), (np.float64, 64), (np.float64, 128)])
@pytest.mark.parametrize("code", WIN_RATES)
def test_window_rhs_rates(env, dtype, k, code):
    # 700 rows: 5.5 phases of 128 rows (256-byte rows) ... 22 phases of 32 rows (1 KiB rows): a short last tile, a phase count
    # that is not a multiple of four (the kernel pads with empty phases), overflow at the low rates
    from rcppml_amd._abi import BackendError
    A = random_csc(700, 1500, 0.012, seed=code + k)
    try:
        info = _run(env, A, k, dtype, slots=code)
    except BackendError as e:           # a rate whose slot blocks do not fit the LDS behind the ring for every compiled shape
        assert "not available" in str(e)
        return
    assert abs(info["slots"] - (code - 100) // 4) == 0          # plan_info reports the rate truncated to an integer


@pytest.mark.parametrize("partitions", [1, 2, 3, 5, 8])
@pytest.mark.parametrize("dtype,k", [(np.float32, 64), (np.float64, 64)])
def test_window_rhs_partitions_and_overflow(env, partitions, dtype, k):
    # 5 % dense: ~6 nonzeros per (column, 128-row tile) against one slot per phase -> most nonzeros overflow into the
    # finishing kernel; partitions cut the 24 / 47 phases into uneven runs
    A = random_csc(3000, 700, 0.05, seed=partitions)
    info = _run(env, A, k, dtype, partitions=partitions, slots=104)
    assert info["spilled_nnz"] > 0.3 * A.nnz
    assert info["partitions"] == min(partitions, info["tiles"])


def test_window_rhs_large_offsets_and_ring_wrap(env):
    # 9 000 rows = 71 phases: every ring buffer is reused many times, offsets beyond 64 KiB (buffers 2 and 3) in use
    A = random_csc(9000, 3000, 0.008, seed=4)
    for code in (0, 106, 107):
        _run(env, A, 64, np.float32, slots=code)
    _run(env, A, 64, np.float64, slots=0)
    _run(env, A.transpose(), 128, np.float32, slots=0)


@pytest.mark.parametrize("partitions", [1, 2, 3, 8])
def test_planned_rhs_partitions(env, partitions):
    A = random_csc(3000, 900, 0.004, seed=partitions)
    info = _run(env, A, 64, np.float32, partitions=partitions)
    assert info["partitions"] == min(partitions, info["tiles"])


def test_planned_rhs_spills_and_empty_columns(env):
    # dense-ish columns: every (column, tile) segment holds ~25 nonzeros, far beyond any slot count -> most spill
    A = random_csc(512, 300, 0.10, seed=5)
    # empty columns in the middle and at the end
    p = A.p.copy()
    keep = np.ones(A.cols, bool)
    keep[[3, 4, 100, 299]] = False
    cnt = np.diff(p) * keep
    sel = np.repeat(keep, np.diff(p))
    A2 = O.Csc((A.rows, A.cols), np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32), A.i[sel], A.x[sel])
    info = _run(env, A2, 64, np.float32, slots=2)
    assert info["spilled_nnz"] > 0
    info = _run(env, A2, 32, np.float64, slots=8)
    assert info["spilled_nnz"] > 0


def test_planned_rhs_many_workgroups(env):
    # 40 000 columns on 256 CUs: several rounds per wave, short last workgroup
    A = random_csc(600, 40000, 0.004, seed=9)
    info = _run(env, A, 64, np.float32)
    assert info["workgroups_per_partition"] >= 1
    _run(env, A.transpose(), 64, np.float32, partitions=2)


def test_planned_rhs_ineligible_shapes(env):
    A = random_csc(300, 200, 0.05, seed=1)
    _run(env, A, 10, np.float32, expect_plan=False)      # 40-byte rows
    _run(env, A, 100, np.float64, expect_plan=False)     # 800-byte rows
    _run(env, A, 256, np.float32, expect_plan=False)     # 1 KiB rows are compiled for fp64 only
    # unsorted rows inside a column
    i = A.i.copy()
    s, e = A.p[7], A.p[8]
    if e - s >= 2:
        i[s], i[s + 1] = i[s + 1], i[s]
    B = O.Csc((A.rows, A.cols), A.p, i, A.x)
    torch, _abi, ctx = env
    assert ctx.rhs_plan(_abi.F32, _dev(torch, B.p), _dev(torch, B.i), _dev(torch, B.values(np.float32)), B.cols, B.rows, 64) is None


def test_hypersparse_input_gets_no_plan(env):
    """The slot stream has ncols x ntiles x S slots whatever nnz is: a 200 000 x 200 000 matrix with ~1.1 M nonzeros would need
    ~313 M slots (1.9 GB) and ~1 ms per product where the gather kernel takes ~14 us.  The planner declines (fill below a
    quarter / stream far beyond the CSC itself) and the harness keeps the gather kernel."""
    torch, _abi, ctx = env
    n = 200000
    rs = np.random.default_rng(3)
    cols = np.sort(rs.integers(0, n, size=1100000))
    rows = rs.integers(0, n, size=cols.shape[0])
    key = np.unique(cols.astype(np.int64) * n + rows)            # sorted by column, then row; duplicates removed
    cols, rows = (key // n).astype(np.int32), (key % n).astype(np.int32)
    p = np.zeros(n + 1, np.int32)
    np.cumsum(np.bincount(cols, minlength=n), out=p[1:])
    x = rs.uniform(0.5, 1.5, size=rows.shape[0]).astype(np.float32)
    free0 = torch.cuda.mem_get_info()[0]
    plan = ctx.rhs_plan(_abi.F32, _dev(torch, p), _dev(torch, rows), _dev(torch, x), n, n, 64)
    assert plan is None
    assert free0 - torch.cuda.mem_get_info()[0] < (64 << 20)       # and it did not get there by allocating the stream first
    # the products still run (gather kernel) and are right: checksum sum_j B(:, j) = F^T (A 1)
    F = rs.uniform(size=(n, 64)).astype(np.float32)
    dB = torch.empty((n, 64), dtype=torch.float32, device="cuda")
    ctx.rhs(_abi.F32, _dev(torch, p), _dev(torch, rows), _dev(torch, x), n, _dev(torch, F), 64, dB)
    rowsum = np.bincount(rows, weights=x.astype(np.float64), minlength=n)
    want = F.astype(np.float64).T @ rowsum
    got = dB.double().sum(dim=0).cpu().numpy()
    assert np.abs(got - want).max() / np.abs(want).max() < 1e-5


@pytest.mark.parametrize("dtype,k", [(np.float32, 64), (np.float64, 64)])
def test_window_plan_in_two_steps_equals_one_step(env, dtype, k):
    """rcppml_hip_rhs_plan_create_indices + _set_values (the plugin builds the index half while the values cross PCIe) gives the
    plan rcppml_hip_rhs_plan_create gives: bitwise the same product; set_values again with other values re-uses the schedule; a
    plan without values refuses to run."""
    torch, _abi, ctx = env
    dt = _abi.F32 if dtype == np.float32 else _abi.F64
    tt = torch.float32 if dtype == np.float32 else torch.float64
    A = random_csc(3000, 2500, 0.02, seed=41)
    F = np.random.default_rng(5).standard_normal((A.rows, k)).astype(dtype)
    dp, di, dx, dF = _dev(torch, A.p), _dev(torch, A.i), _dev(torch, A.values(dtype)), _dev(torch, F)
    one = ctx.rhs_plan(dt, dp, di, dx, A.cols, A.rows, k, 2, 106)
    two = ctx.rhs_plan_indices(dt, dp, di, A.cols, A.rows, k, 2, 106)
    assert one is not None and two is not None and two.info() == one.info()
    B1 = torch.empty((A.cols, k), dtype=tt, device="cuda")
    B2 = torch.full((A.cols, k), 5.0, dtype=tt, device="cuda")
    with pytest.raises(_abi.BackendError):
        ctx.rhs_planned(two, dF, B2)
    ctx.rhs_plan_set_values(two, dx)
    ctx.rhs_planned(one, dF, B1)
    ctx.rhs_planned(two, dF, B2)
    assert torch.equal(B1, B2)
    assert rel_err(B2.cpu().numpy(), O.rhs(A, F, dtype)) < TOL[dtype]
    x2 = A.values(dtype) * dtype(-0.5)
    dx2 = _dev(torch, x2)
    ctx.rhs_plan_set_values(two, dx2)
    ctx.rhs_planned(two, dF, B2)
    A2 = O.Csc((A.rows, A.cols), A.p, A.i, A.x * -0.5)
    assert rel_err(B2.cpu().numpy(), O.rhs(A2, F, dtype)) < TOL[dtype]


@pytest.mark.parametrize("dtype,k", [(np.float32, 64), (np.float64, 64)])
def test_window_rhs_non_finite_rows_stay_in_their_columns(env, dtype, k):
    """Empty slots read a dedicated row of zeros in LDS, not a row of F: an Inf in F reaches exactly the columns that have a nonzero in
    that row (the reference's behaviour: primitives/cpu/rhs.hpp adds a * F(:, i) for the stored entries only), however the padding
    of the slot stream falls."""
    torch, _abi, ctx = env
    dt = _abi.F32 if dtype == np.float32 else _abi.F64
    tt = torch.float32 if dtype == np.float32 else torch.float64
    A = random_csc(2000, 1200, 0.01, seed=77)
    R = 32768 // (k * np.dtype(dtype).itemsize)
    F = np.random.default_rng(1).standard_normal((A.rows, k)).astype(dtype)
    bad_rows = np.arange(0, A.rows, R)              # the first row of every tile: where r2/r3-style padding would have pointed
    F[bad_rows] = np.inf
    plan = ctx.rhs_plan(dt, _dev(torch, A.p), _dev(torch, A.i), _dev(torch, A.values(dtype)), A.cols, A.rows, k, 2, 105)
    assert plan is not None and plan.info()["fill"] < 0.9          # there IS padding
    dB = torch.empty((A.cols, k), dtype=tt, device="cuda")
    ctx.rhs_planned(plan, _dev(torch, F), dB)
    B = dB.cpu().numpy()
    touched = np.zeros(A.cols, bool)
    isbad = np.zeros(A.rows, bool)
    isbad[bad_rows] = True
    for j in range(A.cols):
        touched[j] = isbad[A.i[A.p[j]:A.p[j + 1]]].any()
    assert touched.any() and (~touched).sum() > A.cols // 2
    assert np.all(np.isfinite(B[~touched])) and not np.any(np.isfinite(B[touched]).all(axis=1))
    Fz = F.copy()
    Fz[bad_rows] = 0
    ref = O.rhs(A, Fz, dtype)
    assert rel_err(B[~touched], ref[~touched]) < TOL[dtype]
