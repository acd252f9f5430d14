"""GPU parity tests of the StreamPress v2 `.spz` reader (rcppml_amd/csrc/ops_spz.hip) through the C-ABI:
bit-exact against the CPU oracle (oracle/spz_oracle.cpp, itself pinned to the reference codec) and against the golden
fixtures -- files written by the reference ENCODER with the CSC its DECODER returns -- including the bundled
pbmc3k.spz; then the reference's end-to-end use: sp_read_gpu -> rcppml_gpu_nmf_zerocopy_double -> sp_free_gpu."""
import hashlib
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "spz_vectors.npz"))
PBMC = os.path.join(HERE, "golden", "pbmc3k.spz")


@pytest.fixture(scope="module")
def env():
    import torch
    from rcppml_amd import _abi
    return torch, _abi, _abi.Context(0)


def _decode(env, buf):
    torch, _abi, ctx = env
    st, m, n, nnz, vt = _abi.spz_info(buf)
    assert st == 0
    dp = torch.full((n + 1,), -7, dtype=torch.int32, device="cuda")
    di = torch.full((max(nnz, 1),), -7, dtype=torch.int32, device="cuda")
    dx = torch.full((max(nnz, 1),), -7.0, dtype=torch.float64, device="cuda")
    ctx.spz_decode(buf, dp, di, dx)
    return m, n, nnz, vt, dp.cpu().numpy(), di.cpu().numpy()[:nnz], dx.cpu().numpy()[:nnz]


@pytest.mark.parametrize("name", [str(n) for n in GOLD["names"]])
def test_decode_matches_reference_and_oracle(env, name):
    """Every value type (uint8 / uint16 with both escape paths / fp32 / fp16 / quant8 / fp64), empty columns, a heavy
    column, a one-column file, and a file with nonzero-free chunks (whose column pointers are the reference decoder's
    reading of the bytes that follow -- replicated).  Integer / byte work: bit-exact."""
    buf = GOLD[name + "_spz"]
    m, n, nnz, vt, p, i, x = _decode(env, buf)
    assert [m, n, nnz, vt] == list(GOLD[name + "_info"][:4])
    po, io, xo = O.spz_decode(buf)
    assert np.array_equal(p.view(np.uint32), po) and np.array_equal(i.view(np.uint32), io) and np.array_equal(x, xo)
    assert np.array_equal(p.view(np.uint32), GOLD[name + "_p"]) and np.array_equal(i.view(np.uint32), GOLD[name + "_i"])
    assert np.array_equal(x.view(np.uint64), GOLD[name + "_x"].view(np.uint64))


def test_bundled_pbmc3k(env):
    """inst/extdata/pbmc3k.spz (13714 x 2700, 2.28 M nonzeros, uint16 counts, 11 chunks): SHA-256 of the decoded arrays
    = the reference decoder's, and equal to the oracle's arrays."""
    buf = np.fromfile(PBMC, np.uint8)
    m, n, nnz, vt, p, i, x = _decode(env, buf)
    assert [m, n, nnz, vt] == list(GOLD["pbmc3k_info"][:4])
    sha = lambda a: np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)
    assert np.array_equal(sha(p.view(np.uint32)), GOLD["pbmc3k_sha_p"])
    assert np.array_equal(sha(i.view(np.uint32)), GOLD["pbmc3k_sha_i"])
    assert np.array_equal(sha(x), GOLD["pbmc3k_sha_x"])
    po, io, xo = O.spz_decode(buf)
    assert np.array_equal(p.view(np.uint32), po) and np.array_equal(i.view(np.uint32), io) and np.array_equal(x, xo)
    # size-independent properties of a CSC
    assert p[0] == 0 and p[-1] == nnz and np.all(np.diff(p) >= 0) and i.min() >= 0 and i.max() < m
    seg = np.repeat(np.arange(n), np.diff(p))
    assert np.all((np.diff(i) > 0) | (np.diff(seg) > 0))             # rows strictly increasing inside every column


def test_bad_files_are_rejected(env):
    torch, _abi, ctx = env
    buf = GOLD["u8_spz"].copy()
    assert _abi.spz_info(buf[:4])[0] == 3                            # too small (sp_gpu_bridge.cu:69-73)
    v3 = buf.copy(); v3[4] = 3
    assert _abi.spz_info(v3)[0] == 4                                 # not v2 (:75-81)
    rs = buf.copy(); rs[42] = 1
    assert _abi.spz_info(rs)[0] == 0                                 # flagged row-sorted without a stored permutation: decodes
    m0, n0, nnz0, vt0, p0, i0, x0 = _decode(env, buf)                # unpermuted, as the reference does (sparsepress_v2.hpp:1094)
    m1, n1, nnz1, vt1, p1, i1, x1 = _decode(env, rs)
    assert np.array_equal(p0, p1) and np.array_equal(i0, i1) and np.array_equal(x0, x1)
    trunc = buf[: buf.size // 2]
    d = torch.zeros(4096, dtype=torch.int32, device="cuda")
    dx = torch.zeros(4096, dtype=torch.float64, device="cuda")
    with pytest.raises(_abi.BackendError):
        ctx.spz_decode(trunc, d, d.clone(), dx)
    r = _abi.sp_read_gpu("/nonexistent/file.spz")
    assert r["status"] == 1 and r["col_ptr"] == 0.0


def test_read_then_zero_copy_fit(env):
    """R/sp_gpu.R: h <- sp_read_gpu(path); nmf on the device-resident CSC; sp_free_gpu(h).  The fit must equal the
    73-pointer fit on the host copy of the same matrix."""
    torch, _abi, ctx = env
    h = _abi.sp_read_gpu(PBMC)
    assert h["status"] == 0, h["error"]
    m, n, nnz = h["m"], h["n"], h["nnz"]
    assert [m, n, nnz] == list(GOLD["pbmc3k_info"][:3]) and h["col_ptr"] != 0.0
    po, io, xo = O.spz_decode(np.fromfile(PBMC, np.uint8))
    k = 8
    W0, H0 = O.init_factors(3, k, m, n, np.float64)
    W1, H1 = W0.copy(), H0.copy()
    r1 = _abi.nmf_zerocopy(h["col_ptr"], h["row_idx"], h["values"], m, n, nnz, k, W1, H1, max_iter=4, tol=0.0)
    assert r1["status"] == 0, r1.get("error")
    W2, H2 = W0.copy(), H0.copy()
    r2 = _abi.nmf_unified(po.astype(np.int32), io.astype(np.int32), xo, m, n, k, W2, H2, entry="double", max_iter=4, tol=0.0, solver_mode=0)
    assert r2["status"] == 0
    assert r1["loss"] == r2["loss"] and np.array_equal(W1, W2) and np.array_equal(H1, H2)
    assert _abi.sp_free_gpu(h) == 0 and h["col_ptr"] == 0.0 and h["row_idx"] == 0.0 and h["values"] == 0.0


def test_malformed_files_never_read_out_of_bounds(env):
    """Fuzz of the host parser: truncations at every 97th byte and 600 single-byte corruptions of a valid file (header,
    chunk index, count sections, stream sizes) either decode or are rejected with a status -- never a crash, and whatever
    sp_read_gpu hands out is a well-formed CSC (monotone column pointers ending at nnz, rows inside the matrix)."""
    import os, tempfile
    torch, _abi, ctx = env
    good = GOLD["u16_escape_spz"].copy() if "u16_escape_spz" in GOLD.files else GOLD["u8_spz"].copy()
    rng = np.random.default_rng(0)
    cases = [good[:cut] for cut in range(6, good.size, 97)]
    for _ in range(600):
        b = good.copy()
        pos = int(rng.integers(0, min(b.size, 4096)))
        b[pos] = rng.integers(0, 256)
        cases.append(b)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "f.spz")
        for b in cases:
            b.tofile(path)
            r = _abi.sp_read_gpu(path)
            if r["status"] == 0:
                m, n, nnz = r["m"], r["n"], r["nnz"]
                p = torch.empty(n + 1, dtype=torch.int32, device="cuda")
                _abi.copy_from_device_address(p, r["col_ptr"], (n + 1) * 4)
                pp = p.cpu().numpy()
                assert pp[0] == 0 and pp[-1] == nnz and np.all(np.diff(pp) >= 0)
                if nnz:
                    i = torch.empty(nnz, dtype=torch.int32, device="cuda")
                    _abi.copy_from_device_address(i, r["row_idx"], nnz * 4)
                    ii = i.cpu().numpy()
                    assert ii.min() >= 0 and ii.max() < m
                assert _abi.sp_free_gpu(r) == 0
            else:
                assert r["status"] in (2, 3, 4, 5) and r["col_ptr"] == 0.0


def test_row_sorted_file_through_reader_and_zero_copy_fit(env, tmp_path):
    """A row_sort = TRUE file (st_convert's default) through rcppml_sp_read_gpu: the device CSC equals what the reference
    decoder returns (rows mapped through the stored permutation, no longer ascending inside a column), and the zero-copy
    fit on it equals the 73-pointer fit on the host copy of the same arrays."""
    torch, _abi, ctx = env
    buf = GOLD["rowsort_u8_spz"]
    path = str(tmp_path / "rowsort.spz")
    buf.tofile(path)
    h = _abi.sp_read_gpu(path)
    assert h["status"] == 0 and [h["m"], h["n"], h["nnz"]] == list(GOLD["rowsort_u8_info"][:3])
    try:
        n, nnz = h["n"], h["nnz"]
        dp = torch.empty(n + 1, dtype=torch.int32, device="cuda")
        di = torch.empty(nnz, dtype=torch.int32, device="cuda")
        dx = torch.empty(nnz, dtype=torch.float64, device="cuda")
        _abi.copy_from_device_address(dp, h["col_ptr"], 4 * (n + 1))
        _abi.copy_from_device_address(di, h["row_idx"], 4 * nnz)
        _abi.copy_from_device_address(dx, h["values"], 8 * nnz)
        p, i, x = dp.cpu().numpy(), di.cpu().numpy(), dx.cpu().numpy()
        assert np.array_equal(p.view(np.uint32), GOLD["rowsort_u8_p"]) and np.array_equal(i.view(np.uint32), GOLD["rowsort_u8_i"])
        assert np.array_equal(x, GOLD["rowsort_u8_x"])
        k = 5
        W0, H0 = O.init_factors(3, k, h["m"], n, np.float64)
        W1, H1 = W0.copy(), H0.copy()
        r1 = _abi.nmf_unified(p, i, x, h["m"], n, k, W1, H1, entry="double", max_iter=4, tol=0.0, solver_mode=0)
        W2, H2 = W0.copy(), H0.copy()
        r2 = _abi.nmf_zerocopy(h["col_ptr"], h["row_idx"], h["values"], h["m"], n, nnz, k, W2, H2, max_iter=4, tol=0.0)
        assert r1["status"] == 0 and r2["status"] == 0
        assert r1["loss"] == r2["loss"] and np.array_equal(W1, W2) and np.array_equal(H1, H2)
        A = O.Csc((h["m"], n), p, i, x)
        ref = O.nmf_fit(A, W0, H0, np.float64, max_iter=4, tol=0.0, solver_mode=0)
        assert abs(r1["loss"] - ref.loss) <= 1e-6 * abs(ref.loss)
    finally:
        _abi.sp_free_gpu(h)
