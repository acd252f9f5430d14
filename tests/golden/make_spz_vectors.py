#!/usr/bin/env python3
"""tests/golden/make_spz_vectors.py -- StreamPress v2 fixtures produced BY THE REFERENCE ITSELF.

Encodes small seeded CSC matrices with the reference's v2 ENCODER and decodes them (and the bundled
inst/extdata/pbmc3k.spz, the data file the reference's tests/testthat/test_streampress_cpp.R:129-143 reads) with the
reference's v2 DECODER, both through oracle/_ref/libref_spz.so (built by `make -C oracle ref` from /root/reference,
sources never copied).  Stores the .spz byte streams and the decoded p / i / x in tests/golden/spz_vectors.npz, and a
copy of the bundled data file in tests/golden/pbmc3k.spz with its decoded CSC as digests.  Data only."""
import ctypes as C
import hashlib
import os
import shutil
import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
L = C.CDLL(os.path.join(root, "oracle", "_ref", "libref_spz.so"))
L.ref_spz_encode.restype = C.c_uint64
u8p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))
vp = lambda a: a.ctypes.data_as(C.c_void_p)


def ref_decode(buf):
    m, n, nc = C.c_uint32(), C.c_uint32(), C.c_uint32()
    nnz = C.c_uint64()
    vt, rs = C.c_int(), C.c_int()
    assert L.ref_spz_info(u8p(buf), C.c_uint64(buf.size), C.byref(m), C.byref(n), C.byref(nnz), C.byref(vt), C.byref(rs), C.byref(nc)) == 0
    p = np.zeros(n.value + 1, np.uint32)
    i = np.zeros(nnz.value, np.uint32)
    x = np.zeros(nnz.value, np.float64)
    assert L.ref_spz_decode(u8p(buf), C.c_uint64(buf.size), 1, vp(p), vp(i), vp(x)) == 0
    return dict(m=m.value, n=n.value, nnz=nnz.value, vt=vt.value, row_sorted=rs.value, chunks=nc.value, p=p, i=i, x=x)


L.ref_spz_encode_rowsort.restype = C.c_uint64


def ref_encode(m, n, p, i, x, precision, chunk_cols, row_sort=False):
    cap = 64 + 16 * x.size + 4096 * (n // chunk_cols + 2) + 8 * n + 8 * m
    out = np.zeros(cap, np.uint8)
    sz = (L.ref_spz_encode_rowsort if row_sort else L.ref_spz_encode)(C.c_uint32(m), C.c_uint32(n), C.c_uint64(x.size), vp(p), vp(i), vp(x), precision.encode(), C.c_uint32(chunk_cols), u8p(out), C.c_uint64(cap))
    assert 0 < sz <= cap, (sz, cap)
    return out[:sz].copy()


def random_csc(m, n, density, seed, kind):
    rng = np.random.default_rng(seed)
    cnt = rng.binomial(m, density, size=n).astype(np.int64)
    if n > 3:
        cnt[1] = 0                       # an empty column
        cnt[n - 2] = min(m, 3 * int(cnt.max()) + 1)   # a heavy column
    p = np.zeros(n + 1, np.uint32)
    p[1:] = np.cumsum(cnt)
    i = np.concatenate([np.sort(rng.choice(m, size=c, replace=False)) for c in cnt]).astype(np.uint32) if p[-1] else np.zeros(0, np.uint32)
    nnz = int(p[-1])
    if kind == "counts":                 # small counts with a few large values (escape path of the value stream)
        x = rng.poisson(2.0, nnz).astype(np.float64) + 1
        x[rng.random(nnz) < 0.01] = rng.integers(300, 70000, size=int((rng.random(nnz) < 0.01).sum()) or 1)[0]
    elif kind == "bytes":
        x = rng.integers(1, 200, nnz).astype(np.float64)
    else:
        x = rng.lognormal(0.0, 1.5, nnz)
    return p, i, x


cases = [  # name, m, n, density, kind, precision, chunk_cols
    ("u8", 300, 90, 0.05, "bytes", "auto", 32),
    ("u16_escape", 70000, 130, 0.002, "counts", "auto", 64),     # row gaps > 254 and values > 254: both escape paths
    ("f32", 257, 70, 0.08, "real", "fp32", 16),
    ("f16", 257, 70, 0.08, "real", "fp16", 16),
    ("quant8", 199, 65, 0.1, "real", "quant8", 32),
    ("f64", 120, 40, 0.1, "real", "fp64", 2048),                # single chunk
    ("one_col", 50, 1, 0.3, "bytes", "auto", 256),
    # chunks without nonzeros: the reference encoder omits the 4-byte count-section size for them (sparsepress_v2.hpp:94)
    # while its decoder expects it (:988-991), so the decoded column pointers of such a chunk are whatever the following
    # bytes spell.  The fixture records what the reference decoder returns; a faithful decoder returns the same.
    ("empty_chunk_quirk", 40, 9, 0.0, "bytes", "auto", 4),
    # row_sort = TRUE (st_convert's default, R/streampress.R:306-309): rows renumbered by the encoder, the permutation
    # stored in the metadata section and applied by the decoder (sparsepress_v2.hpp:1089-1103) -- the fixture records
    # what the reference decoder returns for what the reference encoder wrote
    ("rowsort_u8", 300, 90, 0.05, "bytes", "auto", 32),
    ("rowsort_f32", 257, 70, 0.08, "real", "fp32", 16),
]
out = {}
names = []
for name, m, n, dens, kind, prec, cc in cases:
    p, i, x = random_csc(m, n, dens, seed=len(names) + 11, kind=kind)
    buf = ref_encode(m, n, p, i, x, prec, cc, row_sort=name.startswith("rowsort"))
    dec = ref_decode(buf)
    assert dec["m"] == m and dec["n"] == n
    assert dec["row_sorted"] == (1 if name.startswith("rowsort") else 0)
    if name.startswith("rowsort"):
        assert np.array_equal(dec["p"], p)
        print("   row-sorted round trip returns the input rows:", bool(np.array_equal(dec["i"], i)),
              "| as sets per column:", all(set(dec["i"][p[j]:p[j + 1]]) == set(i[p[j]:p[j + 1]]) for j in range(n)))
    elif name != "empty_chunk_quirk":
        assert np.array_equal(dec["p"], p) and np.array_equal(dec["i"], i)
    if prec in ("auto", "fp64"):
        assert np.array_equal(dec["x"], x)
    out[name + "_spz"] = buf
    out[name + "_p"], out[name + "_i"], out[name + "_x"] = dec["p"], dec["i"], dec["x"]
    out[name + "_info"] = np.array([m, n, dec["nnz"], dec["vt"], dec["chunks"]], np.int64)
    names.append(name)
    print("%-12s %6d x %-4d nnz %-7d vt %d chunks %-3d %d bytes" % (name, m, n, dec["nnz"], dec["vt"], dec["chunks"], buf.size))
out["names"] = np.array(names)

src = "/root/reference/inst/extdata/pbmc3k.spz"
dst = os.path.join(here, "pbmc3k.spz")
shutil.copyfile(src, dst)
buf = np.fromfile(dst, np.uint8)
dec = ref_decode(buf)
h = lambda a: np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)
out["pbmc3k_info"] = np.array([dec["m"], dec["n"], dec["nnz"], dec["vt"], dec["chunks"], dec["row_sorted"]], np.int64)
out["pbmc3k_sha_p"], out["pbmc3k_sha_i"], out["pbmc3k_sha_x"] = h(dec["p"]), h(dec["i"]), h(dec["x"])
out["pbmc3k_head_i"], out["pbmc3k_head_x"], out["pbmc3k_sum_x"] = dec["i"][:64], dec["x"][:64], np.array([dec["x"].sum()])
print("pbmc3k", out["pbmc3k_info"])
np.savez_compressed(os.path.join(here, "spz_vectors.npz"), **out)
