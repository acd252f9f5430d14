"""Inputs for the ALS-NNLS path: CSC container, factor initialisation, synthetic matrices.

* `splitmix64_uniform` / `init_factors` reproduce the reference's random initialisation stream
  (inst/include/FactorNet/rng/rng.hpp:60-104,194-201; nmf/nmf_init.hpp:166-182) in vectorised
  numpy, at any offset of the stream (ranks of a sharded run draw their own slice of H).
* `simulate_nmf_sparse` restates R/simulateNMF.R:26-70 (block-structured w, h; additive Gaussian
  noise scaled to the mean signal; clamp at 0) but only evaluates the entries that are kept by the
  Bernoulli sampling mask, so 20 000 x 100 000 at 1 % never materialises 2e9 dense entries.
  Random numbers come from numpy (CPU) or torch (GPU) -- not R's RNG; the generator defines the
  workload shape, parity always compares oracle and GPU on the SAME generated arrays.
"""
import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


class CSC:
    """Host CSC matrix: shape (rows, cols), int32 `p` (cols+1), int32 `i` (nnz, sorted per column), float64 `x`."""

    def __init__(self, shape, p, i, x):
        self.rows, self.cols = int(shape[0]), int(shape[1])
        self.p = np.ascontiguousarray(p, dtype=np.int32)
        self.i = np.ascontiguousarray(i, dtype=np.int32)
        self.x = np.ascontiguousarray(x, dtype=np.float64)
        if self.p.shape[0] != self.cols + 1 or self.p[-1] != self.i.shape[0] or self.i.shape[0] != self.x.shape[0]:
            raise ValueError("inconsistent CSC arrays")

    @property
    def nnz(self):
        return int(self.x.shape[0])

    @property
    def shape(self):
        return (self.rows, self.cols)

    def transpose(self):
        """CSC of A^T (rows sorted within each column): stable sort of the entries by row."""
        counts = np.diff(self.p)
        cols = np.repeat(np.arange(self.cols, dtype=np.int32), counts)
        order = np.argsort(self.i, kind="stable")
        tp = np.zeros(self.rows + 1, np.int64)
        np.cumsum(np.bincount(self.i, minlength=self.rows), out=tp[1:])
        return CSC((self.cols, self.rows), tp.astype(np.int32), cols[order], self.x[order])

    def col_slice(self, c0, c1):
        s, e = int(self.p[c0]), int(self.p[c1])
        return CSC((self.rows, c1 - c0), self.p[c0:c1 + 1] - self.p[c0], self.i[s:e], self.x[s:e])

    @staticmethod
    def from_scipy(a):
        a = a.tocsc()
        a.sort_indices()
        return CSC(a.shape, a.indptr, a.indices, a.data)

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csc_matrix((self.x, self.i, self.p), shape=self.shape)


def splitmix64_raw(seed, offset, count):
    """`count` outputs of SplitMix64(seed) starting at stream position `offset` (0-based), as uint64."""
    seed = np.uint64(12345 if int(seed) == 0 else int(seed))
    with np.errstate(over="ignore"):
        idx = np.arange(int(offset) + 1, int(offset) + int(count) + 1, dtype=np.uint64)
        z = seed + idx * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def splitmix64_uniform(seed, offset, count, dtype=np.float64):
    """uniform<T>() = T(next()) / T(UINT64_MAX)  (rng.hpp:100-104); T(UINT64_MAX) rounds to 2^64."""
    z = splitmix64_raw(seed, offset, count)
    dtype = np.dtype(dtype)
    return (z.astype(dtype) / dtype.type(2.0 ** 64)).astype(dtype)


def init_factors(seed, k, m, n, dtype=np.float64, col_offset=0, n_total=None):
    """initialize_factors (nmf_init.hpp:166-182): one stream fills W_T (k x m) then H (k x n_total), column-major.
    Returns W_T as (m, k) and the H slice for columns [col_offset, col_offset + n) as (n, k) arrays."""
    W_T = splitmix64_uniform(seed, 0, k * m, dtype).reshape(m, k)
    H = splitmix64_uniform(seed, k * m + k * col_offset, k * n, dtype).reshape(n, k)
    return W_T, H


# --------------------------------------------------------------------------------------------
# simulateNMF restatement
# --------------------------------------------------------------------------------------------
def _sim_factors(nrow, ncol, k, rng):
    """Block-structured w (nrow x k) and h (k x ncol), normalised (R/simulateNMF.R:29-55)."""
    w = np.zeros((nrow, k))
    bw = nrow // k
    for f in range(k):
        s, e = f * bw, (nrow if f == k - 1 else (f + 1) * bw)
        w[s:e, f] = np.abs(rng.normal(1.0, 0.3, size=e - s))
    w += np.abs(rng.normal(0.0, 0.05, size=(nrow, k)))
    h = np.zeros((k, ncol))
    bh = ncol // k
    for f in range(k):
        s, e = f * bh, (ncol if f == k - 1 else (f + 1) * bh)
        h[f, s:e] = np.abs(rng.normal(1.0, 0.3, size=e - s))
    h += np.abs(rng.normal(0.0, 0.05, size=(k, ncol)))
    w /= w.sum(axis=0, keepdims=True)
    h /= h.sum(axis=1, keepdims=True)
    return w, h


def simulate_nmf_sparse(nrow, ncol, k, density, noise=0.5, seed=123, device=None, col_offset=0, ncol_total=None,
                        drop_zeros=True):
    """Sparse sample of simulateNMF(nrow, ncol_total, k, noise): entries kept with probability `density`
    (the Bernoulli `dropout` mask of R/simulateNMF.R:66-69), values (w h + N(0, noise*mean))_+ evaluated only
    there.  Entries clamped to exactly 0 are dropped (a dgCMatrix does not store them) unless drop_zeros=False.
    With `device` (a torch cuda device) sampling and evaluation run on the GPU.  Columns
    [col_offset, col_offset+ncol) of the ncol_total-wide matrix are generated (column shard of a rank).
    Returns (CSC, w, h_local)."""
    ncol_total = ncol_total or ncol
    rng = np.random.default_rng(seed)
    w, h = _sim_factors(nrow, ncol_total, k, rng)
    h = h[:, col_offset:col_offset + ncol]
    mean_sig = k / (float(nrow) * float(ncol_total))      # columns of w and rows of h sum to 1
    sd = noise * mean_sig
    total = int(nrow) * int(ncol)
    if device is not None:
        import torch
        g = torch.Generator(device=device)
        g.manual_seed(int(seed) * 7919 + int(col_offset))
        tw = torch.from_numpy(w).to(device)
        th = torch.from_numpy(np.ascontiguousarray(h.T)).to(device)          # (ncol, k)
        cnt = int(rng.binomial(total, density))
        lin = torch.randint(0, total, (int(cnt * 1.02) + 16,), generator=g, device=device, dtype=torch.int64)
        lin = torch.unique(lin)                                              # sorted, duplicates removed
        if lin.numel() > cnt:
            keep = torch.randperm(lin.numel(), generator=g, device=device)[:cnt]
            lin = lin[torch.sort(keep).values]
        cols = torch.div(lin, nrow, rounding_mode="floor")
        rows = lin - cols * nrow
        vals = torch.empty(lin.numel(), dtype=torch.float64, device=device)
        CH = 1 << 20
        for s in range(0, lin.numel(), CH):
            e = min(s + CH, lin.numel())
            vals[s:e] = (tw[rows[s:e]] * th[cols[s:e]]).sum(dim=1)
        if noise > 0:
            vals += torch.randn(vals.shape, generator=g, device=device, dtype=torch.float64) * sd
            vals.clamp_(min=0.0)
        if drop_zeros:
            nz = vals > 0
            rows, cols, vals = rows[nz], cols[nz], vals[nz]
        counts = torch.bincount(cols, minlength=ncol)
        p = torch.zeros(ncol + 1, dtype=torch.int64, device=device)
        p[1:] = torch.cumsum(counts, 0)
        A = CSC((nrow, ncol), p.cpu().numpy(), rows.cpu().numpy(), vals.cpu().numpy())
        return A, w, h
    cnt = int(rng.binomial(total, density))
    lin = np.unique(rng.integers(0, total, size=int(cnt * 1.02) + 16, dtype=np.int64))
    if lin.shape[0] > cnt:
        lin = np.sort(rng.choice(lin, size=cnt, replace=False))
    cols = lin // nrow
    rows = lin - cols * nrow
    vals = np.empty(lin.shape[0])
    CH = 1 << 20
    hT = np.ascontiguousarray(h.T)
    for s in range(0, lin.shape[0], CH):
        e = min(s + CH, lin.shape[0])
        vals[s:e] = np.einsum("ij,ij->i", w[rows[s:e]], hT[cols[s:e]])
    if noise > 0:
        vals += rng.normal(0.0, sd, size=vals.shape[0])
        np.maximum(vals, 0.0, out=vals)
    if drop_zeros:
        nz = vals > 0
        rows, cols, vals = rows[nz], cols[nz], vals[nz]
    p = np.zeros(ncol + 1, np.int64)
    np.cumsum(np.bincount(cols, minlength=ncol), out=p[1:])
    return CSC((nrow, ncol), p, rows, vals), w, h


def simulate_nmf_sparse_shards(nrow, ncol_total, k, density, shards, seed=123, device=None, round_f32=False):
    """The ncol_total-wide simulateNMF sample built shard by shard (column shard r from its own generator stream, exactly the
    matrix the `shards` ranks of a column-sharded run hold between them) and concatenated into ONE host CSC -- how a matrix too
    large to sample in one piece (BASELINE configs[3]: 30 000 x 1 300 000, 1.17e9 nonzeros) is generated.  round_f32: values
    rounded to fp32 (so that fp32 and fp64 consumers see the same numbers)."""
    if ncol_total % shards:
        raise ValueError("ncol_total must be a multiple of shards")
    nsh = ncol_total // shards
    ps, iis, xs, off = [np.zeros(1, np.int64)], [], [], 0
    for r in range(shards):
        a = simulate_nmf_sparse(nrow, nsh, k, density, seed=seed, device=device, col_offset=r * nsh, ncol_total=ncol_total)[0]
        ps.append(a.p[1:].astype(np.int64) + off)
        iis.append(a.i)
        xs.append(a.x.astype(np.float32).astype(np.float64) if round_f32 else a.x)
        off += a.nnz
    if off >= 2 ** 31:
        raise ValueError("more than 2^31 - 1 nonzeros: the CSC boundary's int col_ptr cannot hold them")
    return CSC((nrow, ncol_total), np.concatenate(ps), np.concatenate(iis), np.concatenate(xs))


def simulate_nb_counts(nrow, ncol, k, density=0.02, size=5.0, seed=123, scale=None):
    """NB counts for BASELINE config C5 (structure of tests/testthat/test_nb_nmf.R:11-27): mu = (w h) * scale,
    y ~ NegBin(size, mu) at Bernoulli(density)-sampled positions; zeros dropped."""
    rng = np.random.default_rng(seed)
    w, h = _sim_factors(nrow, ncol, k, rng)
    total = int(nrow) * int(ncol)
    cnt = int(rng.binomial(total, density))
    lin = np.unique(rng.integers(0, total, size=int(cnt * 1.02) + 16, dtype=np.int64))
    cols = lin // nrow
    rows = lin - cols * nrow
    hT = np.ascontiguousarray(h.T)
    mu = np.einsum("ij,ij->i", w[rows], hT[cols])
    if scale is None:
        scale = 5.0 / mu.mean()
    mu = mu * scale
    lam = rng.gamma(shape=size, scale=mu / size)
    y = rng.poisson(lam).astype(np.float64)
    nz = y > 0
    rows, cols, y = rows[nz], cols[nz], y[nz]
    p = np.zeros(ncol + 1, np.int64)
    np.cumsum(np.bincount(cols, minlength=ncol), out=p[1:])
    return CSC((nrow, ncol), p, rows, y), w, h


# --------------------------------------------------------------------------------------------
# R's default RNG (Mersenne-Twister, inversion) -- `set.seed(seed); runif(n)`.
# nmf(seed = <int>) draws W_init this way (reference R/nmf_thin.R:790-797).  R itself is not part of
# the reference tree and cannot run here, so this restates R's public algorithm (src/main/RNG.c:
# 50 + 625 steps of the LCG 69069*s+1, MT19937, scaling by 2.3283064365386963e-10, fixup into (0,1));
# pinned in tests/ against widely published values (set.seed(42); runif(3) = 0.9148060 0.9370754 0.2861395).
# --------------------------------------------------------------------------------------------
def r_runif(seed, n):
    s = np.uint32(int(seed) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        for _ in range(50):
            s = np.uint32(69069) * s + np.uint32(1)
        key = np.empty(625, np.uint32)
        for j in range(625):
            s = np.uint32(69069) * s + np.uint32(1)
            key[j] = s
    bg = np.random.MT19937()
    st = bg.state
    st["state"]["key"] = key[1:].copy()     # i_seed[0] is the `mti` slot, forced to 624 by FixupSeeds
    st["state"]["pos"] = 624
    bg.state = st
    u = bg.random_raw(int(n)).astype(np.float64) * 2.3283064365386963e-10
    lo = 2.328306437080797e-10
    u = np.where(u <= 0.0, 0.5 * lo, u)
    u = np.where(1.0 - u <= 0.0, 1.0 - 0.5 * lo, u)
    return u
