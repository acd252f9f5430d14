"""Shared helpers for the test-suite (fixtures, small synthetic problems, comparisons)."""
import os

import numpy as np

from oracle import oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fixture(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    return O.Csc(d["shape"], d["p"], d["i"], d["x"])


def random_csc(m, n, density, seed, values="uniform"):
    """Small random nonnegative CSC (sorted rows), numpy RNG."""
    rng = np.random.default_rng(seed)
    p = [0]
    ii, xx = [], []
    for j in range(n):
        cnt = rng.binomial(m, density)
        rows = np.sort(rng.choice(m, size=cnt, replace=False)).astype(np.int32)
        if values == "uniform":
            v = rng.uniform(0.1, 1.0, size=cnt)
        else:
            v = rng.poisson(3.0, size=cnt).astype(np.float64) + 1.0
        ii.append(rows)
        xx.append(v)
        p.append(p[-1] + cnt)
    return O.Csc((m, n), np.asarray(p, np.int32), np.concatenate(ii) if ii else np.zeros(0, np.int32),
                 np.concatenate(xx) if xx else np.zeros(0))


def lowrank_csc(m, n, k, density, seed, noise=0.05):
    """Sparse sample of a noisy nonnegative rank-k matrix."""
    rng = np.random.default_rng(seed)
    w = rng.gamma(1.0, 1.0, size=(m, k))
    h = rng.gamma(1.0, 1.0, size=(k, n))
    p = [0]
    ii, xx = [], []
    for j in range(n):
        cnt = max(1, rng.binomial(m, density))
        rows = np.sort(rng.choice(m, size=cnt, replace=False)).astype(np.int32)
        v = w[rows] @ h[:, j]
        v = np.maximum(v + noise * rng.standard_normal(cnt) * v.mean(), 1e-3)
        ii.append(rows)
        xx.append(v)
        p.append(p[-1] + cnt)
    return O.Csc((m, n), np.asarray(p, np.int32), np.concatenate(ii), np.concatenate(xx))


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))
