"""Column-sharded alternating-NNLS loop on device-resident data (one process per GPU).

The loop is the reference's nmf_fit<CPU> (inst/include/FactorNet/nmf/fit_cpu.hpp:444-1855, MSE, no
mask) restated over the device-level C-ABI ops, with the one exchange step the path has when the
columns of A (and of H) are sharded over ranks:

    H side  : every rank holds all of W_T; columns of A are independent -> no communication.
    scaling : row L1 (or squared L2) sums of H are sums over ALL columns, and so are
    W side  : H H^T and H A^T -> each rank forms, from its shard and the UNSCALED H, its partial k x k Gram, k x m right-hand
              side and k row sums, and ONE all-reduce carries [G_raw | B_raw | rowsums] (fused buffer); the scaling
              D = diag(d) is applied afterwards, locally: G = D^-1 G_raw D^-1 + eps I, B = B_raw D^-1, H_loc <- H_loc D^-1 --
              algebraically the reference's "normalise, then multiply", different rounding (inside the N > 1 tolerance).
              The m columns of W_T are then solved in row blocks, one per rank, and one all-gather replicates W_T again.

For world_size == 1 the same code runs with the collectives skipped, so N = 1 and N > 1 share every
kernel launch.  The compute backend is an `ops` object; the product backend is `HipOps` (HIP kernels
through rcppml_amd._abi).  Nothing here falls back to PyTorch or CPU math.
"""
from dataclasses import dataclass

import os

import numpy as np


ORDER_MIN_COLUMNS = 16384   # sides with at least this many columns solve them in sweep-sorted order (plugin: kOrderMinColumns)

@dataclass
class AlsConfig:
    k: int
    max_iter: int = 100
    tol: float = 1e-4
    L1_W: float = 0.0
    L1_H: float = 0.0
    L2_W: float = 0.0
    L2_H: float = 0.0
    ub_W: float = 0.0
    ub_H: float = 0.0
    cd_maxit: int = 100
    cd_tol: float = 1e-8
    patience: int = 5
    nonneg_W: bool = True
    nonneg_H: bool = True
    norm_type: int = 0          # 0 = L1, 1 = L2, 2 = none
    solver_mode: int = 0        # 0 = CD, 1 = Cholesky + clip
    cd_variant: int = 0         # 0 = auto (fp32: MFMA tiles, fp64 k <= 64: 16-column MFMA tiles, else lane-group), 1 = lane, 2 = wave, 5 = lane-group
    order_columns: bool = True  # schedule CD columns by the sweep counts of the previous iteration
    # world > 1, W half-update: "block" = every rank solves its block of W's rows, one all-gather (the solve shrinks with the
    # world size); "replicated" = every rank solves all of W from the all-reduced (G, B) and nothing is gathered (what the
    # plugin's RCPPML_GPU_DEVICES path does by default).  Same numbers either way; which is faster over xGMI is a
    # measurement the first 8-GPU run decides (unmeasured on hardware so far).
    w_solve: str = "block"


class HipOps:
    """Device-level ops on torch CUDA tensors (allocation/stream plumbing only) via the C-ABI."""

    def __init__(self, device, dtype="f32", record_events=False):
        import torch
        from . import _abi
        if not torch.cuda.is_available():
            raise _abi.BackendError("HipOps needs a GPU: torch.cuda.is_available() is False and there is no CPU fallback")
        self.torch = torch
        self._abi = _abi
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.dt = _abi.F32 if dtype == "f32" else _abi.F64
        self.tdtype = torch.float32 if dtype == "f32" else torch.float64
        self.ndtype = np.float32 if dtype == "f32" else np.float64
        self.ctx = _abi.Context(device)
        self.record = record_events
        self.events = {}
        self._order = {}
        self.fused_tail = True          # scale_order() / gram_loss_mse(): independent kernels share launches (False: the separate ops, for A/B runs and tests)
        self.keep_order_used = False    # solve() keeps a copy of the work order it ran in (bench.py's idle-slot figure)

    # -- plumbing
    def to_device(self, a, dtype=None):
        t = self.torch.from_numpy(np.ascontiguousarray(a))
        if dtype is not None:
            t = t.to(dtype)
        return t.to(self.device)

    def empty(self, shape, dtype=None):
        return self.torch.empty(shape, dtype=dtype or self.tdtype, device=self.device)

    def zeros(self, shape, dtype=None):
        return self.torch.zeros(shape, dtype=dtype or self.tdtype, device=self.device)

    def upload_csc(self, A):
        return dict(rows=A.rows, cols=A.cols, nnz=A.nnz, p=self.to_device(A.p), i=self.to_device(A.i),
                    x=self.to_device(A.x, self.tdtype))

    def transpose_csc(self, csc):
        """CSC of the transpose, on the device (stable sort of the nonzeros by row: rows stay sorted inside every column)."""
        t = self.torch
        out = dict(rows=csc["cols"], cols=csc["rows"], nnz=csc["nnz"], p=self.empty((csc["rows"] + 1,), t.int32),
                   i=self.empty((max(csc["nnz"], 1),), t.int32), x=self.empty((max(csc["nnz"], 1),)))
        self.ctx.transpose_csc(self.dt, csc["rows"], csc["cols"], csc["p"], csc["i"], csc["x"], out["p"], out["i"], out["x"])
        return out

    def _timed(self, name):
        return _EventScope(self, name) if self.record else _NULL

    def reset_events(self):
        self.events = {}

    def event_ms(self):
        """name -> (count, total ms); call after a device synchronize."""
        return {k: (len(v), float(sum(s.elapsed_time(e) for s, e in v))) for k, v in self.events.items()}

    # -- ops
    def gram(self, F, eps, l2, out=None, tag="gram"):
        r, k = F.shape
        G = out if out is not None else self.empty((k, k))
        with self._timed(tag):
            self.ctx.gram(self.dt, F, k, r, eps, l2, G)
        return G

    def rhs(self, csc, F, out=None, tag="rhs"):
        k = F.shape[1]
        B = out if out is not None else self.empty((csc["cols"], k))
        plan = csc.get("plans", {}).get(k)
        with self._timed(tag):
            if plan is not None:
                self.ctx.rhs_planned(plan, F, B)
            else:
                self.ctx.rhs(self.dt, csc["p"], csc["i"], csc["x"], csc["cols"], F, k, B)
        return B

    def plan_rhs(self, csc, k, min_nnz=1 << 20, partitions=0, slots=0):
        """Build the LDS row-tiled plan of csc for rank k (large inputs only; kept in csc["plans"], used by rhs())."""
        if os.environ.get("RCPPML_GPU_RHS_TILED", "1") == "0" or csc["nnz"] < min_nnz:
            return None
        try:
            plan = self.ctx.rhs_plan(self.dt, csc["p"], csc["i"], csc["x"], csc["cols"], csc["rows"], k, partitions, slots)
        except self._abi.BackendError:      # the plan is an optimisation: without it the products run on the gather kernel
            plan = None
        if plan is not None:
            csc.setdefault("plans", {})[k] = plan
        return plan

    def solve(self, G, B, X, cfg, side, warm, tag="solve"):
        n, k = X.shape
        l1 = cfg.L1_H if side == "H" else cfg.L1_W
        ub = cfg.ub_H if side == "H" else cfg.ub_W
        nonneg = cfg.nonneg_H if side == "H" else cfg.nonneg_W
        with self._timed(tag):
            if cfg.solver_mode == 0:
                # work order: columns sorted by the sweeps they needed in the previous ALS iteration (a wave runs
                # until its slowest column converges); per-column results do not depend on the order
                st = self._order.get(side) if cfg.order_columns else None
                if cfg.order_columns and (st is None or st["sweeps"].shape[0] != n):
                    st = dict(sweeps=self.zeros((n,), self.torch.int32), order=self.empty((n,), self.torch.int32), valid=False)
                    self._order[side] = st
                # (worth its three small kernels from one 16-column wavefront per SIMD: C2's 20 000-column W side, whose
                # slowest columns set the kernel's time, goes 0.350 -> 0.313 ms)
                use = st is not None and st["valid"] and cfg.cd_tol > 0 and n >= ORDER_MIN_COLUMNS
                if use and not st.get("fresh", False):     # (scale_order() already ranked these sweep counts: fused tail)
                    self.ctx.order_columns(st["sweeps"], n, st["order"])
                if use and self.keep_order_used:            # bench.py's idle-slot figure wants the order THIS solve ran in
                    st["order_used"] = st["order"].clone()
                self.ctx.solve_cd(self.dt, G, B, X, k, n, l1_pre=l1 if l1 > 0 else 0.0, warm=int(warm), zero_init=0,
                                  nonneg=int(nonneg), maxit=cfg.cd_maxit, tol=cfg.cd_tol, ub_post=ub,
                                  variant=cfg.cd_variant, sweeps_out=st["sweeps"] if st is not None else None,
                                  col_order=st["order"] if use else None)
                if st is not None:
                    st["valid"] = True
                    st["fresh"] = False                     # new sweep counts: the order buffer is one solve behind again
            else:
                self.ctx.solve_chol(self.dt, G, B, X, k, n, l1_pre=l1 if l1 > 0 else 0.0, nonneg=int(nonneg), ub_post=ub)

    def row_norms(self, X, norm_type, out=None):
        n, k = X.shape
        s = out if out is not None else self.empty((k,))
        with self._timed("scale"):
            self.ctx.row_norms(self.dt, X, k, n, norm_type, s)
        return s

    def apply_scaling(self, X, sums, norm_type, d):
        n, k = X.shape
        with self._timed("scale"):
            self.ctx.apply_scaling(self.dt, X, k, n, norm_type, sums, d)

    def scale_order(self, X, sums, norm_type, d, cfg, side):
        """extract_scaling of X (row_norms + apply_scaling) and, when the next solve of `side` will run in sweep-sorted work
        order, that order from the sweep counts the solve just left -- three launches instead of five (rcppml_hip_scale_order;
        the results of the separate calls bit for bit).  Falls back to the separate calls when fused_tail is off."""
        n, k = X.shape
        st = self._order.get(side) if (cfg.order_columns and cfg.solver_mode == 0) else None
        rank = (st is not None and st["sweeps"].shape[0] == n and st["valid"] and cfg.cd_tol > 0 and n >= ORDER_MIN_COLUMNS)
        if not self.fused_tail:
            self.row_norms(X, norm_type, out=sums)
            self.apply_scaling(X, sums, norm_type, d)
            return
        with self._timed("scale"):
            self.ctx.scale_order(self.dt, X, k, n, norm_type, sums, d, st["sweeps"] if rank else None, st["order"] if rank else None)
        if rank:
            st["fresh"] = True

    def _rank_state(self, X, cfg, side):
        n = X.shape[0]
        st = self._order.get(side) if (cfg.order_columns and cfg.solver_mode == 0) else None
        rank = (st is not None and st["sweeps"].shape[0] == n and st["valid"] and cfg.cd_tol > 0 and n >= ORDER_MIN_COLUMNS)
        return st, rank

    def tail_scale_gram(self, X, sums, norm_type, d, cfg, side, eps, G):
        """The H side's tail in one call: scale_order(X), then gram(X, eps, 0) -> G (rcppml_hip_tail_scale_gram)."""
        if not self.fused_tail:
            self.scale_order(X, sums, norm_type, d, cfg, side)
            self.gram(X, eps, 0.0, out=G, tag="gram")
            return
        n, k = X.shape
        st, rank = self._rank_state(X, cfg, side)
        with self._timed("scale_gram"):
            self.ctx.tail_scale_gram(self.dt, X, k, n, norm_type, sums, d, st["sweeps"] if rank else None, st["order"] if rank else None, eps, 0.0, G)
        if rank:
            st["fresh"] = True

    def tail_scale_gram_loss(self, W_T, sums, norm_type, d, cfg, side, eps, trAtA, B_w, G_saved, G_wt, out):
        """The W side's tail in one call: scale_order(W_T), then gram_loss_mse (rcppml_hip_tail_scale_gram_loss)."""
        if not self.fused_tail:
            self.scale_order(W_T, sums, norm_type, d, cfg, side)
            self.gram_loss_mse(W_T, eps, trAtA, d, B_w, G_saved, G_wt, out)
            return
        n, k = W_T.shape
        st, rank = self._rank_state(W_T, cfg, side)
        with self._timed("scale_gram_loss"):
            self.ctx.tail_scale_gram_loss(self.dt, W_T, k, n, norm_type, sums, d, st["sweeps"] if rank else None, st["order"] if rank else None,
                                          eps, trAtA, B_w, G_saved, G_wt, out)
        if rank:
            st["fresh"] = True

    def gram_loss_mse(self, W_T, eps, trAtA, d, B_w, G_saved, G_wt, out):
        """G_wt = gram(W_T) + eps I, then the Gram-trick loss with it: three launches instead of four (rcppml_hip_gram_loss_mse)."""
        n, k = W_T.shape
        if not self.fused_tail:
            self.gram(W_T, eps, 0.0, out=G_wt, tag="gram")
            self.loss_mse(trAtA, d, W_T, B_w, G_wt, G_saved, out)
            return
        with self._timed("loss"):
            self.ctx.gram_loss_mse(self.dt, W_T, k, n, eps, trAtA, d, B_w, G_saved, G_wt, out)

    def sumsq(self, x):
        out = self.empty((1,), self.torch.float64)
        self.ctx.sumsq(self.dt, x, x.numel(), out)
        return out

    def loss_mse(self, trAtA, d, W_T, B_w, G_wt, G_saved, out):
        m, k = W_T.shape
        with self._timed("loss"):
            self.ctx.loss_mse(self.dt, trAtA, d, W_T, B_w, k, m, G_wt, G_saved, out)

    def add_diag(self, G, v):
        if v != 0:
            G.diagonal().add_(v)        # k scalars; plumbing-level torch op

    def sync(self):
        self.torch.cuda.synchronize(self.device)


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NULL = _Null()


class _EventScope:
    def __init__(self, ops, name):
        self.ops, self.name = ops, name

    def __enter__(self):
        t = self.ops.torch
        self.s = t.cuda.Event(enable_timing=True)
        self.e = t.cuda.Event(enable_timing=True)
        self.s.record()
        return self

    def __exit__(self, *a):
        self.e.record()
        self.ops.events.setdefault(self.name, []).append((self.s, self.e))
        return False


class _CommScope:
    def __init__(self, comm, name, torch, nbytes):
        self.comm, self.name, self.torch, self.nbytes = comm, name, torch, nbytes

    def __enter__(self):
        self.s = self.torch.cuda.Event(enable_timing=True)
        self.e = self.torch.cuda.Event(enable_timing=True)
        self.s.record()

    def __exit__(self, *a):
        self.e.record()
        self.comm.events.setdefault(self.name, []).append((self.s, self.e, self.nbytes))


class Comm:
    """Collectives of the path.  world_size 1: no-ops -- unless `force` (a process group of ONE rank): then the sharded branch
    of step() runs and every collective is really issued (a one-rank sum is the identity), which executes the RCCL wiring
    and measures its per-call floor on a one-GPU box (bench.py RCPPML_BENCH_FORCE_DIST=1)."""

    def __init__(self, dist=None, time_collectives=False, force=False):
        self.dist = dist
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        self.sharded = self.world > 1 or (bool(force) and dist is not None)
        self.timed = bool(time_collectives) and self.sharded
        self.events = {}

    def _scope(self, name, t):
        """Device-side timing of one collective (events on the current stream; GPU tensors only)."""
        if not (self.timed and t.is_cuda):
            return _NULL
        import torch
        return _CommScope(self, name, torch, t.numel() * t.element_size())

    def reset_events(self):
        self.events = {}

    def collective_ms(self):
        """name -> dict(count, total_ms, bytes) of the timed collectives; call after a device synchronize."""
        return {k: dict(count=len(v), total_ms=float(sum(s.elapsed_time(e) for s, e, _ in v)), bytes=int(v[0][2]) if v else 0)
                for k, v in self.events.items()}

    def all_reduce_sum(self, t, tag="all_reduce"):
        if self.sharded:
            with self._scope(tag, t):
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t

    def all_gather_rows(self, full, rows_per, tag="all_gather_W"):
        """In-place all-gather of equal row blocks: rank r's block full[r*rows_per:(r+1)*rows_per] goes to every rank."""
        if self.sharded:
            mine = full[self.rank * rows_per:(self.rank + 1) * rows_per]
            with self._scope(tag, full):
                if self.dist.get_backend() == "gloo":          # CPU tests: gloo has no in-place tensor all-gather
                    parts = [full[r * rows_per:(r + 1) * rows_per] for r in range(self.world)]
                    self.dist.all_gather(parts, mine.clone())
                else:
                    self.dist.all_gather_into_tensor(full, mine)
        return full

    def barrier(self):
        if self.sharded:
            self.dist.barrier()


class ShardedALS:
    """State of one rank: its column shard A_loc (m x n_loc), A_loc^T, H_loc (n_loc x k), replicated W_T (m x k)."""

    def __init__(self, ops, comm, A_loc, At_loc, W_T0, H0, cfg):
        if cfg.w_solve not in ("block", "replicated"):
            raise ValueError("AlsConfig.w_solve must be 'block' or 'replicated'")
        self.ops, self.comm, self.cfg = ops, comm, cfg
        self.m, self.n_loc, self.k = A_loc.rows, A_loc.cols, cfg.k
        self.A = ops.upload_csc(A_loc)
        # At_loc = None: A^T is built on the device from the uploaded CSC (rcppml_hip_transpose_csc, the plugin's own set-up
        # path) -- a host-side transpose of 1e9 nonzeros takes minutes
        self.At = ops.upload_csc(At_loc) if At_loc is not None else ops.transpose_csc(self.A)
        k, m = self.k, self.m
        ops.plan_rhs(self.A, k)
        ops.plan_rhs(self.At, k)
        # W_T is replicated; for world > 1 its m columns (rows of the (m, k) array) are SOLVED in contiguous blocks of
        # rows_per per rank and all-gathered, so the W solve shrinks with the world size instead of being repeated.
        # rows_per is a multiple of 4 so every block starts 16-byte aligned for any k; the pad rows stay zero.
        self.rows_per = ((m + comm.world - 1) // comm.world + 3) // 4 * 4 if comm.sharded else m
        self.W_pad = ops.zeros((self.rows_per * comm.world, k))
        self.W_T = self.W_pad[:m]
        self.W_T.copy_(ops.to_device(W_T0, ops.tdtype))
        self.row_lo = min(m, comm.rank * self.rows_per)
        self.row_hi = min(m, self.row_lo + self.rows_per)
        self.H = ops.to_device(H0, ops.tdtype)
        self.d = ops.zeros((k,)) + 1
        self.Bh = ops.empty((self.n_loc, k))
        # fused exchange buffer [G_p (k*k) | B_p (m*k) | row sums of H (k)]: ONE all-reduce per iteration
        self.xbuf = ops.empty((k * k + m * k + k,))
        self.Gp = self.xbuf[:k * k].view(k, k)
        self.Bw = self.xbuf[k * k:k * k + m * k].view(m, k)
        self.xsums = self.xbuf[k * k + m * k:]
        self.d_tmp = ops.empty((k,))
        self.G = ops.empty((k, k))
        self.G_saved = ops.empty((k, k))
        self._gwt_of_current_w = False    # G_wt holds the Gram of the current W_T (set by step(); cleared by set_factors())
        self.G_wt = ops.empty((k, k))
        self.sums = ops.empty((k,))
        self.loss_out = ops.zeros((4,), ops.torch.float64)
        tr = ops.sumsq(self.A["x"])
        self.trAtA = comm.all_reduce_sum(tr)
        self.iter = 0
        self.eps = 1e-15

    def step(self):
        """One ALS iteration (H half-update, W half-update, loss).  Returns the device loss tensor."""
        ops, cfg, comm = self.ops, self.cfg, self.comm
        warm = self.iter > 0
        # ---- H half-update (fit_cpu.hpp:486-645)
        # W_T^T W_T + eps I: the previous iteration's loss formed exactly this Gram from exactly this W_T (same kernel, same
        # input: bitwise the same matrix), so it is reused when nothing is added to it
        if self.iter > 0 and cfg.L2_H == 0 and self._gwt_of_current_w:
            G_h = self.G_wt
        else:
            ops.gram(self.W_T, self.eps, cfg.L2_H, out=self.G, tag="gram")
            G_h = self.G
        empty = self.n_loc == 0          # a rank whose shard holds no column (more ranks than columns): it only takes part in the exchange
        if not empty:
            ops.rhs(self.A, self.W_T, out=self.Bh, tag="rhs_H")
            ops.solve(G_h, self.Bh, self.H, cfg, "H", warm, tag="solve_H")
        if comm.sharded:
            # ---- W half-update, sharded (fit_cpu.hpp:711-893): partial sums from the UNSCALED H, one all-reduce, then D^-1
            if empty:
                self.xbuf.zero_()
            else:
                ops.row_norms(self.H, cfg.norm_type, out=self.xsums)
                ops.gram(self.H, 0.0, 0.0, out=self.Gp, tag="gram")             # partial H_loc H_loc^T, eps after the sum
                ops.rhs(self.At, self.H, out=self.Bw, tag="rhs_W")
            comm.all_reduce_sum(self.xbuf, tag="all_reduce_gram_rhs_rowsums")
            if empty:
                ops.apply_scaling(self.Gp[:1].clone(), self.xsums, cfg.norm_type, self.d)   # no column of H here: only d = the global row norms (on a scratch row)
            else:
                ops.apply_scaling(self.H, self.xsums, cfg.norm_type, self.d)     # H_loc <- H_loc D^-1, d = global row norms
            ops.apply_scaling(self.Bw, self.xsums, cfg.norm_type, self.d_tmp)        # B = B_raw D^-1
            ops.apply_scaling(self.Gp, self.xsums, cfg.norm_type, self.d_tmp)        # G[:, g] /= d_g ...
            self.Gp.copy_(self.Gp.t().contiguous())                                   # ... (k x k, symmetric up to rounding order)
            ops.apply_scaling(self.Gp, self.xsums, cfg.norm_type, self.d_tmp)        # ... and G[f, :] /= d_f
            ops.add_diag(self.Gp, self.eps)
        else:
            # extract_scaling of H (+ the next H solve's work order), then -- W half-update (fit_cpu.hpp:711-893) -- its Gram
            ops.tail_scale_gram(self.H, self.sums, cfg.norm_type, self.d, cfg, "H", self.eps, self.Gp)
            ops.rhs(self.At, self.H, out=self.Bw, tag="rhs_W")
        # G_saved = Gram of H before L2 (:719-722), G = G_saved + L2_W I (:738).  Without an L2 penalty on W both are the
        # buffer the Gram kernel just wrote (it stays untouched until the next iteration's Gram of H): no copies
        if cfg.L2_W > 0:
            self.G_saved.copy_(self.Gp)
            self.G.copy_(self.Gp)
            ops.add_diag(self.G, cfg.L2_W)
            G_w, G_saved = self.G, self.G_saved
        else:
            G_w, G_saved = self.Gp, self.Gp
        if comm.sharded and cfg.w_solve == "block":
            if self.row_hi > self.row_lo:
                ops.solve(G_w, self.Bw[self.row_lo:self.row_hi], self.W_T[self.row_lo:self.row_hi], cfg, "W", warm, tag="solve_W")
            comm.all_gather_rows(self.W_pad, self.rows_per)
        else:
            ops.solve(G_w, self.Bw, self.W_T, cfg, "W", warm, tag="solve_W")
        if comm.sharded and cfg.w_solve == "block":
            # (a rank solved only its block of rows: their sweep counts do not describe the whole W_T this pass scales)
            ops.row_norms(self.W_T, cfg.norm_type, out=self.sums)
            ops.apply_scaling(self.W_T, self.sums, cfg.norm_type, self.d)
            # ---- loss (fit_cpu.hpp:1729-1753): B_w is the h_at of the reference's third sparse pass
            ops.gram_loss_mse(self.W_T, self.eps, self.trAtA, self.d, self.Bw, G_saved, self.G_wt, self.loss_out)
        else:
            ops.tail_scale_gram_loss(self.W_T, self.sums, cfg.norm_type, self.d, cfg, "W", self.eps, self.trAtA, self.Bw, G_saved, self.G_wt,
                                     self.loss_out)
        self._gwt_of_current_w = True
        self.iter += 1
        return self.loss_out

    def fit(self):
        """Full fit with the reference's convergence rule (fit_cpu.hpp:1769-1809).  Returns a dict."""
        cfg = self.cfg
        prev = float(np.finfo(self.ops.ndtype).max)
        patience_counter, converged, final_tol, history = 0, False, 0.0, []
        iterations = 0
        for it in range(cfg.max_iter):
            loss = float(self.step()[0].item())
            if self.ops.ndtype == np.float32:
                loss = float(np.float32(loss))
            history.append(loss)
            hit = False
            if it > 0:
                rel = abs(prev - loss) / (abs(prev) + 1e-15)
                final_tol = rel
                hit = rel < cfg.tol
            prev = loss
            if it > 0:
                if hit:
                    patience_counter += 1
                    if patience_counter >= cfg.patience:
                        converged = True
                        iterations = it + 1
                        break
                else:
                    patience_counter = 0
            iterations = it + 1
        return dict(iter=iterations, converged=converged, loss=prev, tol=final_tol, loss_history=history)

    def set_factors(self, W_T=None, H=None, d=None, iteration=None):
        """Overwrite factors from outside (device tensors or arrays of the same shapes).  Invalidates the loss Gram that step()
        would otherwise reuse as the next H-side Gram."""
        if W_T is not None:
            self.W_T.copy_(W_T if hasattr(W_T, "is_cuda") else self.ops.to_device(W_T, self.ops.tdtype))
            self._gwt_of_current_w = False
        if H is not None:
            self.H.copy_(H if hasattr(H, "is_cuda") else self.ops.to_device(H, self.ops.tdtype))
        if d is not None:
            self.d.copy_(d if hasattr(d, "is_cuda") else self.ops.to_device(d, self.ops.tdtype))
        if iteration is not None:
            self.iter = int(iteration)

    def factors(self):
        """(W_T (m,k), d (k), H_loc (n_loc,k)) as float64 numpy, unsorted."""
        c = lambda t: t.detach().cpu().numpy().astype(np.float64)
        return c(self.W_T), c(self.d), c(self.H)


def partition_columns_by_nnz(p, world):
    """Contiguous column blocks balanced by nnz (prefix sum over col_ptr; SURVEY.md 8e 'Partitioning').
    Returns world+1 column boundaries."""
    p = np.asarray(p, dtype=np.int64)
    n = p.shape[0] - 1
    nnz = p[-1]
    bounds = [0]
    for r in range(1, world):
        target = nnz * r // world
        c = int(np.searchsorted(p, target, side="left"))
        c = max(bounds[-1], min(c, n))
        bounds.append(c)
    bounds.append(n)
    return bounds
