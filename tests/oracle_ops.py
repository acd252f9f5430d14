"""CPU backend for rcppml_amd.als.ShardedALS used ONLY by the test-suite: the same `ops` interface as HipOps,
implemented with the oracle on CPU torch tensors, so the sharded loop (partitioning, fused all-reduce buffer,
scaling exchange) can be exercised under gloo with world_size > 1 on a machine without GPUs."""
import numpy as np
import torch

from oracle import oracle as O


class OracleOps:
    def __init__(self, dtype="f64"):
        self.torch = torch
        self.ndtype = np.float32 if dtype == "f32" else np.float64
        self.tdtype = torch.float32 if dtype == "f32" else torch.float64
        self.record = False

    def to_device(self, a, dtype=None):
        t = torch.from_numpy(np.ascontiguousarray(a))
        return t.to(dtype) if dtype is not None else t

    def empty(self, shape, dtype=None):
        return torch.zeros(shape, dtype=dtype or self.tdtype)

    zeros = empty

    def upload_csc(self, A):
        return dict(rows=A.rows, cols=A.cols, nnz=A.nnz, csc=O.Csc((A.rows, A.cols), A.p, A.i, A.x),
                    x=torch.from_numpy(A.x.astype(self.ndtype)))

    def gram(self, F, eps, l2, out=None, tag=None):
        Fn = F.numpy()
        G = O.gram(Fn)                       # includes + 1e-15
        if eps == 0.0:
            G[np.diag_indices(G.shape[0])] -= self.ndtype(1e-15)
        G[np.diag_indices(G.shape[0])] += self.ndtype(l2)
        out.copy_(torch.from_numpy(G))
        return out

    def rhs(self, csc, F, out=None, tag=None):
        out.copy_(torch.from_numpy(O.rhs(csc["csc"], F.numpy(), self.ndtype)))
        return out

    def plan_rhs(self, csc, k, **kw):        # the LDS row-tiled plan is a device-side optimisation: nothing to do here
        return None

    def solve(self, G, B, X, cfg, side, warm, tag=None):
        l1 = cfg.L1_H if side == "H" else cfg.L1_W
        ub = cfg.ub_H if side == "H" else cfg.ub_W
        nonneg = cfg.nonneg_H if side == "H" else cfg.nonneg_W
        Gn, Bn, Xn = G.numpy(), B.numpy().copy(), X.numpy()
        if l1 > 0:
            Bn -= self.ndtype(l1)
        if cfg.solver_mode == 0:
            if warm:
                res = O.nnls_batch(Gn, Bn, X=Xn, maxit=cfg.cd_maxit, tol=cfg.cd_tol, nonneg=nonneg, warm=True)
            else:   # iteration-0 quirk: start from X, no residual correction
                res = np.stack([O.cd_col(Gn, Bn[j], Xn[j], nonneg=nonneg, maxit=cfg.cd_maxit, tol=cfg.cd_tol)[0]
                                for j in range(Xn.shape[0])])
        else:
            res = O.chol_clip_batch(Gn, Bn, nonneg=nonneg)
        if ub > 0:
            res = np.minimum(res, self.ndtype(ub))
        X.copy_(torch.from_numpy(res))

    def row_norms(self, X, norm_type, out=None):
        Xn = X.numpy()
        s = np.abs(Xn).sum(axis=0) if norm_type == 0 else (Xn * Xn).sum(axis=0)
        out.copy_(torch.from_numpy(s.astype(self.ndtype)))
        return out

    def apply_scaling(self, X, sums, norm_type, d):
        if norm_type == 2:
            d.fill_(1.0)
            return
        s = sums.numpy()
        dd = (np.sqrt(s) if norm_type == 1 else s) + self.ndtype(1e-15)
        d.copy_(torch.from_numpy(dd.astype(self.ndtype)))
        X.div_(d)

    def scale_order(self, X, sums, norm_type, d, cfg, side):      # HipOps: the scaling pass + the next solve's work order in shared launches
        self.row_norms(X, norm_type, out=sums)
        self.apply_scaling(X, sums, norm_type, d)

    def tail_scale_gram(self, X, sums, norm_type, d, cfg, side, eps, G):
        self.scale_order(X, sums, norm_type, d, cfg, side)
        self.gram(X, eps, 0.0, out=G)

    def tail_scale_gram_loss(self, W_T, sums, norm_type, d, cfg, side, eps, trAtA, B_w, G_saved, G_wt, out):
        self.scale_order(W_T, sums, norm_type, d, cfg, side)
        self.gram_loss_mse(W_T, eps, trAtA, d, B_w, G_saved, G_wt, out)

    def gram_loss_mse(self, W_T, eps, trAtA, d, B_w, G_saved, G_wt, out):
        self.gram(W_T, eps, 0.0, out=G_wt)
        self.loss_mse(trAtA, d, W_T, B_w, G_wt, G_saved, out)

    def sumsq(self, x):
        return torch.tensor([float((x.double() ** 2).sum())], dtype=torch.float64)

    def loss_mse(self, trAtA, d, W_T, B_w, G_wt, G_saved, out):
        dn = d.double()
        cross = float(((W_T.double() * dn) * B_w.double()).sum())
        recon = float((torch.outer(dn, dn) * G_wt.double() * G_saved.double()).sum())
        out[0] = float(trAtA[0]) - 2.0 * cross + recon
        out[1] = cross
        out[2] = recon

    def add_diag(self, G, v):
        if v != 0:
            G.diagonal().add_(v)

    def sync(self):
        pass
