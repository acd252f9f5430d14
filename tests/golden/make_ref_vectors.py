#!/usr/bin/env python3
"""tests/golden/make_ref_vectors.py -- golden vectors produced BY THE REFERENCE ITSELF.

Runs the two self-contained reference headers (rng/rng.hpp, math/loss.hpp) through oracle/_ref/libref_rng.so and
libref_loss.so (built by `make -C oracle ref` from /root/reference, sources never copied) and stores inputs + outputs
in tests/golden/ref_vectors.npz, so that the parity of the restatement can be checked where the reference tree does not
exist (the GPU box).  Data only: seeds, shapes, argument grids and the reference's numeric outputs."""
import ctypes as C
import os
import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
rng = C.CDLL(os.path.join(root, "oracle", "_ref", "libref_rng.so"))
loss = C.CDLL(os.path.join(root, "oracle", "_ref", "libref_loss.so"))
rng.ref_next.restype = C.c_uint64
rng.ref_hash.restype = C.c_uint64
for f, t in ((loss.ref_irls_weight_nb_f64, C.c_double), (loss.ref_loss_nb_f64, C.c_double),
             (loss.ref_irls_weight_nb_f32, C.c_float), (loss.ref_loss_nb_f32, C.c_float),
             (loss.ref_irls_weight_kl_f64, C.c_double), (loss.ref_irls_weight_kl_f32, C.c_float),
             (loss.ref_loss_gp_f64, C.c_double), (loss.ref_loss_gp_f32, C.c_float),
             (loss.ref_irls_weight_power_f64, C.c_double), (loss.ref_irls_weight_power_f32, C.c_float),
             (loss.ref_loss_dev_f64, C.c_double), (loss.ref_loss_dev_f32, C.c_float),
             (loss.ref_robust_modifier_f64, C.c_double), (loss.ref_robust_modifier_f32, C.c_float),
             (loss.ref_robust_loss_f64, C.c_double), (loss.ref_robust_loss_f32, C.c_float)):
    f.restype = t

out = {}
seeds = np.array([0, 1, 42, 123, 12345, 2**32 + 7, 2**63 + 5], dtype=np.uint64)
out["seeds"] = seeds
k, m, n = 7, 13, 11
out["init_shape"] = np.array([k, m, n])
for name, dt, fn in (("f64", np.float64, rng.ref_init_factors_f64), ("f32", np.float32, rng.ref_init_factors_f32)):
    Ws, Hs = [], []
    for s in seeds:
        W = np.zeros((m, k), dt)
        H = np.zeros((n, k), dt)
        fn(C.c_uint64(int(s)), k, m, n, W.ctypes.data_as(C.c_void_p), H.ctypes.data_as(C.c_void_p))
        Ws.append(W)
        Hs.append(H)
    out["init_W_" + name] = np.stack(Ws)
    out["init_H_" + name] = np.stack(Hs)
# raw 64-bit stream
nxt = []
for s in seeds:
    st = C.c_uint64(12345 if int(s) == 0 else int(s))
    nxt.append([rng.ref_next(C.byref(st)) for _ in range(8)])
out["next8"] = np.array(nxt, dtype=np.uint64)
# CV hash / holdout
ii, jj = np.meshgrid(np.array([0, 1, 2, 17, 999, 65535, 2**31 - 1], dtype=np.uint32),
                     np.array([0, 1, 5, 100, 12345, 2**31 - 1], dtype=np.uint32), indexing="ij")
out["hash_i"], out["hash_j"] = ii.ravel(), jj.ravel()
for s in (42, 7):
    out["hash_seed%d" % s] = np.array([rng.ref_hash(C.c_uint64(s), C.c_uint32(int(a)), C.c_uint32(int(b)))
                                       for a, b in zip(ii.ravel(), jj.ravel())], dtype=np.uint64)
    out["holdout_seed%d_inv10" % s] = np.array([rng.ref_is_holdout(C.c_uint64(s), C.c_uint32(int(a)), C.c_uint32(int(b)), C.c_uint64(10))
                                                 for a, b in zip(ii.ravel(), jj.ravel())], dtype=np.int32)
# NB weight and NLL grids
pred = np.array([0.0, 1e-20, 1e-12, 1e-6, 0.01, 0.5, 1.0, 3.7, 25.0, 1e3, 1e6])
size = np.array([0.0, 1e-12, 0.01, 0.5, 1.0, 5.0, 10.0, 1e3, 1e6])
obs = np.array([0.0, 1.0, 2.0, 7.0, 50.0, 1000.0])
P, R = np.meshgrid(pred, size, indexing="ij")
out["nb_pred"], out["nb_size"] = P.ravel(), R.ravel()
out["nb_weight_f64"] = np.array([loss.ref_irls_weight_nb_f64(C.c_double(a), C.c_double(b)) for a, b in zip(P.ravel(), R.ravel())])
out["nb_weight_f32"] = np.array([loss.ref_irls_weight_nb_f32(C.c_float(a), C.c_float(b)) for a, b in zip(P.ravel(), R.ravel())], dtype=np.float32)
Y, P3, R3 = np.meshgrid(obs, pred, size, indexing="ij")
out["nll_obs"], out["nll_pred"], out["nll_size"] = Y.ravel(), P3.ravel(), R3.ravel()
out["nll_f64"] = np.array([loss.ref_loss_nb_f64(C.c_double(y), C.c_double(a), C.c_double(b)) for y, a, b in zip(Y.ravel(), P3.ravel(), R3.ravel())])
out["nll_f32"] = np.array([loss.ref_loss_nb_f32(C.c_float(y), C.c_float(a), C.c_float(b)) for y, a, b in zip(Y.ravel(), P3.ravel(), R3.ravel())], dtype=np.float32)
# GP: KL weight (what the GP half-updates use) and GP likelihood (theta = 0 and a few positive values)
out["kl_pred"] = pred
out["kl_weight_f64"] = np.array([loss.ref_irls_weight_kl_f64(C.c_double(a)) for a in pred])
out["kl_weight_f32"] = np.array([loss.ref_irls_weight_kl_f32(C.c_float(a)) for a in pred], dtype=np.float32)
theta = np.array([0.0, 0.1, 1.5])
Yg, Pg, Tg = np.meshgrid(obs, pred, theta, indexing="ij")
out["gp_obs"], out["gp_pred"], out["gp_theta"] = Yg.ravel(), Pg.ravel(), Tg.ravel()
out["gp_loss_f64"] = np.array([loss.ref_loss_gp_f64(C.c_double(y), C.c_double(a), C.c_double(b)) for y, a, b in zip(Yg.ravel(), Pg.ravel(), Tg.ravel())])
out["gp_loss_f32"] = np.array([loss.ref_loss_gp_f32(C.c_float(y), C.c_float(a), C.c_float(b)) for y, a, b in zip(Yg.ravel(), Pg.ravel(), Tg.ravel())], dtype=np.float32)
# GP weight as the CV half-updates call it (nmf/cv_detail.hpp:101-131 -> math/loss.hpp irls_weight_gp): grid over
# (observed, predicted, theta, blend), including the CV call's observed = 0, theta = 0, blend = 1
for nm, T in (("f64", C.c_double), ("f32", C.c_float)):
    getattr(loss, "ref_irls_weight_gp_" + nm).restype = T
    getattr(loss, "ref_irls_weight_gp_" + nm).argtypes = [T] * 4
Yw, Pw_, Tw, Bw = np.meshgrid(obs, pred, theta, np.array([0.0, 0.5, 1.0]), indexing="ij")
out["gpw_obs"], out["gpw_pred"], out["gpw_theta"], out["gpw_blend"] = Yw.ravel(), Pw_.ravel(), Tw.ravel(), Bw.ravel()
out["gpw_f64"] = np.array([loss.ref_irls_weight_gp_f64(y, a, t, b) for y, a, t, b in zip(Yw.ravel(), Pw_.ravel(), Tw.ravel(), Bw.ravel())])
out["gpw_f32"] = np.array([loss.ref_irls_weight_gp_f32(y, a, t, b) for y, a, t, b in zip(Yw.ravel(), Pw_.ravel(), Tw.ravel(), Bw.ravel())], dtype=np.float32)
# power-variance family: weight 1/mu^p and the Gamma / inverse-Gaussian / Tweedie deviance terms
powers = np.array([1.0, 1.5, 2.0, 2.5, 3.0])
Pp, Pw = np.meshgrid(pred, powers, indexing="ij")
out["pow_pred"], out["pow_power"] = Pp.ravel(), Pw.ravel()
out["pow_weight_f64"] = np.array([loss.ref_irls_weight_power_f64(C.c_double(a), C.c_double(b)) for a, b in zip(Pp.ravel(), Pw.ravel())])
out["pow_weight_f32"] = np.array([loss.ref_irls_weight_power_f32(C.c_float(a), C.c_float(b)) for a, b in zip(Pp.ravel(), Pw.ravel())], dtype=np.float32)
yv = np.array([0.0, 0.3, 1.0, 2.5, 40.0])
LT, Yd, Pd, Wd = np.meshgrid(np.array([6, 7, 8]), yv, pred, np.array([1.0, 1.5, 2.0, 2.7]), indexing="ij")
out["dev_type"], out["dev_obs"], out["dev_pred"], out["dev_power"] = LT.ravel(), Yd.ravel(), Pd.ravel(), Wd.ravel()
out["dev_f64"] = np.array([loss.ref_loss_dev_f64(C.c_int(int(t)), C.c_double(y), C.c_double(a), C.c_double(b))
                           for t, y, a, b in zip(LT.ravel(), Yd.ravel(), Pd.ravel(), Wd.ravel())])
out["dev_f32"] = np.array([loss.ref_loss_dev_f32(C.c_int(int(t)), C.c_float(y), C.c_float(a), C.c_float(b))
                           for t, y, a, b in zip(LT.ravel(), Yd.ravel(), Pd.ravel(), Wd.ravel())], dtype=np.float32)
# robust: Huber modifier of the Pearson residual and the robust loss for every implemented distribution (0 = MSE)
rr = np.array([-50.0, -1.345, -0.3, 0.0, 1e-9, 0.7, 1.345, 2.0, 1e4])
dl = np.array([1e-4, 1.345, 3.0])
Rr, Dl = np.meshgrid(rr, dl, indexing="ij")
out["rob_r"], out["rob_delta"] = Rr.ravel(), Dl.ravel()
out["rob_mod_f64"] = np.array([loss.ref_robust_modifier_f64(C.c_double(a), C.c_double(b)) for a, b in zip(Rr.ravel(), Dl.ravel())])
out["rob_mod_f32"] = np.array([loss.ref_robust_modifier_f32(C.c_float(a), C.c_float(b)) for a, b in zip(Rr.ravel(), Dl.ravel())], dtype=np.float32)
T_, Y_, P_, D_ = np.meshgrid(np.array([0, 4, 5, 6, 7, 8]), np.array([0.0, 1.0, 3.0, 40.0]), np.array([1e-12, 0.01, 1.0, 3.7, 1e3]),
                             np.array([0.5, 1.345]), indexing="ij")
out["rl_type"], out["rl_obs"], out["rl_pred"], out["rl_delta"] = T_.ravel(), Y_.ravel(), P_.ravel(), D_.ravel()
out["rl_f64"] = np.array([loss.ref_robust_loss_f64(C.c_int(int(t)), C.c_double(y), C.c_double(p_), C.c_double(5.0), C.c_double(1.5), C.c_double(d_))
                          for t, y, p_, d_ in zip(T_.ravel(), Y_.ravel(), P_.ravel(), D_.ravel())])
out["rl_f32"] = np.array([loss.ref_robust_loss_f32(C.c_int(int(t)), C.c_float(y), C.c_float(p_), C.c_float(5.0), C.c_float(1.5), C.c_float(d_))
                          for t, y, p_, d_ in zip(T_.ravel(), Y_.ravel(), P_.ravel(), D_.ravel())], dtype=np.float32)
np.savez_compressed(os.path.join(here, "ref_vectors.npz"), **out)
print("wrote ref_vectors.npz:", {k2: v.shape for k2, v in out.items()})
