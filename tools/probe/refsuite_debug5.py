import sys, os
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib.util
spec = importlib.util.spec_from_file_location("rp", os.path.join(ROOT, "tests", "test_gpu_reference_properties.py"))
rp = importlib.util.module_from_spec(spec); spec.loader.exec_module(rp)
O = rp.O
gpu = "--gpu" in sys.argv
if gpu:
    from rcppml_amd import nmf as N
A, _, _ = rp.simulate(50, 40, 3, noise=0.1, dropout=0.2, seed=8)
k, m, n = 3, 50, 40
T = np.abs(np.random.default_rng(3).standard_normal((k, n)))
W0, H0 = rp.inits(42, m, n, k)
for lam in (0.5, -0.5):
    for solver, sm in (("cd", 0), ("cholesky", 1)):
        for it in (1, 2, 3, 5, 10):
            ref = O.nmf_fit(rp.csc_o(A), W0, H0, np.float64, max_iter=it, tol=0.0, solver_mode=sm, target_H=(T.T.copy(), lam))
            line = "target lam %+.1f %-8s it %2d oracle loss %.9g" % (lam, solver, it, ref.loss)
            if gpu:
                mod = N.nmf(A, k, maxit=it, tol=0.0, seed=42, precision="fp64", solver=solver, target_H=T, target_lambda=lam)
                line += "  gpu %.9g rel %.2e dH %.2e dW %.2e" % (mod.misc["loss"], abs(mod.misc["loss"] - ref.loss) / abs(ref.loss), np.abs(mod.h.T - ref.H).max(), np.abs(mod.w - ref.W_T).max())
            print(line, flush=True)
S = rp.simulate_gp(60, 40, 3, 1.0)
W0, H0 = rp.inits(42, 60, 40, 3)
ref = O.nmf_fit(rp.csc_o(S), W0, H0, np.float64, max_iter=50, tol=0.0, loss_type=4, dispersion_mode=2)
print("oracle GP per_row loss every 10:", ref.loss_history[9::10])
if gpu:
    mod = N.nmf(S, 3, loss="gp", dispersion="per_row", maxit=50, tol=0.0, seed=42, precision="fp64")
    print("gpu    GP per_row loss every 10:", np.asarray(mod.misc["loss_history"])[9::10])
