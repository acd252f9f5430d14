"""Per-sweep latency of the MFMA CD kernels at fixed sweep counts (tol = 0): one lone wave, one wave per SIMD, C2 sizes."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from rcppml_amd import als, _abi
k = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sweeps = 100
ops = als.HipOps(0, "f32")
g = torch.Generator(device="cuda").manual_seed(1)
F = torch.rand((20000, k), device="cuda", dtype=torch.float32, generator=g)
G = ops.gram(F, 1e-15, 0.0)
for n in (16, 64, 16 * 1024, 20000, 32 * 1024, 100000):
    X = torch.rand((n, k), device="cuda", generator=g)
    B = X @ G + 0.1 * torch.randn((n, k), device="cuda", generator=g)
    for variant, name in ((_abi.CD_MFMA, "mfma32"), (_abi.CD_MFMA16, "mfma16")):
        Xw = torch.zeros_like(X)
        def run():
            ops.ctx.solve_cd(ops.dt, G, B, Xw, k, n, 0.0, 0, 1, 0.0, 0.0, 1, sweeps, 0.0, 0.0, 0.0, variant)
        run(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            run()
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 5
        print("cd f32 %s k=%d n=%d: %.3f ms for %d sweeps -> %.2f us per sweep (%.0f cycles per coordinate at 2.4 GHz)" % (
            name, k, n, ms, sweeps, ms * 1e3 / sweeps, ms * 1e3 / sweeps * 2400 / k))
