"""bench.py end to end on a reduced shape: the JSON line carries what the driver contract and the tier's measurement rules
ask for, and the two launch modes (replays of one captured hipGraph / eager loop) time the same arithmetic -- identical
final loss, identical work counters."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--rows", "3000", "--cols", "24000", "--density", "0.01", "--k", "64",
           "--steps", "4", "--warmup", "3", "--no-cpu-baseline", "--no-plugin-figure", "--no-cpu-ref", "--no-fp64-leg",
           "--no-noop-count", *extra]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_bench_fused_tail_equals_separate_kernels():
    """The fused iteration tail (15 launches per iteration: DESIGN 4.5) and the separate kernels (--no-fused-tail, 22 launches) are the
    same arithmetic: identical final loss and work counters, in fp32 (scaling inside the Gram's partial-tile kernel at k = 64) and fp64;
    the line says which form ran and names its phases accordingly."""
    for dt in ("f32", "f64"):
        f = _run("--dtype", dt)
        s = _run("--dtype", dt, "--no-fused-tail")
        assert f["fused_tail"] and not s["fused_tail"]
        assert f["final_loss"] == s["final_loss"], (dt, f["final_loss"], s["final_loss"])
        assert f["roofline"]["mean_sweeps_per_column"] == s["roofline"]["mean_sweeps_per_column"]
        assert set(s["phases_ms_per_step"]) >= {"scale", "gram", "loss", "solve_H", "solve_W"} and "scale_gram" in f["phases_ms_per_step"]
        for side in ("H", "W"):      # (the idle-slot figure comes from the work order the last solve RAN in, whichever form ranked it)
            assert abs(f["roofline"]["idle_slot_fraction"][side] - s["roofline"]["idle_slot_fraction"][side]) < 1e-12


def test_bench_line_graph_and_eager_modes_agree():
    g = _run()
    e = _run("--no-graph")
    for d in (g, e):
        assert d["unit"] == "cols/s" and d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 3 and d["higher_is_better"]
        assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["scaling"] == "weak" and d["vs_baseline"] is None
        assert abs(d["value"] - 4 * (3000 + 24000) / (d["ms_per_step"] * 4e-3)) / d["value"] < 1e-6
        for roof in (d["roofline"], d["roofline_rhs"]):
            assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(roof)
            assert 0 < roof["frac"] < 1 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
        # (the tail between two solves is one call per side since round 5: "scale_gram" = scaling + next work order + Gram of H,
        # "scale_gram_loss" = the same for W_T + the loss; bench.py --no-fused-tail keeps the separate "scale" / "gram" / "loss" phases)
        assert d["fused_tail"] and set(d["phases_ms_per_step"]) >= {"rhs_H", "rhs_W", "solve_H", "solve_W", "scale_gram", "scale_gram_loss"}
        assert sum(d["phases_ms_per_step"].values()) < 1.25 * max(d["ms_per_step"], d["eager_ms_per_step"] or 0)
    # the reference's protocol beside the steady state: the same number of iterations from the SplitMix64 start (no warm start and no
    # work order in the first two, more CD sweeps early in a fit): never faster than the steady-state step, and its own cols/s
    for d in (g, e):
        ffs = d["fit_from_start"]
        assert ffs["iterations"] == 4 and ffs["unit"] == "cols/s" and ffs["launch"] == "eager"
        assert ffs["ms_per_step"] >= 0.98 * d["ms_per_step"]
        assert abs(ffs["value"] - (3000 + 24000) / (ffs["ms_per_step"] * 1e-3)) / ffs["value"] < 1e-6
    assert g["fit_from_start"]["final_loss"] == e["fit_from_start"]["final_loss"]
    assert "hipGraph" in g["launch"] and g["eager_ms_per_step"] > 0
    assert e["launch"] == "eager" and e["eager_ms_per_step"] is None
    # same iterations, same kernels: the modes differ in how launches are issued, not in what is computed
    assert g["final_loss"] == e["final_loss"]
    assert g["roofline"]["mean_sweeps_per_column"] == e["roofline"]["mean_sweeps_per_column"]


def test_bench_line_carries_cpu_reference_deviation_and_fp64_leg():
    """BASELINE.json's metric is "cols solved/s + MSE vs CPU ref": the line holds the relative deviation of the fit's loss
    from the CPU reference fit (fp64 oracle, same inputs, same iteration count) for the fp32 headline AND for the fp64 leg, the
    latter inside the north star's 1e-6; plus the all-zero coordinate-step rate of the CD solve."""
    d = _run_full()
    assert d["cpu_ref"]["iterations"] == 7 and d["cpu_ref"]["dtype"] == "f64"
    assert 0 <= d["loss_rel_dev_vs_cpu_ref"] < 2e-4                     # fp32 arithmetic vs the fp64 CPU fit
    f = d["fp64"]
    assert f["ms_per_step"] > 0 and f["unit"] == "cols/s"
    assert 0 <= f["loss_rel_dev_vs_cpu_ref"] < 1e-6                     # the north star's bar
    nz = d["cd_noop_steps"]
    for key in ("H_16_columns_per_wave", "H_32_columns_per_wave", "W_16_columns_per_wave", "W_32_columns_per_wave"):
        assert 0.0 <= nz[key] < 1.0


def _run_full():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--rows", "3000", "--cols", "24000", "--density", "0.01", "--k", "64",
           "--steps", "4", "--warmup", "3", "--no-cpu-baseline", "--no-plugin-figure"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("config", ["c2", "c4"])
def test_bench_self_launcher_two_ranks_sharing_the_gpu(config):
    """`python bench.py --gpus 2` without a launcher re-executes itself under torch.distributed.run (what the driver's first
    multi-GPU run will do).  On this one-GPU box both ranks share cuda:0 and the collectives run over gloo
    (RCPPML_BENCH_BACKEND / RCPPML_BENCH_SHARE_GPU): the sharded loop must see world size 2, issue ONE all-reduce and one
    all-gather per iteration, and land on the loss of the single-rank run over the same 2 x cols columns."""
    shape = ["--rows", "3000", "--cols", "12000", "--density", "0.01"] + (["--k", "64"] if config == "c2" else ["--k", "128"])
    common = ["--config", config, "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-plugin-figure", "--no-cpu-ref",
              "--no-fp64-leg", "--no-graph", "--dtype", "f64"]
    env = dict(os.environ, RCPPML_BENCH_BACKEND="gloo", RCPPML_BENCH_SHARE_GPU="1")
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", *shape, *common], capture_output=True,
                         text=True, timeout=900, cwd=ROOT, env=env)
    assert two.returncode == 0, two.stderr[-3000:]
    d2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert d2["n_gpus"] == 2 and d2["world_size_seen"] == 2 and d2["backend"] == "gloo"
    coll = d2["collectives_ms_per_step"]
    assert set(coll) == {"all_reduce_gram_rhs_rowsums", "all_gather_W"}
    assert coll["all_reduce_gram_rhs_rowsums"]["calls_per_step"] == 1 and coll["all_gather_W"]["calls_per_step"] == 1
    k = 64 if config == "c2" else 128
    assert coll["all_reduce_gram_rhs_rowsums"]["bytes"] == 8 * (k * k + 3000 * k + k)
    one_shape = [a if a != "12000" else "24000" for a in shape]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *one_shape, "--data-shards", "2", *common], capture_output=True, text=True,
                         timeout=900, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-3000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    # fp64: the two runs differ by the summation order of the reduced Gram / right-hand side and by "scale after the sum"
    assert abs(d2["final_loss"] - d1["final_loss"]) / abs(d1["final_loss"]) < 1e-9


def test_bench_forced_one_rank_group_runs_rccl():
    """RCPPML_BENCH_FORCE_DIST=1: `--gpus 1` with a process group of ONE rank on the nccl (= RCCL) backend and the loop in its
    sharded branch -- the all-reduce of [G | B | row sums] and the all-gather of W's row blocks are really issued on the GPU,
    so RCCL has executed before the first multi-GPU run.  A one-rank sum is the identity: the loss equals the unsharded
    loop's (which scales BEFORE forming the Gram; the sharded branch scales after the sum) to fp64 rounding."""
    shape = ["--rows", "3000", "--cols", "24000", "--density", "0.01", "--k", "64"]
    common = ["--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-plugin-figure", "--no-cpu-ref", "--no-fp64-leg",
              "--no-noop-count", "--dtype", "f64"]
    env = dict(os.environ, RCPPML_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    f = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *shape, *common], capture_output=True, text=True, timeout=900,
                       cwd=ROOT, env=env)
    assert f.returncode == 0, f.stderr[-3000:]
    d = json.loads([l for l in f.stdout.splitlines() if l.startswith("{")][-1])
    assert d["backend"] == "nccl" and d["world_size_seen"] == 1 and d["forced_one_rank_group"] is True and d["launch"] == "eager"
    coll = d["collectives_ms_per_step"]
    assert set(coll) == {"all_reduce_gram_rhs_rowsums", "all_gather_W"}
    for name in coll:
        assert coll[name]["calls_per_step"] == 1 and coll[name]["ms"] > 0
    assert coll["all_reduce_gram_rhs_rowsums"]["bytes"] == 8 * (64 * 64 + 3000 * 64 + 64)
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *shape, *common, "--no-graph"], capture_output=True, text=True,
                         timeout=900, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-3000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert d1["backend"] is None and not d1["collectives_ms_per_step"]
    assert abs(d["final_loss"] - d1["final_loss"]) / abs(d1["final_loss"]) < 1e-9


def test_bench_c3_line_on_the_movielens_fixture():
    """--config c3: BASELINE configs[2] (movielens, k = 32, L1 = c(0, 0.1)) through the same loop; the fp64 leg stays inside the
    north star's 1e-6 of the CPU reference fit."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c3", "--steps", "4", "--warmup", "3", "--no-plugin-figure",
           "--cpu-seconds", "1", "--no-noop-count"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["config"]["rows"] == 3867 and d["config"]["cols_per_gpu"] == 610 and d["config"]["k"] == 32 and d["config"]["nnz_per_gpu"] == 75238
    assert "movielens" in d["config"]["workload"] and d["value"] > 0 and d["cpu_baseline"]["value"] > 0
    assert 0 <= d["loss_rel_dev_vs_cpu_ref"] < 2e-4 and 0 <= d["fp64"]["loss_rel_dev_vs_cpu_ref"] < 1e-6


def test_bench_c5_line_reduced_shape():
    """--config c5 (NB IRLS) on a reduced shape: the line carries the weighted-Gram roofline with the passes the kernel counted
    (between 1 and irls_max_iter per column) and a CPU baseline of the same half-updates."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c5", "--rows", "2000", "--cols", "12000", "--steps", "3",
           "--warmup", "2", "--cpu-seconds", "1"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["unit"] == "cols/s" and d["dtype"] == "f32" and "nb" in d["metric"] and d["value"] > 0
    assert abs(d["value"] - 3 * (2000 + 12000) / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6
    for roof in (d["roofline"], d["roofline_other_side"]):
        assert 1.0 <= roof["mean_passes_per_column"] <= 5.0 and roof["nonzero_passes"] >= d["config"]["nnz_per_gpu"]
        assert 0 < roof["frac"] < 1
        # the CD sweeps of the per-pass solves are counted by the kernels and are part of `frac` since round 6: between 1 and cd_maxit
        # per column and pass, and the weighted-Gram share alone is the smaller figure of rounds 3-5
        assert 1.0 <= roof["mean_sweeps_per_column_pass"] <= 100.0 and 0 < roof["frac_weighted_gram_only"] < roof["frac"]
    assert d["roofline"]["bound"] == "mfma" and abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-9
    assert set(d["phases_ms_per_step"]) >= {"gram", "solve_H", "solve_W", "scale", "nb_size_loss"}
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    assert d["final_loss"] == d["final_loss"]          # finite, not NaN


def test_bench_c1_line_one_kernel_fit():
    """--config c1: BASELINE configs[0] (hawaiibirds, k = 10) as one persistent kernel: the line's step is an iteration inside the
    kernel, `inside_the_kernel` accounts for workgroup 0's clock, the same fit from the start and on the multi-launch loop are beside
    it, and the loss stays within the fp32 bar of the CPU reference fit."""
    for solver in ("cd", "chol"):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c1", "--solver", solver, "--steps", "6", "--warmup", "3"]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads(out.stdout.strip().splitlines()[-1])
        assert d["config"]["rows"] == 183 and d["config"]["cols_per_gpu"] == 1183 and d["config"]["nnz_per_gpu"] == 30815 and d["config"]["k"] == 10
        assert d["steps"] == 6 and d["warmup"] == 3 and d["n_gpus"] == 1 and d["unit"] == "cols/s" and d["vs_baseline"] is None
        assert abs(d["value"] - (183 + 1183) / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
        ins = d["inside_the_kernel"]
        assert 0 < ins["half_updates_us"] < ins["kernel_us"] and 0 <= ins["barrier_wait_us"] < ins["kernel_us"]
        assert ins["kernel_us"] <= d["ms_per_step"] * 1e3 * 1.05          # the kernel's own clock cannot exceed the host's bracket
        assert d["fit_from_start"]["ms_per_step"] > 0 and d["multi_launch"]["ms_per_step"] > 0
        assert d["plugin_pcie_inclusive"]["one_kernel"]["status"] == 0 and d["plugin_pcie_inclusive"]["multi_launch"]["status"] == 0
        assert abs(d["plugin_pcie_inclusive"]["one_kernel"]["loss"] - d["plugin_pcie_inclusive"]["multi_launch"]["loss"]) <= 2e-4 * abs(d["plugin_pcie_inclusive"]["multi_launch"]["loss"])
        assert 0 <= d["loss_rel_dev_vs_cpu_ref"] < 2e-4 and d["cpu_baseline"]["value"] > 0
        assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"])
