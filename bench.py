#!/usr/bin/env python3
"""bench.py -- ALS-NNLS NMF hot-path benchmark on MI355X (driver contract: see task statement).

A "step" is ONE full ALS iteration (H half-update, W half-update, loss) over a synthetic CSC matrix
already resident in HBM:  BASELINE.json configs[1] -- simulateNMF 20000 x 100000, 1 %-dense, k = 64,
MSE, coordinate-descent NNLS (cd_maxit = 100, cd_tol = 1e-8), fp32 arithmetic (what the reference
computes in: src/RcppFunctions_nmf.cpp:4-5; gpu/bridge_nmf.hpp:187).  Metric: ALS updates/s =
column solves per second = steps * (m + n_total) / wall.

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): columns are sharded, every rank
owns a fresh 100000-column shard (weak scaling), W_T is replicated; per iteration ONE fused
all-reduce [H H^T | H A^T | row sums of H] and -- the m columns of W being solved in row blocks,
one per rank -- one all-gather of W_T (rcppml_amd/als.py).

Launch: at N = 1 the K timed iterations are K replays of one captured hipGraph of the iteration (the plugin's loop does
the same); the per-phase HIP-event durations in the line come from an eager re-run of the same K iterations, checked
bit-identical (--no-graph times the eager loop; N > 1 always does).

Prints ONE JSON line on rank 0.
"""
import argparse
import contextlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=["c1", "c2", "c3", "c4", "c4full", "c5"], default="c2",
                    help="BASELINE.json configs[1] (20000 x 100000 per GPU, 1 %%, k = 64: the headline) or configs[3] "
                         "(pbmc3k-shaped 30000 x 162500 per GPU = 1.3 M columns over 8 GPUs, 3 %%, k = 128)")
    ap.add_argument("--rows", type=int, default=None)
    ap.add_argument("--cols", type=int, default=None, help="columns PER GPU")
    ap.add_argument("--density", type=float, default=None)
    ap.add_argument("--k", type=int, default=None)
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--solver", choices=["cd", "chol"], default="cd")
    ap.add_argument("--variant", choices=["auto", "lane", "wave"], default="auto")
    ap.add_argument("--cd-maxit", type=int, default=100)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-order", action="store_true", help="disable sweep-count column ordering")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline work (s)")
    ap.add_argument("--no-plugin-figure", action="store_true", help="skip the PCIe-inclusive 73-pointer plugin call")
    ap.add_argument("--no-fused-tail", action="store_true",
                    help="scaling / work order / loss as their separate kernels (nine and seven launches between two solves instead of four): A/B of the fused tail")
    ap.add_argument("--w-solve", choices=["block", "replicated"], default="block",
                    help="N > 1: W half-update solved in row blocks + one all-gather (default) or replicated on every rank")
    ap.add_argument("--no-graph", action="store_true", help="time an eager launch loop instead of replays of one captured hipGraph")
    ap.add_argument("--no-cpu-ref", action="store_true", help="skip the CPU reference fit (fp64 oracle, same inputs, same iteration count) "
                                                              "behind loss_rel_dev_vs_cpu_ref")
    ap.add_argument("--no-noop-count", action="store_true", help="skip the extra CD solves that count all-zero coordinate steps")
    ap.add_argument("--no-fp64-leg", action="store_true", help="skip the fp64 (parity mode) run reported under `fp64`")
    ap.add_argument("--data-shards", type=int, default=1, help="N = 1 only: build the matrix as the N = <data-shards> run does (that many "
                                                              "column shards of cols / <data-shards>, each from its own generator "
                                                              "stream) -- the single-rank twin of a multi-rank run")
    ap.add_argument("--init-f32", action="store_true", help="fp64 run starting from the fp32-rounded factors (so that one CPU "
                                                            "reference fit serves both precisions)")
    args = ap.parse_args()
    # c3: the movielens fixture (tests/golden/movielens.npz, extracted from the reference's data/movielens.rda), k = 32,
    # L1 = c(0, 0.1), mask = "zeros" (a fit-time no-op in the reference); c5: NB counts, IRLS half-updates
    # c1: the hawaiibirds fixture (tests/golden/hawaiibirds.npz, extracted from the reference's data/hawaiibirds.rda), k = 10: the whole fit
    # as ONE persistent kernel (rcppml_hip_als_small_fit)
    preset = {"c1": (183, 1183, 0.142, 10), "c2": (20000, 100000, 0.01, 64), "c4": (30000, 162500, 0.03, 128), "c4full": (30000, 1300000, 0.03, 128), "c3": (3867, 610, 0.0319, 32),
              "c5": (10000, 200000, 0.02, 32)}[args.config]
    if args.config == "c4full":
        # configs[3] at its FULL stated extent on ONE device (1.17e9 nonzeros, ~55 GB resident): generated as the eight shards of the
        # 8-GPU run and concatenated; A^T is built on the device; the CPU reference fit and the fp64 leg are skipped (hours of host time)
        if args.data_shards == 1:
            args.data_shards = 8
        args.no_cpu_ref = True
        args.no_fp64_leg = True
    args.rows = args.rows or preset[0]
    args.cols = args.cols or preset[1]
    args.density = args.density or preset[2]
    args.k = args.k or preset[3]
    return args


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU (127.0.0.1
    rendezvous on a free port), and hand its exit code back."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def calibrated_density(rows, k, target, seed):
    """simulateNMF clamps negative noisy entries to 0 and a dgCMatrix drops them; oversample so the
    stored density is `target`."""
    from rcppml_amd import data
    pilot, _, _ = data.simulate_nmf_sparse(rows, 2000, k, target, seed=seed, ncol_total=2000)
    frac = pilot.nnz / (target * rows * 2000.0)
    return target / max(frac, 1e-3)


def algorithmic_bytes_rhs(nnz, ncols, rrows, k, sv):
    """stream the CSC once, read F once, write B once."""
    return nnz * (4 + sv) + (ncols + 1) * 4 + k * rrows * sv + k * ncols * sv


def cpu_baseline(A_loc, At_loc, W_T, H, G_h, G_w, cfg_k, dtype, seconds, cd_maxit, L1_H=0.0):
    """Time the oracle's fused RHS+CD half-updates (reference fused_nnls.hpp:70-134 restated, OpenMP, fp32)
    on a column SAMPLE of the same workload, with the live factors (so CD sweep counts are representative)."""
    from oracle import oracle as O
    native = True
    try:
        O.build(native=True)
    except Exception:
        native = False
    cores = O.num_threads()
    nd = np.float32 if dtype == "f32" else np.float64
    from oracle.oracle import Csc

    def sample(A, ncols):
        ncols = min(ncols, A.cols)
        e = int(A.p[ncols])
        return Csc((A.rows, ncols), A.p[:ncols + 1], A.i[:e], A.x[:e])

    def timed(A, F, G, X, ncols):
        S = sample(A, ncols)
        t0 = time.perf_counter()
        O.fused_cd(S, F.astype(nd), G.astype(nd), X[:S.cols].astype(nd), maxit=cd_maxit, tol=1e-8, threads=0, warm=True,
                   native=native, L1=L1_H if A is A_loc else 0.0)
        return S.cols, time.perf_counter() - t0

    # pilot to size the sample, then the timed sample (H side and W side share the time budget)
    out = {}
    for side, (A, F, G, X) in dict(H=(A_loc, W_T, G_h, H), W=(At_loc, H, G_w, W_T)).items():
        c0, t0 = timed(A, F, G, X, 256 * max(1, cores // 8))
        rate = c0 / max(t0, 1e-6)
        want = int(min(A.cols, max(c0, rate * seconds / 2)))
        c1, t1 = timed(A, F, G, X, want)
        # the sample may already be the whole side (a 128-thread host needs < 1 s for it): repeat it to fill the
        # time budget and keep the median
        reps = int(max(1, min(25, seconds / 2 / max(t1, 1e-3))))
        ts = sorted([t1] + [timed(A, F, G, X, want)[1] for _ in range(reps - 1)])
        out[side] = (c1, ts[len(ts) // 2], len(ts))
    m, n = A_loc.rows, A_loc.cols
    t_iter = n * out["H"][1] / out["H"][0] + m * out["W"][1] / out["W"][0]
    return dict(value=(m + n) / t_iter, unit="cols/s", cores=cores, kind="port",
                sample="fused RHS+CD half-updates (warm start, live factors after warm-up) on the first %d of %d columns "
                       "(H side, median %.2fs of %d runs) and first %d of %d rows (W side, median %.2fs of %d runs); extrapolated "
                       "to one full iteration; excludes the loss pass; oracle built %s" % (
                           out["H"][0], n, out["H"][1], out["H"][2], out["W"][0], m, out["W"][1], out["W"][2],
                           "-O3 -march=native" if native else "-O2"),
                dtype=dtype)


def collectives_model(world, m, k, sv, w_solve):
    """Expected payloads of one ALS iteration over `world` ranks and their xGMI time from the link figures of
    MI355X_MICROARCH.md (7 links x ~153 GB/s per GPU, both directions together): a ring all-reduce moves 2 (N-1)/N of
    the buffer over every rank's links, an all-gather (N-1)/N.  `per_link_ms`: one ring on one link direction (76.5 GB/s) --
    the bound xGMI's point-to-point links set a single ring; `all_links_ms`: the buffer cut over N-1 rings, one per peer
    link.  A MODEL to compare `collectives_ms_per_step` with -- no multi-GPU run has been made (unmeasured on hardware)."""
    if world <= 1:
        return None
    ar = (k * k + k * m + k) * sv                      # [G_p | B_p | row sums of H]: the one all-reduce per iteration
    ag = (k * m * sv) if w_solve == "block" else 0     # the solved blocks of W_T
    f = (world - 1) / world
    link = 76.5e9
    t_ar, t_ag = 2 * f * ar / link, f * ag / link
    return {"all_reduce_bytes": ar, "all_gather_bytes": ag, "w_solve": w_solve,
            "per_link_ms": {"all_reduce": round(t_ar * 1e3, 4), "all_gather": round(t_ag * 1e3, 4)},
            "all_links_ms": {"all_reduce": round(t_ar * 1e3 / max(1, world - 1), 4), "all_gather": round(t_ag * 1e3 / max(1, world - 1), 4)},
            "note": "model from the xGMI link figures; unmeasured on hardware"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from rcppml_amd import als, data

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d launched with WORLD_SIZE=%d" % (args.gpus, world))
    if args.config == "c5":
        if world != 1:
            raise SystemExit("--config c5 is benchmarked on one GPU (BASELINE configs[4])")
        return bench_c5(args)
    if args.config == "c1":
        if world != 1:
            raise SystemExit("--config c1 (hawaiibirds) is a one-GPU workload")
        return bench_c1(args)
    # functional smoke of the N > 1 loop on a one-GPU box: RCPPML_BENCH_BACKEND=gloo RCPPML_BENCH_SHARE_GPU=1 maps every
    # rank onto cuda:0 (RCCL refuses two ranks per device); never used for reported numbers
    backend = os.environ.get("RCPPML_BENCH_BACKEND", "nccl")
    if os.environ.get("RCPPML_BENCH_SHARE_GPU"):
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    # RCPPML_BENCH_FORCE_DIST=1 with --gpus 1: a process group of ONE rank on the real backend (nccl = RCCL), and the loop takes
    # its sharded branch -- every collective of the N > 1 path is issued (a one-rank sum is the identity), so the RCCL wiring
    # executes on a one-GPU box and `collectives_ms_per_step` is the per-call launch floor of the library
    force_dist = world == 1 and bool(os.environ.get("RCPPML_BENCH_FORCE_DIST"))
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_dist:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    comm = als.Comm(dist if (world > 1 or force_dist) else None, time_collectives=world > 1 or force_dist, force=force_dist)

    m, n_loc, k = args.rows, args.cols, args.k
    n_total = n_loc * world
    if args.config == "c3":
        if world != 1:
            raise SystemExit("--config c3 (610 columns) is a one-GPU workload")
        fx = np.load(os.path.join(ROOT, "tests", "golden", "movielens.npz"))
        A_loc = data.CSC(tuple(int(v) for v in fx["shape"]), fx["p"], fx["i"], fx["x"])
        m, n_loc = A_loc.shape
        n_total = n_loc
    elif world == 1 and args.data_shards > 1:
        dens = calibrated_density(m, k, args.density, seed=123)
        if n_loc % args.data_shards:
            raise SystemExit("--cols must be a multiple of --data-shards")
        nsh = n_loc // args.data_shards
        parts = [data.simulate_nmf_sparse(m, nsh, k, dens, seed=123, device=torch.device("cuda", local_rank), col_offset=r * nsh,
                                          ncol_total=n_total)[0] for r in range(args.data_shards)]
        off = np.cumsum([0] + [a.nnz for a in parts])
        A_loc = data.CSC((m, n_loc), np.concatenate([parts[0].p[:1]] + [a.p[1:].astype(np.int64) + off[r] for r, a in enumerate(parts)]),
                         np.concatenate([a.i for a in parts]), np.concatenate([a.x for a in parts]))
    else:
        dens = calibrated_density(m, k, args.density, seed=123)
        A_loc, _, _ = data.simulate_nmf_sparse(m, n_loc, k, dens, seed=123, device=torch.device("cuda", local_rank),
                                               col_offset=rank * n_loc, ncol_total=n_total)
    At_loc = A_loc.transpose() if args.config != "c4full" else None      # (full extent: transposed on the device, als.ShardedALS)
    nd = np.float32 if args.dtype == "f32" else np.float64
    W0, H0 = data.init_factors(args.seed, k, m, n_loc, np.float32 if args.init_f32 else nd, col_offset=rank * n_loc, n_total=n_total)
    W0, H0 = W0.astype(nd), H0.astype(nd)
    cfg = als.AlsConfig(k=k, max_iter=args.warmup + args.steps, tol=0.0, cd_maxit=args.cd_maxit,
                        L1_H=0.1 if args.config == "c3" else 0.0,
                        solver_mode=0 if args.solver == "cd" else 1,
                        cd_variant={"auto": 0, "lane": 1, "wave": 2}[args.variant], order_columns=not args.no_order, w_solve=args.w_solve)
    # One GPU: the K timed iterations are replays of ONE captured hipGraph of the iteration -- how the plugin's loop
    # (rcppml_amd/csrc/plugin.hip) issues its steady-state iterations; an eager loop pays ~3 us of launch gap per kernel, ~35
    # kernels per iteration.  Everything then lives on a side stream (a capture cannot run on the default stream); the device
    # library is bound to it through the context's stream.  N > 1 (collectives in the loop) and --no-graph time the eager loop.
    # (a captured iteration bakes in the host-side branches of step(): warm start, Gram reuse, sweep-sorted order -- all in
    # their steady state only from the third iteration on, so shorter warm-ups time the eager loop)
    use_graph = world == 1 and not force_dist and not args.no_graph and args.warmup >= 2
    side = torch.cuda.Stream(device=local_rank) if use_graph else None
    stream_ctx = torch.cuda.stream(side) if use_graph else contextlib.nullcontext()
    with stream_ctx:
        ops = als.HipOps(local_rank, args.dtype, record_events=False)
        ops.fused_tail = not args.no_fused_tail
        st = als.ShardedALS(ops, comm, A_loc, At_loc, W0, H0, cfg)
        for _ in range(args.warmup):
            st.step()
        ops.sync()
        torch.cuda.synchronize()

        def snapshot():
            return dict(W=st.W_T.clone(), H=st.H.clone(), d=st.d.clone(), it=st.iter, Gwt=st.G_wt.clone(), gwt_ok=st._gwt_of_current_w,
                        order={sd: {key: (v.clone() if hasattr(v, "clone") else v) for key, v in o.items()} for sd, o in ops._order.items()})

        def restore(snap):
            st.set_factors(W_T=snap["W"], H=snap["H"], d=snap["d"], iteration=snap["it"])
            st.G_wt.copy_(snap["Gwt"]); st._gwt_of_current_w = snap["gwt_ok"]     # (the loss Gram doubles as the next H-side Gram)
            for sd, o in snap["order"].items():
                for key, v in o.items():
                    if hasattr(v, "clone"):
                        ops._order[sd][key].copy_(v)
                    else:
                        ops._order[sd][key] = v

        graph, launch_mode = None, "eager"
        if use_graph:
            snap = snapshot()
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    st.step()
                torch.cuda.synchronize()
                graph, launch_mode = g, "hipGraph replay (one captured ALS iteration)"
            except Exception as exc:                      # capture not available: time the eager loop, and say so
                sys.stderr.write("bench: graph capture failed (%s); timing the eager loop\n" % (exc,))
                torch.cuda.synchronize()
            restore(snap)                                  # the capture pass does not execute; state is as after the warm-up
            torch.cuda.synchronize()
        # ---- timed region: EXACTLY `steps` iterations, barrier + synchronize on both sides
        ops.record = graph is None
        ops.reset_events()
        comm.reset_events()
        ops.ctx.stats(reset=True)          # zero the library's work counters (column-sweeps of the CD kernels)
        comm.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if graph is not None:
            for _ in range(args.steps):
                graph.replay()
            loss = st.loss_out
        else:
            for i in range(args.steps):
                # (the last step's solves keep a copy of the work order they ran in -- two small device copies -- for the idle-slot figure)
                ops.keep_order_used = i == args.steps - 1
                loss = st.step()
            ops.keep_order_used = False
        torch.cuda.synchronize()
        comm.barrier()
        dt = time.perf_counter() - t0
        ops.record = False
        final_loss_t = loss.clone()
        work = ops.ctx.stats()
        eager_ms_per_step = None
        if graph is not None:
            # per-phase HIP events cannot sit inside a replayed graph: the SAME K iterations are run once more, eagerly, from the
            # state the timed region started from, with an event pair around every phase (kernel durations do not depend on how
            # the launch was issued; the eager pass's own wall time is reported as `eager_ms_per_step`)
            st.iter += args.steps
            sweeps_after = {sd: o["sweeps"].clone() for sd, o in ops._order.items()}
            restore(snap)
            ops.record = True
            ops.reset_events()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(args.steps):
                ops.keep_order_used = i == args.steps - 1
                st.step()
            ops.keep_order_used = False
            torch.cuda.synchronize()
            eager_ms_per_step = (time.perf_counter() - t1) / args.steps * 1e3
            ops.record = False
            for sd, v in sweeps_after.items():          # same iterations, same arithmetic: the two passes must agree bit for bit
                if not torch.equal(v, ops._order[sd]["sweeps"]):
                    raise RuntimeError("bench: graph replay and eager pass disagree on the %s-side sweep counts" % sd)
            if not torch.equal(final_loss_t, st.loss_out):
                raise RuntimeError("bench: graph replay and eager pass disagree on the loss")
        loss = final_loss_t
        # ---- the reference's own protocol (tools/gpu_bench_final.R:15-33: nmf(..., maxit = 20, tol = 0), timed from iteration 0): the
        # SAME number of iterations from the SplitMix64(seed) start, device-resident, loss formed every iteration.  Iterations 0 and 1
        # differ from the steady state (no warm start / no work order yet; warm-started CD needs fewer sweeps late in a fit than
        # early), so this is an eager loop -- the steady-state `ms_per_step` above is the contract's timed region, this figure is
        # what a fit from scratch pays per iteration.
        fit_from_start = None
        if world == 1 and not force_dist:
            snap_end = snapshot()                          # (everything below the timed region reads the state it left behind)
            for o in ops._order.values():
                o["valid"], o["fresh"] = False, False       # no sweep counts yet: iteration 0 and 1 run in natural column order
            st.set_factors(W_T=W0, H=H0, d=np.ones(k, nd), iteration=0)
            ops.sync()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for _ in range(args.steps):
                st.step()
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t2
            fit_from_start = {"iterations": args.steps, "ms_per_step": dt2 / args.steps * 1e3, "value": args.steps * (m + n_total) / dt2,
                              "unit": "cols/s", "launch": "eager", "final_loss": float(st.loss_out[0].item()),
                              "what": "iterations 0 .. %d from the SplitMix64(%d) start (tol = 0, loss every iteration, inputs resident in HBM, "
                                      "no host read inside the loop): the reference's bench protocol, tools/gpu_bench_final.R:15-33" % (args.steps - 1, args.seed)}
            restore(snap_end)
            torch.cuda.synchronize()
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    final_loss = float(loss[0].item())
    ev = ops.event_ms()
    coll = comm.collective_ms()

    if rank == 0:
        sv = 4 if args.dtype == "f32" else 8
        nnz = A_loc.nnz
        bytes_h = algorithmic_bytes_rhs(nnz, n_loc, m, k, sv)
        bytes_w = algorithmic_bytes_rhs(nnz, m, n_loc, k, sv)
        cnt_h, ms_h = ev.get("rhs_H", (0, 0.0))
        cnt_w, ms_w = ev.get("rhs_W", (0, 0.0))
        launches = cnt_h + cnt_w
        avg_s = (ms_h + ms_w) / max(launches, 1) * 1e-3
        avg_bytes = (bytes_h * cnt_h + bytes_w * cnt_w) / max(launches, 1)
        achieved = avg_bytes / max(avg_s, 1e-12) / 1e9
        phases = {name: round(ms / args.steps, 4) for name, (c, ms) in sorted(ev.items())}
        planned = bool(st.A.get("plans")) and bool(st.At.get("plans"))
        plan_info = {side: csc["plans"][k].info() for side, csc in (("H", st.A), ("W", st.At)) if csc.get("plans", {}).get(k)}
        # delivery rate of the gathered k-rows of F (nnz * k * s_v per launch: not compulsory HBM traffic) against
        # the two on-chip ceilings: vector L1 (64 B/clk/CU) for the gather kernel, LDS (256 B/clk/CU, ds_read_b128) for the
        # row-tiled kernel
        gather_bytes = float(nnz) * k * sv
        deliv = gather_bytes / max(avg_s, 1e-12) / 1e12
        roof_rhs = {"bound": "hbm",
                    "kernel": ("rhs_win_kernel + rhs_win_finish_kernel: SpMM-like B = F * A(:,j), F through a ring of LDS row tiles, nonzeros scheduled "
                               "over a sliding window, partition slabs + overflow summed by the finishing pass; both half-updates"
                               if planned and all(p.get("kind", "window") == "window" for p in plan_info.values()) else
                               "rhs_tiled_kernel + spill / reduce kernels (slab plan): SpMM-like B = F * A(:,j), both half-updates"
                               if planned else "rhs_stage_kernel (gather form; SpMM-like B = F * A(:,j), both half-updates)"),
                    "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                    "traffic": None, "algorithmic_bytes_per_launch": avg_bytes, "avg_launch_ms": avg_s * 1e3,
                    "rhs_H_ms": ms_h / max(cnt_h, 1), "rhs_W_ms": ms_w / max(cnt_w, 1),
                    "gathered_row_delivery": {"achieved_TBps": deliv, "bytes_per_launch": gather_bytes,
                                              "frac_of_L1_39.3TBps": deliv / 39.3, "frac_of_LDS_157TBps": deliv / 157.3},
                    "plan": plan_info}
        # NNLS solve: algorithmic flops = 2 k^2 per column and sweep (k coordinate steps, each a k-long residual
        # update; nnls_batch.hpp:96-121), sweeps counted by the kernels themselves.  fp32 k<=64 runs these updates
        # as v_mfma_f32_32x32x2_f32 (dense f32 MFMA peak 157.3 TFLOP/s = the f32 vector rate); the phase time
        # also holds the Gram padding/permutation helper kernel (a few us; the column-ordering kernels ride in the launches of
        # the scaling pass since round 5: phase "scale").
        cnt_sh, ms_sh = ev.get("solve_H", (0, 0.0))
        cnt_sw, ms_sw = ev.get("solve_W", (0, 0.0))
        cd_launches = cnt_sh + cnt_sw
        cd_s = (ms_sh + ms_sw) / max(cd_launches, 1) * 1e-3
        cd_flops = 2.0 * k * k * work["cd_column_sweeps"] / max(cd_launches, 1)
        cd_peak = 157.3 if args.dtype == "f32" else 78.6
        roof_cd = None
        if args.solver == "cd" and work["cd_columns"] > 0:
            # the two launches run different kernels at the default sizes (32-column MFMA tiles for H, 16-column tiles for
            # the W side, which has fewer tiles than SIMDs): `roofline` is the H-side kernel, the longer of the two; the
            # counted column-sweeps are split between the sides in the proportion of the last iteration's per-column sweeps
            share_h = ms_sh / max(ms_sh + ms_sw, 1e-12)
            try:
                sw_h = float(st.ops._order["H"]["sweeps"].sum().item())
                sw_w = float(st.ops._order["W"]["sweeps"].sum().item())
                if sw_h + sw_w > 0:
                    share_h = sw_h / (sw_h + sw_w)
            except Exception:
                pass
            # idle-slot fraction of the MFMA tiles: a wave sweeps until its slowest column has converged, so a tile of T
            # columns spends T * max(sweeps) column-sweeps for sum(sweeps) useful ones (last iteration's counts, in the
            # work order the kernel used: sweep-sorted for the H side, natural for the W side)
            idle = {}
            try:
                for side, tile in (("H", 32), ("W", 16 if n_loc >= 0 and m <= 32 * 1024 else 32)):
                    o = st.ops._order[side]
                    sw = o["sweeps"].cpu().numpy().astype(np.int64)
                    use_order = o["valid"] and cfg.cd_tol > 0 and sw.shape[0] >= als.ORDER_MIN_COLUMNS
                    # (with the fused tail o["order"] already ranks THESE counts for the next solve: the copy the last solve kept)
                    sq = sw[o.get("order_used", o["order"]).cpu().numpy()] if use_order else sw
                    pad = (-len(sq)) % tile
                    tl = np.concatenate([sq, np.zeros(pad, np.int64)]).reshape(-1, tile)
                    idle[side] = float(1.0 - tl.sum() / max(1, (tl.max(axis=1) * tile).sum()))
            except Exception:
                pass
            # what solve_cd_impl.hip.h launches for variant = auto: fp32 k <= 128 -> 32-column v_mfma_f32_32x32x2 tiles (16-column
            # v_mfma_f32_16x16x4 tiles on sides with fewer than 4 tiles per CU, k <= 64; lane = column 4x4 blocks for k <= 32 on large
            # sides); fp64 k <= 128 -> 16-column v_mfma_f64_16x16x4 tiles; above 128: the VALU lane-group kernel
            mfma = k <= 128 and args.variant == "auto"
            f32 = args.dtype == "f32"
            w_small16 = f32 and k <= 64 and (m + 31) // 32 < 4 * 256
            flops_h = 2.0 * k * k * work["cd_column_sweeps"] * share_h / max(cnt_sh, 1)
            flops_w = 2.0 * k * k * work["cd_column_sweeps"] * (1.0 - share_h) / max(cnt_sw, 1)
            s_h, s_w = ms_sh / max(cnt_sh, 1) * 1e-3, ms_sw / max(cnt_sw, 1) * 1e-3
            tf_h, tf_w = flops_h / max(s_h, 1e-12) / 1e12, flops_w / max(s_w, 1e-12) / 1e12
            tf = cd_flops / max(cd_s, 1e-12) / 1e12
            roof_cd = {"bound": "mfma" if mfma else "valu",
                       "kernel": (("cd_mfma_kernel (coordinate-descent NNLS on 32-column v_mfma_f32_32x32x2 tiles, H half-update)" if f32 else
                                   "cd_mfma64_kernel<double> (coordinate-descent NNLS on 16-column v_mfma_f64_16x16x4 tiles, H half-update)") if mfma
                                  else "cd_group_kernel (VALU lane groups; rank above 128 or a forced variant), H half-update"),
                       "achieved": tf_h, "peak": cd_peak, "unit": "TFLOP/s", "frac": tf_h / cd_peak, "traffic": None,
                       "algorithmic_flops_per_launch": flops_h, "avg_launch_ms": s_h * 1e3,
                       "mean_sweeps_per_column": work["cd_column_sweeps"] / work["cd_columns"],
                       "idle_slot_fraction": idle,
                       "solve_H_ms": s_h * 1e3, "solve_W_ms": s_w * 1e3,
                       "w_side": {"kernel": ("cd_mfma64_kernel<float> (16-column v_mfma_f32_16x16x4 tiles)" if (mfma and w_small16) else "same kernel"),
                                  "achieved": tf_w, "frac": tf_w / cd_peak, "avg_launch_ms": s_w * 1e3,
                                  "algorithmic_flops_per_launch": flops_w},
                       "both_sides": {"achieved": tf, "frac": tf / cd_peak, "avg_launch_ms": cd_s * 1e3}}
            if mfma and f32:
                # the kernel's own ceiling: per coordinate pair a wave issues ceil(k / 32) f32 MFMAs of 64 cycles each and a
                # dependent chain of ~100 VALU / LDS cycles that cannot overlap them on gfx950 (profiles/r03_issue_probe.txt); idle slots and
                # the residency tail come on top
                mf = 64.0 * ((k + 31) // 32)
                roof_cd["ceiling_frac"] = mf / (mf + 100.0)
                roof_cd["ceiling_what"] = ("MFMA-issue share of a coordinate pair's cycles (%d MFMA + ~100 dependent VALU / LDS, mutually exclusive on "
                                           "the f32 pipes): what `frac` would be with no idle column slot and no residency tail" % int(mf))
        cd_dominant = roof_cd is not None and (ms_sh + ms_sw) > (ms_h + ms_w)
        # HBM/fabric bytes per launch from the committed rocprofv3 PMC passes of this same command (FETCH_SIZE x2 per the
        # gfx950 correction + WRITE_SIZE, profiles/summarize.py); only meaningful for the default workload
        default_wl = (m, n_loc, k, args.dtype, world) == (20000, 100000, 64, "f32", 1)
        import glob
        pmcs = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r[0-9][0-9]_pmc.json")))
        pmc_path = pmcs[-1] if pmcs else ""          # the latest round's counter passes
        pmc_name = "profiles/" + os.path.basename(pmc_path)
        if default_wl and os.path.exists(pmc_path):
            try:
                pmc = json.load(open(pmc_path))
                if "rhs_per_launch" in pmc:
                    roof_rhs["traffic"] = pmc["rhs_per_launch"]["hbm_bytes_per_launch"]
                    roof_rhs["traffic_over_algorithmic"] = roof_rhs["traffic"] / avg_bytes
                    roof_rhs["traffic_source"] = pmc_name + " (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes; all kernels of one rhs call, mean of both sides)"
                for name, v in pmc.get("kernels", {}).items():
                    if roof_cd is not None and "cd_mfma_kernel" in name:
                        roof_cd["traffic"] = v["hbm_bytes_per_launch"]
                        roof_cd["traffic_source"] = pmc_name
            except Exception:
                pass
        out = {
            "metric": "ALS updates/sec (cols solved/s), k=%d sparse NMF" % k,
            "value": args.steps * (m + n_total) / dt,
            "unit": "cols/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%s %dx%d (x%d GPUs, column shards) %.3g%%-dense CSC, k=%d, MSE, %s NNLS "
                                   "(cd_maxit=%d, cd_tol=1e-8), L1 row normalisation, loss every iteration"
                                   % ({"c2": "configs[1]: simulateNMF", "c4": "configs[3] (one GPU's share): simulateNMF", "c4full": "configs[3] at FULL extent on one device: simulateNMF",
                                       "c3": "configs[2]: movielens (fixture of data/movielens.rda), L1 = c(0, 0.1), mask = 'zeros' (fit-time "
                                             "no-op in the reference)"}[args.config], m, n_loc, world,
                                      100.0 * nnz / (m * float(n_loc)), k,
                                      "coordinate-descent" if args.solver == "cd" else "Cholesky+clip", args.cd_maxit),
                       "rows": m, "cols_per_gpu": n_loc, "nnz_per_gpu": nnz, "k": k, "solver": args.solver,
                       "cd_variant": args.variant, "parallelism": "column-shard x%d" % world},
            # `roofline` = the kernel the iteration spends most time in; the other regime of SURVEY 8(d) beside it
            "roofline": roof_cd if cd_dominant else roof_rhs,
            ("roofline_rhs" if cd_dominant else "roofline_cd"): roof_rhs if cd_dominant else roof_cd,
            "phases_ms_per_step": phases,
            # how the timed iterations were issued; with graph replay the per-phase / per-kernel HIP-event durations above
            # come from an eager re-run of the same K iterations (checked bit-identical) right after the timed region
            "launch": launch_mode,
            "fused_tail": bool(ops.fused_tail),
            "eager_ms_per_step": eager_ms_per_step,
            "final_loss": final_loss,
            "fit_from_start": fit_from_start,
            "world_size_seen": world,
            "backend": (dist.get_backend() if (world > 1 or force_dist) else None),
            "forced_one_rank_group": force_dist,
            "collectives_ms_per_step": {name: dict(ms=round(v["total_ms"] / args.steps, 4), calls_per_step=v["count"] / args.steps,
                                                   bytes=v["bytes"]) for name, v in sorted(coll.items())},
            "collectives_model": collectives_model(world, m, k, 4 if args.dtype == "f32" else 8, args.w_solve),
        }
        if world == 1 and not args.no_plugin_figure and args.config == "c2":
            try:
                out["plugin_pcie_inclusive"] = plugin_figure(A_loc, m, n_loc, k, args.seed)
            except Exception as e:
                out["plugin_pcie_inclusive"] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1:
            try:
                W_T, dvec, H = st.factors()
                G_h = st.ops.gram(st.W_T, 1e-15, 0.0).cpu().numpy()
                G_w = st.ops.gram(st.H, 1e-15, 0.0).cpu().numpy()
                if At_loc is None:          # the W-side sample needs the first rows of A only: the leading columns of the device's A^T
                    from oracle.oracle import Csc
                    c1 = min(m, 2048)
                    tp = st.At["p"][:c1 + 1].cpu().numpy()
                    e1 = int(tp[-1])
                    At_host = Csc((n_loc, c1), tp, st.At["i"][:e1].cpu().numpy(), st.At["x"][:e1].cpu().numpy().astype(np.float64))
                else:
                    At_host = _to_oracle(At_loc)
                out["cpu_baseline"] = cpu_baseline(_to_oracle(A_loc), At_host, W_T, H, G_h, G_w, k, args.dtype,
                                                   args.cpu_seconds, args.cd_maxit, L1_H=cfg.L1_H)
                out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
            except Exception as e:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "cols/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        if world == 1 and args.solver == "cd" and args.dtype == "f32" and k <= 64 and not args.no_noop_count:
            try:
                out["cd_noop_steps"] = cd_noop_fraction(st, ops, cfg, k)
                zs = out["cd_noop_steps"]
                rf = out.get("roofline") or {}
                if rf.get("bound") == "mfma" and "H_zero_step_share" in zs:
                    # the roofline counts DENSE sweeps (every coordinate of every sweep: 2 k_pad^2 flops per column-sweep); the
                    # reference executes only the steps whose update is non-zero -- useful_frac prices the kernel against those
                    rf["zero_step_share"] = {"H": zs["H_zero_step_share"], "W": zs.get("W_zero_step_share")}
                    rf["useful_frac"] = rf["frac"] * (1.0 - zs["H_zero_step_share"])
                    rf["useful_frac_what"] = ("frac x (1 - share of (column, coordinate) steps of the H-side solve whose update is exactly 0): "
                                              "the matrix-core rate on the work the reference's CD (which skips those steps) would count")
            except Exception as e:
                out["cd_noop_steps"] = {"error": repr(e)}
        # ---- the metric's second half: deviation of the fit's loss from a CPU reference fit (fp64 oracle restatement of
        # nmf_fit<CPU>, same matrix, same starting factors, same number of iterations, tol = 0; outside the timed region)
        ref_loss = None
        if world == 1 and not args.no_cpu_ref:
            try:
                from oracle import oracle as O
                t0 = time.perf_counter()
                ref = O.nmf_fit(_to_oracle(A_loc), W0.astype(np.float64), H0.astype(np.float64), np.float64, L1=(cfg.L1_W, cfg.L1_H),
                                max_iter=args.warmup + args.steps, tol=0.0, cd_maxit=args.cd_maxit, cd_tol=1e-8,
                                solver_mode=0 if args.solver == "cd" else 1, threads=0, native=True)
                ref_loss = ref.loss
                out["cpu_ref"] = {"loss": ref.loss, "iterations": int(ref.iter), "dtype": "f64", "threads": O.num_threads(),
                                  "seconds": time.perf_counter() - t0,
                                  "what": "oracle/nmf_oracle.cpp nmf_fit (restatement of nmf/fit_cpu.hpp), fp64, same CSC and starting "
                                          "factors as the GPU run, same iteration count, tol = 0"}
                out["loss_rel_dev_vs_cpu_ref"] = abs(final_loss - ref.loss) / abs(ref.loss)
            except Exception as e:
                out["cpu_ref"] = {"error": repr(e)}
        if world == 1 and not args.no_fp64_leg and args.dtype == "f32":
            try:
                out["fp64"] = fp64_leg(args, ref_loss)
            except Exception as e:
                out["fp64"] = {"error": repr(e)}
        print(json.dumps(out))
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


def bench_c1(args):
    """BASELINE configs[0]: data(hawaiibirds) (183 x 1183, 30 815 nonzeros), k = 10, MSE, fp32.  The whole fit is ONE persistent kernel on
    one XCD (rcppml_amd/csrc/kernels_small.hip.h): a "step" is one ALS iteration inside it -- the warm-up launch runs W iterations from
    the SplitMix64 start, the timed launch the next K (iter0 = W: warm starts, no iteration-0 quirk), each bracketed by a device
    synchronise.  Beside it: the same K iterations from the start (`fit_from_start`, the reference's protocol), the multi-launch loop
    on the same input (`multi_launch`), the plugin call end to end with the one-kernel path on and off, and the CPU oracle's whole
    fit at several thread counts (`cpu_baseline` = the best of them; 256 OpenMP threads on 1 366 columns is not it)."""
    import torch
    from rcppml_amd import _abi, als, data
    from oracle import oracle as O
    fx = np.load(os.path.join(ROOT, "tests", "golden", "hawaiibirds.npz"))
    A = data.CSC(tuple(int(v) for v in fx["shape"]), fx["p"], fx["i"], fx["x"])
    At = A.transpose()
    m, n = A.shape
    k = args.k
    nd = np.float32 if args.dtype == "f32" else np.float64
    sv = 4 if args.dtype == "f32" else 8
    solver = 0 if args.solver == "cd" else 1
    W0, H0 = data.init_factors(args.seed, k, m, n, nd)
    if not _abi.small_eligible(m, n, A.nnz, k):
        raise SystemExit("--config c1: the one-kernel fit does not take this size")
    ops = als.HipOps(0, args.dtype)
    a, at = ops.upload_csc(A), ops.upload_csc(At)
    tr = ops.sumsq(a["x"])
    W, H, d = ops.to_device(W0), ops.to_device(H0), ops.zeros((k,)) + 1
    res = torch.zeros(8, dtype=torch.float64, device="cuda")
    hist = torch.zeros(max(args.warmup, args.steps, 1), dtype=torch.float64, device="cuda")

    def launch(iters, iter0):
        ops.ctx.als_small_fit(ops.dt, a, at, m, n, k, W, H, d, tr, solver_mode=solver, cd_maxit=args.cd_maxit, cd_tol=1e-8, max_iter=iters, tol=0.0,
                              iter0=iter0, loss_history=hist, result8=res)

    def reset():
        W.copy_(ops.to_device(W0)); H.copy_(ops.to_device(H0)); d.fill_(1)

    def timed(iters, iter0):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        launch(iters, iter0)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    launch(max(args.warmup, 1), 0)                      # warm-up iterations (untimed)
    torch.cuda.synchronize()
    if float(res[4].item()) != 1.0:
        raise SystemExit("--config c1: the one-kernel fit gave up at its barrier (result %s)" % res.cpu().numpy())
    dt = timed(args.steps, max(args.warmup, 1))         # the timed region: exactly K iterations
    final_loss = float(res[2].item())
    tk = res.cpu().numpy()[5:8] * 1e-2 / args.steps      # workgroup 0's 100 MHz ticks -> us per iteration
    inside = {"half_updates_us": float(tk[0]), "barrier_wait_us": float(tk[1]), "other_us": float(tk[2] - tk[0] - tk[1]), "kernel_us": float(tk[2]),
              "what": "workgroup 0's wall-clock ticks inside the timed launch, per iteration: fused rhs + solve of both sides | waiting at the four "
                      "barriers for the slowest workgroup | scaling, Gram partials, fixed-order sums, loss"}
    state = (W.clone(), H.clone(), d.clone())
    reps = []
    for _ in range(15):                                 # the same region again (fresh start + warm-up each time): spread of a 0.5 ms measurement
        reset(); launch(max(args.warmup, 1), 0)
        reps.append(timed(args.steps, max(args.warmup, 1)) / args.steps * 1e3)
    starts = []
    for _ in range(15):
        reset()
        starts.append(timed(args.steps, 0) / args.steps * 1e3)
    from_start_loss = float(res[2].item())
    # ---- the multi-launch loop on the same input (graph replays of one captured iteration: how every larger fit runs)
    multi = None
    try:
        side = torch.cuda.Stream(device=0)
        with torch.cuda.stream(side):
            ops2 = als.HipOps(0, args.dtype)
            cfg = als.AlsConfig(k=k, max_iter=args.warmup + args.steps, tol=0.0, cd_maxit=args.cd_maxit, solver_mode=solver)
            st = als.ShardedALS(ops2, als.Comm(None), A, At, W0, H0, cfg)
            for _ in range(max(args.warmup, 2)):
                st.step()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            snapW, snapH, snapd, it0 = st.W_T.clone(), st.H.clone(), st.d.clone(), st.iter
            with torch.cuda.graph(g, stream=side):
                st.step()
            st.set_factors(W_T=snapW, H=snapH, d=snapd, iteration=it0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                g.replay()
            torch.cuda.synchronize()
            multi = {"ms_per_step": (time.perf_counter() - t0) / args.steps * 1e3, "launch": "hipGraph replay of the 15-launch iteration",
                     "final_loss": float(st.loss_out[0].item())}
    except Exception as e:
        multi = {"error": repr(e)}
    # ---- the plugin call end to end (upload, transpose, fit, download), one-kernel path on / off
    plug = {}
    for name, env in (("one_kernel", None), ("multi_launch", "1")):
        old = os.environ.get("RCPPML_GPU_NO_SMALL")
        if env is None:
            os.environ.pop("RCPPML_GPU_NO_SMALL", None)
        else:
            os.environ["RCPPML_GPU_NO_SMALL"] = env
        best = None
        for _ in range(5):
            Wp, Hp = W0.astype(np.float64).copy(), H0.astype(np.float64).copy()
            t0 = time.perf_counter()
            r = _abi.nmf_unified(A.p, A.i, A.x, m, n, k, Wp, Hp, entry="float" if args.dtype == "f32" else "double", max_iter=100, tol=0.0,
                                 solver_mode=solver, cd_maxit=args.cd_maxit)
            el = time.perf_counter() - t0
            best = el if best is None else min(best, el)
        plug[name] = {"fit_100_iterations_ms": best * 1e3, "status": r["status"], "loss": r["loss"]}
        if old is None:
            os.environ.pop("RCPPML_GPU_NO_SMALL", None)
        else:
            os.environ["RCPPML_GPU_NO_SMALL"] = old
    total_iters = max(args.warmup, 1) + args.steps
    alg_bytes = 2 * (A.nnz * (4 + sv) + (m + n + 2) * 4) + 4 * k * (m + n) * sv        # both sparse passes + factors read and written
    line = {
        "metric": "ALS updates/sec (cols solved/s), k=%d sparse NMF" % k,
        "value": args.steps * (m + n) / dt, "unit": "cols/s", "n_gpus": 1, "steps": args.steps, "warmup": max(args.warmup, 1),
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "fixture (tests/golden/hawaiibirds.npz = the reference's data/hawaiibirds.rda)",
        "config": {"workload": "configs[0]: data(hawaiibirds) %dx%d, nnz %d, k=%d, MSE, %s, L1 row normalisation, loss every iteration; the whole "
                               "fit = one persistent kernel on one XCD" % (m, n, A.nnz, k, "coordinate-descent NNLS (cd_maxit=%d, cd_tol=1e-8)" % args.cd_maxit
                                                                           if solver == 0 else "Cholesky + clip"),
                   "rows": m, "cols_per_gpu": n, "nnz_per_gpu": A.nnz, "k": k, "solver": args.solver, "parallelism": "one GPU, one XCD (32 workgroups)"},
        "roofline": {"bound": "hbm", "kernel": "als_small_kernel (the whole ALS loop; 4 grid barriers per iteration among 32 workgroups of XCD 0)",
                     "achieved": alg_bytes / (dt / args.steps) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": alg_bytes / (dt / args.steps) / 1e9 / 8000.0,
                     "traffic": None, "algorithmic_bytes_per_launch": alg_bytes * args.steps, "avg_launch_ms": dt * 1e3,
                     "note": "latency-bound by construction: %.0f KB of algorithmic traffic per iteration, all of it L2-resident; the iteration is four "
                             "barriers (0.8 us each, profiles/r06_grid_barrier.txt) plus the dependent chains of one column's sparse product and "
                             "solve per phase" % (alg_bytes / 1e3)},
        "repeats_ms_per_step": {"median": float(np.median(reps)), "min": float(np.min(reps)), "max": float(np.max(reps)), "n": len(reps)},
        "fit_from_start": {"iterations": args.steps, "ms_per_step": float(np.median(starts)), "value": (m + n) / (float(np.median(starts)) * 1e-3), "unit": "cols/s",
                           "launch": "one kernel", "final_loss": from_start_loss, "what": "iterations 0 .. %d from the SplitMix64(%d) start, tol = 0, loss every "
                           "iteration, one launch + one synchronise (median of %d)" % (args.steps - 1, args.seed, len(starts))},
        "inside_the_kernel": inside, "multi_launch": multi, "plugin_pcie_inclusive": plug,
        "phases_ms_per_step": {}, "launch": "one persistent kernel for all %d iterations" % args.steps, "final_loss": final_loss, "world_size_seen": 1,
    }
    if multi and "ms_per_step" in multi:
        line["speedup_vs_multi_launch"] = multi["ms_per_step"] / line["ms_per_step"]
    W.copy_(state[0]); H.copy_(state[1]); d.copy_(state[2])
    if not args.no_cpu_ref:
        try:
            ref = O.nmf_fit(_to_oracle(A), W0.astype(np.float64), H0.astype(np.float64), np.float64, max_iter=total_iters, tol=0.0, cd_maxit=args.cd_maxit,
                            cd_tol=1e-8, solver_mode=solver, threads=1)
            line["cpu_ref"] = {"loss": ref.loss, "iterations": int(ref.iter), "dtype": "f64", "threads": 1,
                               "what": "oracle nmf_fit (restatement of nmf/fit_cpu.hpp), fp64, same CSC and starting factors, same iteration count"}
            line["loss_rel_dev_vs_cpu_ref"] = abs(final_loss - ref.loss) / abs(ref.loss)
        except Exception as e:
            line["cpu_ref"] = {"error": repr(e)}
    if not args.no_cpu_baseline:
        try:
            O.build(native=True)
            native = True
        except Exception:
            native = False
        try:
            per = {}
            for th in (1, 2, 4, 8, 16, 32, 0):
                best = None
                for _ in range(3 if th else 1):
                    t0 = time.perf_counter()
                    O.nmf_fit(_to_oracle(A), W0, H0, nd, max_iter=args.steps, tol=0.0, cd_maxit=args.cd_maxit, cd_tol=1e-8, solver_mode=solver,
                              threads=th, native=native)
                    el = time.perf_counter() - t0
                    best = el if best is None else min(best, el)
                per[str(th) if th else "all(%d)" % O.num_threads()] = best / args.steps * 1e3
            bname = min(per, key=per.get)
            line["cpu_baseline"] = {"value": (m + n) / (per[bname] * 1e-3), "unit": "cols/s", "cores": bname, "kind": "port",
                                    "sample": "the oracle's whole nmf_fit (fused RHS + solve, scaling, loss) on the same input, %d iterations from the same "
                                              "start, best of the thread counts tried; ms per iteration by thread count: %s" % (
                                                  args.steps, {kk: round(v, 4) for kk, v in per.items()}), "dtype": args.dtype}
            line["speedup_vs_cpu_baseline"] = line["fit_from_start"]["value"] / line["cpu_baseline"]["value"]
            line["speedup_vs_cpu_baseline_what"] = "fit_from_start (the same K iterations from the same start as the CPU fit) over the best CPU thread count"
        except Exception as e:
            line["cpu_baseline"] = {"value": None, "unit": "cols/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}
    print(json.dumps(line))


def bench_c5(args):
    """BASELINE configs[4]: loss = "nb" on 10 000 x 200 000 Poisson-Gamma counts (2 % dense), k = 32, fp32, one GPU.  A step is
    one outer NB iteration issued op by op in the order of the plugin's loop (rcppml_amd/csrc/plugin.hip, IRLS branch =
    nmf/fit_cpu.hpp:565-606, :811-852, :1094-1265, :1684-1753): Gram of W, IRLS half-update of H (per-entry NB weights rebuilt
    every pass: per-column weighted Gram + CD solve), scaling, the same for W, method-of-moments size update, NB likelihood."""
    import torch
    from rcppml_amd import als, data
    m, n, k = args.rows, args.cols, args.k
    A, _, _ = data.simulate_nb_counts(m, n, k, density=args.density, size=5.0, seed=123)
    At = A.transpose()
    nd = np.float32 if args.dtype == "f32" else np.float64
    W0, H0 = data.init_factors(args.seed, k, m, n, nd)
    ops = als.HipOps(0, args.dtype, record_events=False)
    W, H = ops.to_device(W0), ops.to_device(H0)
    Ad, Atd = ops.upload_csc(A), ops.upload_csc(At)
    theta = torch.full((m,), 10.0, dtype=ops.tdtype, device="cuda")              # nb_size_init (core/config.hpp)
    d = torch.ones((k,), dtype=ops.tdtype, device="cuda")
    sums, G = ops.empty((k,)), ops.empty((k, k))
    out4 = torch.zeros((4,), dtype=torch.float64, device="cuda")
    irls_max_iter, irls_tol = 5, 1e-4                                            # R defaults of nmf(): irls_max_iter, irls_tol

    def half(side):
        F, X, csc = (W, H, Ad) if side == "H" else (H, W, Atd)
        ops.gram(F, 1e-15, 0.0, out=G)
        with ops._timed("solve_" + side):
            ops.ctx.solve_irls(ops.dt, 5, csc["p"], csc["i"], csc["x"], csc["cols"], F, G, X, k, 0.0, 0.0, 1, args.cd_maxit,
                               irls_max_iter, irls_tol, theta if side == "H" else None, None if side == "H" else theta)
        with ops._timed("scale"):
            ops.row_norms(X, 0, out=sums)
            ops.apply_scaling(X, sums, 0, d)

    def step():
        half("H")
        half("W")
        with ops._timed("nb_size_loss"):         # per-row sizes, then the likelihood with them: one pass over A^T, as the plugin's loop
            ops.ctx.nb_size_update_loss(ops.dt, Atd["p"], Atd["i"], Atd["x"], m, At.nnz, W, d, H, n, k, 0.01, 1e6, theta, out4)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ops.record = True
    ops.reset_events()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops.record = False
    final_loss = float(out4[0].item())
    ev = ops.event_ms()
    phases = {name: round(ms / args.steps, 4) for name, (c, ms) in sorted(ev.items())}
    # ---- work of one more iteration, counted by the kernels themselves (outside the timed region; two atomics per column)
    ops.ctx.set_option(ops._abi.OPT_CD_COUNT_NOOP, 1)
    counts = {}
    for side in ("H", "W"):
        ops.ctx.irls_stats(reset=True)
        half(side)
        counts[side] = ops.ctx.irls_stats(reset=True)
    ops.ctx.set_option(ops._abi.OPT_CD_COUNT_NOOP, 0)
    kp = 32 if k <= 32 else 64
    sv = 4 if args.dtype == "f32" else 8
    roof = {}
    for side, ncols in (("H", n), ("W", m)):
        cnt, ms = ev["solve_" + side]
        sec = ms / cnt * 1e-3
        nzp = counts[side]["irls_nonzero_passes"]
        sweeps = counts[side].get("irls_cd_sweeps", 0)
        flops_gram = 2.0 * kp * kp * nzp                  # one rank-1 update f f^T of the kp x kp accumulator tile per nonzero and pass
        flops = flops_gram + 2.0 * kp * kp * sweeps       # + k coordinate steps of k fmas per column-sweep of the per-pass CD solves
        pk = 157.3 if args.dtype == "f32" else 78.6
        roof[side] = {"avg_launch_ms": sec * 1e3, "mean_passes_per_column": counts[side]["irls_column_passes"] / float(ncols),
                      "nonzero_passes": nzp, "cd_column_sweeps": sweeps, "mean_sweeps_per_column_pass": sweeps / max(1.0, float(counts[side]["irls_column_passes"])),
                      "algorithmic_flops_per_launch": flops, "achieved": flops / sec / 1e12,
                      "frac": flops / sec / 1e12 / pk, "frac_weighted_gram_only": flops_gram / sec / 1e12 / pk,
                      # every pass re-reads the column's (row, value) pairs and gathers one k-row of F per nonzero
                      "gathered_row_TBps": nzp * k * sv / sec / 1e12, "csc_stream_GBps": nzp * (4 + sv) / sec / 1e9}
    big = "H" if roof["H"]["avg_launch_ms"] >= roof["W"]["avg_launch_ms"] else "W"
    line = {
        "metric": "ALS updates/sec (cols solved/s), k=%d sparse NMF, loss = nb" % k,
        "value": args.steps * (m + n) / dt, "unit": "cols/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "configs[4]: NB counts %dx%d (Poisson-Gamma, size 5, %.3g%%-dense CSC), k=%d, loss = 'nb' IRLS "
                               "(irls_max_iter=%d, irls_tol=%g, cd_maxit=%d), per-row size by method of moments, NB likelihood every "
                               "iteration" % (m, n, 100.0 * A.nnz / (m * float(n)), k, irls_max_iter, irls_tol, args.cd_maxit),
                   "rows": m, "cols_per_gpu": n, "nnz_per_gpu": A.nnz, "k": k, "solver": "irls+cd", "parallelism": "one GPU"},
        "roofline": dict({"bound": "mfma", "unit": "TFLOP/s", "peak": 157.3 if args.dtype == "f32" else 78.6, "traffic": None,
                          "kernel": "irls_nb_mfma32_kernel (%s half-update: per-column weighted Gram G + F diag(w - 1) F^T on "
                                    "v_mfma_f32_32x32x2_f32, one wave per column, then the CD solve)" % big,
                          "counted": "2 k_pad^2 flops per nonzero and IRLS pass (the weighted Gram on the matrix cores) + 2 k_pad^2 per column-sweep of "
                                     "the per-pass CD solves (VALU, same f32 peak); passes and sweeps counted by the kernels in one extra, untimed "
                                     "iteration; `frac_weighted_gram_only` is the figure of rounds 3-5"}, **roof[big]),
        "roofline_other_side": roof["W" if big == "H" else "H"],
        "phases_ms_per_step": phases, "launch": "eager", "final_loss": final_loss, "world_size_seen": 1,
    }
    if not args.no_cpu_ref:
        # the metric's second half for this configuration: a WHOLE fit (3 outer NB iterations: IRLS half-updates, size updates,
        # likelihood) through the 73-pointer fp64 entry against the CPU oracle's fp64 fit from the same starting factors
        try:
            line["cpu_ref"] = c5_parity_leg(A, m, n, k, args)
            line["loss_rel_dev_vs_cpu_ref"] = line["cpu_ref"]["loss_rel_dev"]
        except Exception as e:
            line["cpu_ref"] = {"error": repr(e)}
            line["loss_rel_dev_vs_cpu_ref"] = None
    if not args.no_cpu_baseline:
        try:
            line["cpu_baseline"] = cpu_baseline_c5(A, At, W, H, theta, ops, k, nd, args)
            line["speedup_vs_cpu_baseline"] = line["value"] / line["cpu_baseline"]["value"]
        except Exception as e:
            line["cpu_baseline"] = {"value": None, "unit": "cols/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}
    print(json.dumps(line))


def c5_parity_leg(A, m, n, k, args, iters=3):
    """`iters` outer NB iterations through the plugin in fp64 (rcppml_gpu_nmf_ex: the 73-pointer argument list + loss history;
    loss_type = 5, per-row dispersion, the reference's defaults) and the same fit by the CPU oracle in fp64 (nmf/fit_cpu.hpp
    restated, OpenMP over the host's cores): relative deviation of the NB likelihood after EVERY iteration.  Outside the timed region.

    How well-posed is the comparison?  NB-IRLS starts every column at x = 0 (weights at their 1e6 cap) and the method-of-moments
    size divides by a difference of large sums, so rounding-level differences grow by 3-4 orders of magnitude per outer
    iteration -- in the reference's own arithmetic.  `cpu_self_dev_by_iteration` is the deviation of the CPU fit from ITSELF with
    the starting factors perturbed by 1e-14 relative (two draws, the larger one): the floor a second implementation (another
    summation order) can be expected to reach; tools/probe/nb_parity_probe.py shows the two curves side by side at reduced size."""
    from oracle import oracle as O
    from rcppml_amd import _abi, data
    W0, H0 = data.init_factors(args.seed, k, m, n, np.float64)
    W, H = W0.copy(), H0.copy()
    p, i, x = A.p.astype(np.int32), A.i.astype(np.int32), A.x.astype(np.float64)
    t0 = time.perf_counter()
    res = _abi.nmf_unified(p, i, x, m, n, k, W, H, entry="ex", precision=1, want_history=True, max_iter=iters, tol=0.0, solver_mode=0,
                           loss_type=5, cd_maxit=args.cd_maxit)
    t_gpu = time.perf_counter() - t0
    if res["status"] != 0:
        raise RuntimeError(res.get("error"))
    try:
        O.build(native=True)
        native = True
    except Exception:
        native = False
    C = O.Csc((m, n), A.p, A.i, A.x)
    kw = dict(max_iter=iters, tol=0.0, solver_mode=0, loss_type=5, cd_maxit=args.cd_maxit, threads=0, native=native)
    t0 = time.perf_counter()
    ref = O.nmf_fit(C, W0, H0, np.float64, **kw)
    t_cpu = time.perf_counter() - t0
    rh = np.asarray(ref.loss_history, dtype=np.float64)
    gh = np.asarray(res["loss_history"], dtype=np.float64)[:len(rh)]
    self_dev = np.zeros(len(rh))
    for draw in (1, 2):
        rs = np.random.default_rng(draw)
        r2 = O.nmf_fit(C, W0 * (1.0 + 1e-14 * rs.standard_normal(W0.shape)), H0 * (1.0 + 1e-14 * rs.standard_normal(H0.shape)),
                       np.float64, **kw)
        self_dev = np.maximum(self_dev, np.abs(np.asarray(r2.loss_history) - rh) / np.abs(rh))
    dev = np.abs(gh - rh) / np.abs(rh)
    out = {"iterations": iters, "gpu_entry": "rcppml_gpu_nmf_ex (fp64)", "gpu_loss": res["loss"], "cpu_loss": float(ref.loss),
           "loss_rel_dev": float(dev[-1]), "loss_rel_dev_by_iteration": [float(v) for v in dev],
           "cpu_self_dev_by_iteration": [float(v) for v in self_dev],
           "cpu_self_dev_what": "the CPU fit against itself, starting factors perturbed by 1e-14 relative (larger of two draws)",
           "gpu_fit_s": t_gpu, "cpu_fit_s": t_cpu, "cpu_threads": O.num_threads(),
           "d_rel_dev": float(np.abs(res["d"] - ref.d).max() / np.abs(ref.d).max()),
           "what": "same matrix, same starting factors, same iteration count, tol = 0, fp64 on both sides"}
    return out


def cpu_baseline_c5(A, At, W, H, theta, ops, k, nd, args):
    """The oracle's IRLS half-update (nnls_batch_irls.hpp restated, OpenMP over columns) on a column sample of both sides with the
    device's live factors, extrapolated to one iteration (size update and likelihood not included: O(nnz k), minor)."""
    from oracle import oracle as O
    cores = O.num_threads()
    Wh, Hh, th = W.cpu().numpy(), H.cpu().numpy(), theta.cpu().numpy()
    out = {}
    for side, (M, F) in dict(H=(A, Wh), W=(At, Hh)).items():
        G = (F.astype(np.float64).T @ F.astype(np.float64) + 1e-15 * np.eye(k)).astype(nd)
        rate, ncols = None, min(M.cols, 64 * cores)
        for _ in range(2):                                     # pilot, then a sample sized to the time budget
            e = int(M.p[ncols])
            sub = O.Csc((M.rows, ncols), M.p[:ncols + 1], M.i[:e], M.x[:e])
            t0 = time.perf_counter()
            O.irls_nb(sub, F, G, k, threads=0, dtype=nd, cd_maxit=args.cd_maxit, theta_row=th if side == "H" else None,
                      theta_col=None if side == "H" else th[:ncols])
            sec = time.perf_counter() - t0
            out[side] = (ncols, sec)
            ncols = int(min(M.cols, max(ncols, ncols / max(sec, 1e-6) * args.cpu_seconds / 2)))
    t_iter = A.cols * out["H"][1] / out["H"][0] + A.rows * out["W"][1] / out["W"][0]
    return dict(value=(A.rows + A.cols) / t_iter, unit="cols/s", cores=cores, kind="port", dtype=args.dtype,
                sample="NB-IRLS half-updates on the first %d of %d columns (H side, %.2fs) and the first %d of %d rows (W side, %.2fs), "
                       "live factors after the timed iterations; extrapolated to one iteration" % (
                           out["H"][0], A.cols, out["H"][1], out["W"][0], A.rows, out["W"][1]))


def plugin_figure(A, m, n, k, seed):
    """The 73-pointer plugin call (rcppml_gpu_nmf_unified_float: host buffers in, host buffers out -- what R reaches) on the
    same matrix: total time of 1-, 11- and 21-iteration fits -> setup (upload, device transpose, plans, download), the mean
    iteration of an 11-iteration fit (the first iterations run up to 100 CD sweeps per column) and the steady-state
    iteration (slope between 11 and 21 iterations: comparable with `ms_per_step`, which is timed after the warm-up steps).
    Never `value`: the timed region of this bench starts with the data in HBM."""
    from rcppml_amd import _abi, data
    W0, H0 = data.init_factors(seed, k, m, n, np.float64)
    p, i, x = A.p.astype(np.int32), A.i.astype(np.int32), A.x.astype(np.float64)
    t = {}
    for iters in (1, 1, 11, 21, 1, 11, 21):          # the first call warms the process up; then the best of two per length
        W, H = W0.copy(), H0.copy()                  # (a single shot now and then catches a host hiccup of 100+ ms)
        t0 = time.perf_counter()
        r = _abi.nmf_unified(p, i, x, m, n, k, W, H, entry="float", max_iter=iters, tol=0.0, solver_mode=0)
        dt = time.perf_counter() - t0
        if r["status"] != 0:
            raise RuntimeError(r.get("error"))
        if iters == 1 and 1 not in t:
            t[1] = float("inf")                      # warm-up call: not counted
            continue
        t[iters] = min(t.get(iters, float("inf")), dt)
    slope = (t[11] - t[1]) / 10
    steady = (t[21] - t[11]) / 10
    return {"entry": "rcppml_gpu_nmf_unified_float", "ms_per_iteration": slope * 1e3, "ms_per_iteration_steady": steady * 1e3,
            "setup_ms": (t[1] - slope) * 1e3, "fit_11_iterations_ms": t[11] * 1e3, "fit_21_iterations_ms": t[21] * 1e3,
            "cols_per_s_11_iterations": 11 * (m + n) / t[11], "cols_per_s_21_iterations": 21 * (m + n) / t[21]}


def cd_noop_fraction(st, ops, cfg, k):
    """how many (wave, coordinate) steps of the CD solve move NO column of the wave (b - G * 0 is exact, such a
    step could skip its matrix instructions).  One extra solve per side on copies of the live state with the counting build of
    the lane = column kernel (kernels_cd_lmf.hip.h), 16 and 32 columns per wave, in the sweep-sorted work order."""
    from rcppml_amd import _abi
    ctx = ops.ctx
    res = {}
    for side, F, X, csc in (("H", st.W_T, st.H, st.A), ("W", st.H, st.W_T, st.At)):
        G = ops.gram(F, 1e-15, 0.0)
        B = ops.rhs(csc, F)
        o = ops._order.get(side)
        order = o["order"] if (o is not None and o["valid"] and X.shape[0] >= 16384) else None
        for lg, cols in ((4, 16), (2, 32)):
            Xc = X.clone()
            ctx.set_option(_abi.OPT_CD_LMF_LANE_GROUPS, lg)
            ctx.set_option(_abi.OPT_CD_LMF_WAVES_PER_SIMD, 2)
            ctx.set_option(_abi.OPT_CD_COUNT_NOOP, 1)
            ctx.stats(reset=True)
            ctx.cd_step_stats(reset=True)
            ctx.solve_cd(ops.dt, G, B, Xc, k, Xc.shape[0], warm=1, maxit=cfg.cd_maxit, tol=cfg.cd_tol, variant=_abi.CD_LMF, col_order=order)
            stt = ctx.stats(reset=True)
            cst = ctx.cd_step_stats(reset=True)
            steps = stt["cd_slot_sweeps"] / cols * 64        # wave-sweeps x 64 coordinates (k padded to 64)
            res["%s_%d_columns_per_wave" % (side, cols)] = stt["cd_noop_steps"] / max(steps, 1)
            if cols == 32:       # per (column, coordinate): steps whose update is exactly 0 -- what the reference's `continue` skips
                res["%s_zero_step_share" % side] = cst["cd_zero_steps"] / max(cst["cd_steps"], 1)
        for opt in (_abi.OPT_CD_LMF_LANE_GROUPS, _abi.OPT_CD_LMF_WAVES_PER_SIMD, _abi.OPT_CD_COUNT_NOOP):
            ctx.set_option(opt, 0)
    res["what"] = ("*_columns_per_wave: fraction of (wave, coordinate) steps in which every column's step is exactly 0 (what a wave-uniform "
                   "skip could save); *_zero_step_share: fraction of (column, coordinate) steps whose update is exactly 0 (the reference "
                   "skips them: nnls_batch.hpp:102,106,109); steady state, sweep-sorted order")
    return res


def fp64_leg(args, ref_loss):
    """The same workload in fp64 (parity mode) as a child process: ms per step, cols/s and the loss deviation from the CPU
    reference fit.  Reported beside the fp32 headline; never `value`."""
    cmd = [sys.executable, os.path.abspath(__file__), "--dtype", "f64", "--init-f32", "--no-cpu-baseline", "--no-plugin-figure",
           "--no-cpu-ref", "--no-fp64-leg", "--steps", str(args.steps), "--warmup", str(args.warmup), "--config", args.config,
           "--rows", str(args.rows), "--cols", str(args.cols), "--density", str(args.density), "--k", str(args.k),
           "--solver", args.solver, "--cd-maxit", str(args.cd_maxit), "--seed", str(args.seed)]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode != 0 or not line:
        raise RuntimeError("fp64 leg failed: rc %d %s" % (p.returncode, p.stderr[-400:]))
    r = json.loads(line[-1])
    out = {"ms_per_step": r["ms_per_step"], "value": r["value"], "unit": r["unit"], "final_loss": r["final_loss"], "launch": r["launch"],
           "phases_ms_per_step": r["phases_ms_per_step"],
           "what": "same matrix, same (fp32-rounded) starting factors, fp64 arithmetic (rcppml_gpu_nmf_unified_double's kernels)"}
    if ref_loss is not None:
        out["loss_rel_dev_vs_cpu_ref"] = abs(r["final_loss"] - ref_loss) / abs(ref_loss)
    return out


def _to_oracle(A):
    from oracle.oracle import Csc
    return Csc((A.rows, A.cols), A.p, A.i, A.x)


if __name__ == "__main__":
    main()
