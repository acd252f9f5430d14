#!/usr/bin/env python3
"""profiles/summarize.py <tag> -- condense gpurun_out/<tag>/ (rocprofv3 csv output of profiles/run_rocprof.sh)
into profiles/<tag>_kernel_stats.csv (the --stats table, our kernels + top torch kernels), profiles/<tag>_pmc.json and
profiles/<tag>_summary.md.  FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; on gfx950 FETCH_SIZE tallies
128-B requests at 64 B for wide (16 B/lane) coalesced reads, so the read side is DOUBLED as
/opt/skills/guides/MI355X_MICROARCH.md (section HBM) prescribes; WRITE_SIZE is taken as is (uncalibrated)."""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")

stats = list(csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_stats.csv"))))
keep = [r for r in stats if "rk::" in r["Name"]] + [r for r in stats if "rk::" not in r["Name"]][:8]
with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(stats[0].keys()))
    w.writeheader()
    for r in keep:
        w.writerow(r)


def pmc(path, name):
    acc = collections.defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name and "rk::" in r["Kernel_Name"]:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
            meta[r["Kernel_Name"]] = dict(vgpr=r["VGPR_Count"], agpr=r["Accum_VGPR_Count"], sgpr=r["SGPR_Count"],
                                          lds=r["LDS_Block_Size"], scratch=r["Scratch_Size"], wg=r["Workgroup_Size"])
    return acc, meta


fetch, meta = pmc(os.path.join(src, "pmc_fetch", "fetch_counter_collection.csv"), "FETCH_SIZE")
write, _ = pmc(os.path.join(src, "pmc_write", "write_counter_collection.csv"), "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    fv, wv = fetch.get(k, [0.0]), write.get(k, [0.0])
    f_kib, w_kib = sum(fv) / len(fv), sum(wv) / len(wv)
    out[k] = dict(launches=len(fv), fetch_size_kib_avg=f_kib, write_size_kib_avg=w_kib,
                  hbm_bytes_per_launch=(2.0 * f_kib + w_kib) * 1024.0, resources=meta.get(k, {}))
# one rhs CALL = spill kernel + tiled kernel (+ partial reduce, + gather kernel on the tail columns): traffic of everything one
# rcppml_hip_rhs_planned launches, averaged over the two sides (H and W) -- what bench.py's roofline_rhs.traffic quotes
def is_rhs(k):
    return ("rhs_win_kernel" in k or "rhs_win_finish" in k or "rhs_tiled_kernel" in k or "rhs_tiled_spill" in k or "rhs_tiled_reduce" in k
            or "rhs_stage_kernel" in k or "rhs_kernel" in k)
rhs_total = sum(v["hbm_bytes_per_launch"] * v["launches"] for k, v in out.items() if is_rhs(k))
tiled_launches = sum(v["launches"] for k, v in out.items() if "rhs_tiled_kernel" in k or "rhs_win_kernel" in k)
gather_launches = sum(v["launches"] for k, v in out.items() if "rhs_stage_kernel" in k or "rhs_kernel<" in k)
calls = tiled_launches if tiled_launches else gather_launches
doc = dict(kernels=out)
if calls:
    doc["rhs_per_launch"] = dict(calls=calls, hbm_bytes_per_launch=rhs_total / calls,
                                 note="sum over every kernel of one rhs call (r4: window kernel + finishing kernel; r2/r3: spill + tiled + reduce + tail gather), mean of both sides")
json.dump(doc, open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1)

with open(os.path.join(dst, tag + "_summary.md"), "w") as f:
    f.write("# rocprofv3 summary `%s`\n\nCommand: `python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-plugin-figure` "
            "(profiles/run_rocprof.sh; pass 1 `--kernel-trace --stats`, passes 2/3 `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE`).\n\n" % tag)
    f.write("| kernel | calls | avg us | total ms | % |\n|---|---|---|---|---|\n")
    for r in keep:
        f.write("| `%s` | %s | %.1f | %.2f | %s |\n" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                    float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
    f.write("\n| kernel | FETCH_SIZE KiB (avg) | WRITE_SIZE KiB (avg) | HBM bytes/launch (2*F+W) | VGPR | SGPR | LDS |\n|---|---|---|---|---|---|---|\n")
    for k, v in out.items():
        r = v["resources"]
        f.write("| `%s` | %.0f | %.0f | %.3e | %s | %s | %s |\n" % (k[:70], v["fetch_size_kib_avg"], v["write_size_kib_avg"],
                                                              v["hbm_bytes_per_launch"], r.get("vgpr"), r.get("sgpr"), r.get("lds")))
# the --stats averages include the warm-up launches (iteration 0 runs every column to cd_maxit); bench.py times `steps`
# iterations after them (graph replays) and re-runs the same iterations eagerly for its per-phase events, so give the
# kernel-trace average over exactly those last launches as well
trace = os.path.join(src, "trace", "trace_kernel_trace.csv")
if os.path.exists(trace):
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        if "rk::cd_mfma" in r["Kernel_Name"] and "prep" not in r["Kernel_Name"] or "rk::rhs_stage" in r["Kernel_Name"] or "rk::rhs_tiled" in r["Kernel_Name"] or "rk::rhs_win" in r["Kernel_Name"]:
            by[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    with open(os.path.join(dst, tag + "_summary.md"), "a") as f:
        f.write("\nLaunches of the timed iterations only (the run issues 3 warm-up iterations, the 10 timed ones as hipGraph replays, and the same 10 once more eagerly for the per-phase events: the column averages the last 10 launches of each kernel, i.e. that eager re-run; from the kernel trace):\n\n| kernel | launches | avg us (all) | avg us (timed) |\n|---|---|---|---|\n")
        # kernels that run once per side share a name when both sides pick the same shape: split by grid size
        for k, v in by.items():
            v.sort()
            d = [x[1] for x in v]
            nt = 20 if ("rhs_stage" in k or "rhs_win" in k) else 10          # kernels both sides share: 2 launches per iteration
            nt = min(nt, len(d))
            f.write("| `%s` | %d | %.1f | %.1f |\n" % (k[:70], len(d), sum(d) / len(d) / 1e3, sum(d[-nt:]) / nt / 1e3))
# ---- roofline fractions from the trace alone: counted work of the traced run's own bench line / trace durations of the
# same launches (the last `steps` iterations in the trace are the eager re-run of the timed iterations: same arithmetic)
line_path = os.path.join(src, "bench_under_rocprof.json")
if os.path.exists(trace) and os.path.exists(line_path) and os.path.getsize(line_path) > 2:
    line = json.load(open(line_path))
    with open(os.path.join(dst, tag + "_bench_under_rocprof.json"), "w") as f:
        json.dump(line, f)
    steps, warm = line["steps"], line["warmup"]
    iters_in_trace = warm + 2 * steps
    rows = list(csv.DictReader(open(trace)))
    dur = collections.defaultdict(list)
    for r in rows:
        if "rk::" in r["Kernel_Name"]:
            dur[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    def timed_total_ns(pred):
        tot = 0
        for k, v in dur.items():
            if not pred(k) or len(v) < iters_in_trace:          # (kernels of the one-time setup are not part of a launch)
                continue
            v.sort()
            per_iter = len(v) / float(iters_in_trace)
            nt = int(round(per_iter * steps))
            tot += sum(x[1] for x in v[len(v) - nt:]) if nt else 0
        return tot
    cd = line["roofline"] if line["roofline"].get("bound") in ("mfma", "valu") else line.get("roofline_cd")
    rhs = line["roofline_rhs"] if "roofline_rhs" in line else line["roofline"]
    with open(os.path.join(dst, tag + "_summary.md"), "a") as f:
        f.write("\n## Roofline fractions from this trace (counted work of the traced run's own bench line, `%s_bench_under_rocprof.json`, "
                "over the trace durations of the same launches)\n\n| kernel group | counted work per launch | trace avg us per launch (timed launches) | achieved | peak | frac | bench line (HIP events) |\n|---|---|---|---|---|---|---|\n" % tag)
        if cd:
            # H side: the 32-column MFMA kernel (or whatever ran the H side: the longest CD kernel by total time)
            names = [k for k in dur if "cd_" in k and "prep" not in k]
            names.sort(key=lambda k: -sum(x[1] for x in dur[k]))
            if names:
                h_ns = timed_total_ns(lambda k: k == names[0]) / float(steps)
                tf = cd["algorithmic_flops_per_launch"] / (h_ns * 1e-9) / 1e12
                f.write("| CD solve, H side (`%s`) | %.3e flop | %.1f | %.1f TFLOP/s | %.1f | **%.3f** | %.3f (%.1f us) |\n"
                        % (names[0][:60], cd["algorithmic_flops_per_launch"], h_ns / 1e3, tf, cd["peak"], tf / cd["peak"], cd["frac"], cd["avg_launch_ms"] * 1e3))
                if len(names) > 1 and "w_side" in cd:
                    w_ns = timed_total_ns(lambda k: k == names[1]) / float(steps)
                    tfw = cd["w_side"]["algorithmic_flops_per_launch"] / (w_ns * 1e-9) / 1e12
                    f.write("| CD solve, W side (`%s`) | %.3e flop | %.1f | %.1f TFLOP/s | %.1f | **%.3f** | %.3f (%.1f us) |\n"
                            % (names[1][:60], cd["w_side"]["algorithmic_flops_per_launch"], w_ns / 1e3, tfw, cd["peak"], tfw / cd["peak"],
                               cd["w_side"]["frac"], cd["w_side"]["avg_launch_ms"] * 1e3))
        if rhs:
            call_ns = timed_total_ns(lambda k: "rhs_" in k) / float(2 * steps)
            gb = rhs["algorithmic_bytes_per_launch"] / (call_ns * 1e-9) / 1e9
            f.write("| SpMM-like rhs call (all `rhs_*` kernels of a call, mean of both sides) | %.4e B | %.1f | %.0f GB/s | 8000 | **%.3f** | %.3f (%.1f us) |\n"
                    % (rhs["algorithmic_bytes_per_launch"], call_ns / 1e3, gb, gb / 8000.0, rhs["frac"], rhs["avg_launch_ms"] * 1e3))
            if doc.get("rhs_per_launch"):
                f.write("\nHBM traffic of one rhs call from the counter passes: %.4e B = %.2f x the algorithmic bytes.\n"
                        % (doc["rhs_per_launch"]["hbm_bytes_per_launch"], doc["rhs_per_launch"]["hbm_bytes_per_launch"] / rhs["algorithmic_bytes_per_launch"]))
print(open(os.path.join(dst, tag + "_summary.md")).read())
