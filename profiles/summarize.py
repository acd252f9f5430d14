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
    return "rhs_tiled_kernel" in k or "rhs_tiled_spill" in k or "rhs_tiled_reduce" in k or "rhs_stage_kernel" in k or "rhs_kernel" in k
rhs_total = sum(v["hbm_bytes_per_launch"] * v["launches"] for k, v in out.items() if is_rhs(k))
tiled_launches = sum(v["launches"] for k, v in out.items() if "rhs_tiled_kernel" in k)
gather_launches = sum(v["launches"] for k, v in out.items() if "rhs_stage_kernel" in k or "rhs_kernel<" in k)
calls = tiled_launches if tiled_launches else gather_launches
doc = dict(kernels=out)
if calls:
    doc["rhs_per_launch"] = dict(calls=calls, hbm_bytes_per_launch=rhs_total / calls,
                                 note="sum over every kernel of one rhs call (spill + tiled + reduce + tail gather), mean of both sides")
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
        if "rk::cd_mfma" in r["Kernel_Name"] and "prep" not in r["Kernel_Name"] or "rk::rhs_stage" in r["Kernel_Name"] or "rk::rhs_tiled" in r["Kernel_Name"]:
            by[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    with open(os.path.join(dst, tag + "_summary.md"), "a") as f:
        f.write("\nLaunches of the timed iterations only (the run issues 3 warm-up iterations, the 10 timed ones as hipGraph replays, and the same 10 once more eagerly for the per-phase events: the column averages the last 10 launches of each kernel, i.e. that eager re-run; from the kernel trace):\n\n| kernel | launches | avg us (all) | avg us (timed) |\n|---|---|---|---|\n")
        # kernels that run once per side share a name when both sides pick the same shape: split by grid size
        for k, v in by.items():
            v.sort()
            d = [x[1] for x in v]
            nt = 20 if "rhs_stage" in k else 10
            nt = min(nt, len(d))
            f.write("| `%s` | %d | %.1f | %.1f |\n" % (k[:70], len(d), sum(d) / len(d) / 1e3, sum(d[-nt:]) / nt / 1e3))
print(open(os.path.join(dst, tag + "_summary.md")).read())
