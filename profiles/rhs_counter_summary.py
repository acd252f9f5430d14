#!/usr/bin/env python3
"""profiles/rhs_counter_summary.py -- one line per rhs kernel and side from the counter passes of profiles/pmc_rhs_tiled.sh
(gpurun_out/r02/pmc_tiled/) and the untraced kernel trace of profiles/prof_rhs_tiled.sh (gpurun_out/r02/prof_tiled/):
duration, delivery of the gathered k-rows of F (nnz * k * 4 bytes per call, NOT compulsory HBM traffic) against the
vector-L1 (64 B/clk/CU = 39.3 TB/s) and LDS (128 B/clk/CU = 78.6 TB/s; 256 B/clk/CU = 157 TB/s for ds_read_b128) rates,
L2 (TCC) hit rate, requests that left the L2 (TCC_EA0_RDREQ: to Infinity Cache / HBM -- the counters here do not split
those two), and FETCH_SIZE x 2 + WRITE_SIZE (the guide's gfx950 correction).  Writes profiles/r02_rhs_counter_summary.md."""
import collections
import csv
import glob
import os
import re

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "r02", "pmc_tiled")
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(src, "*", "*_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "rhs_stage" in n or "rhs_tiled_kernel" in n or "rhs_tiled_spill" in n or "rhs_tiled_reduce" in n:
            key = (n.split("(")[0].replace("void rk::", ""), r["Grid_Size"])
            cnt[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
tr = glob.glob(os.path.join(root, "gpurun_out", "r02", "prof_tiled", "**", "*kernel_trace.csv"), recursive=True)
for f in tr:
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        key = (n.split("(")[0].replace("void rk::", ""), r.get("Grid_Size") or r.get("Grid_Size_X"))
        dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
log = open(os.path.join(src, "ta.log")).read()
slots = [int(x) for x in re.findall(r"'slot_count': (\d+)", log)]
fills = [float(x) for x in re.findall(r"'fill': ([0-9.]+)", log)]
spill = [int(x) for x in re.findall(r"'spilled_nnz': (\d+)", log)]
nnz = int(round(slots[0] * fills[0] + spill[0])) if slots else 0
k, sv = 64, 4
rows = []
for key in sorted(cnt, key=lambda t: (t[0], -int(t[1]))):
    c = {n: (sum(v[len(v) // 2:]) / max(1, len(v[len(v) // 2:]))) for n, v in cnt[key].items()}
    d = sorted(dur.get(key, [0.0]))
    us = d[len(d) // 2]
    side = "-"
    if "rhs_stage" in key[0]:
        side = {"6400000": "H (gather, all columns)", "1280000": "W (gather, all columns)"}.get(key[1], "H tail columns")
    elif "rhs_tiled_kernel" in key[0]:
        side = "H" if key[1] == "196608" else "W"
    elif "spill" in key[0]:
        side = "H" if int(key[1]) > 1000000 else "W"
    full = side.startswith("H (") or side.startswith("W (") or side in ("H", "W") and "rhs_tiled_kernel" in key[0]
    deliv = nnz * k * sv / (us * 1e-6) / 1e12 if (us > 0 and full) else None
    hit = c.get("TCC_HIT_sum", 0) / max(1.0, c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0))
    rows.append((key[0], side, us, deliv, hit, c.get("TCC_EA0_RDREQ_sum", 0), c.get("TCP_TCC_READ_REQ_sum", 0), c.get("TA_BUSY_avr", 0),
                 (2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024 / 1e6))
out = os.path.join(root, "profiles", "r02_rhs_counter_summary.md")
with open(out, "w") as f:
    f.write("# rhs kernels on the C2 shape (tools/rhs_tiled_bench.py, nnz = %d, k = 64, fp32): counters per launch\n\n" % nnz)
    f.write("Gathered-row delivery = nnz * k * 4 B / duration, against 39.3 TB/s (vector L1, 64 B/clk/CU), 78.6 TB/s (LDS at 128 B/clk/CU) and 157 TB/s "
            "(`ds_read_b128`, 256 B/clk/CU).  TCC_EA0_RDREQ counts 64-byte (and 128-byte) reads that left the L2 for the fabric; these counters do not "
            "separate Infinity Cache hits from HBM reads.  Durations: median of the untraced launches (`prof_rhs_tiled.sh`); counters: mean over the "
            "timed half of the launches (`pmc_rhs_tiled.sh`, one `--pmc` group per pass).  The row-tiled kernel's delivery is quoted on ALL nonzeros of the side "
            "although the spilled ones (5-7 %) are handled by `rhs_tiled_spill_kernel` in front of it; one whole call (spill + tiled + reduce / tail) "
            "is what `bench.py` reports as `roofline_rhs.gathered_row_delivery`.\n\n")
    f.write("| kernel | side | us | delivery TB/s | of L1 39.3 | of LDS 78.6 | of LDS 157 | TCC hit | TCC->fabric reads | TCP->TCC reads | TA busy (avg cycles) | 2 x FETCH + WRITE (MB) |\n")
    f.write("|---|---|---|---|---|---|---|---|---|---|---|---|\n")
    for n, side, us, dl, hit, ea, tcp, ta, mb in rows:
        f.write("| `%s` | %s | %.1f | %s | %s | %s | %s | %.3f | %.3g | %.3g | %.3g | %.0f |\n" % (
            n[:44], side, us, "%.1f" % dl if dl else "-", "%.2f" % (dl / 39.3) if dl else "-", "%.2f" % (dl / 78.6) if dl else "-",
            "%.2f" % (dl / 157.3) if dl else "-", hit, ea, tcp, ta, mb))
print(open(out).read())
