#!/bin/bash
# rocprofv3 kernel trace of tools/rhs_tiled_bench.py (gather kernel vs LDS row-tiled kernel on the C2 shape).
# Run on the GPU box from the repo root: bash profiles/prof_rhs_tiled.sh [args of the tool]
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r02/prof_tiled
mkdir -p $OUT
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT -o tb -- python tools/rhs_tiled_bench.py "$@" > $OUT/bench.log 2>&1
grep -v simple_timer $OUT/bench.log | tail -8
python profiles/sumtrace.py $OUT/tb_kernel_trace.csv; python - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"][:100], r["Calls"], r["AverageNs"], r["Percentage"])
PY
