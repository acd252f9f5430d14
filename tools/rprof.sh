#!/bin/bash
# usage: tools/rprof.sh <tag> <command...>   -- rocprofv3 kernel trace + stats of a command; prints the per-kernel summary
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
( cd $GRAFT_REPO_ROOT && rocprofv3 --output-format csv --kernel-trace --stats -d $out -o t -- "$@" ) > $out/log.txt 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-90s calls %5s avg %9.1f us  min %9.1f  total %6.2f%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["Percentage"])))
PY
find $out -name "*.db" -delete; find $out -name "*trace.csv" -delete
