#!/bin/bash
# profiles/run_rocprof.sh <tag> -- collect the rocprofv3 evidence bench.py's roofline numbers are checked against.
# Run on the GPU box from the repo root (gpurun): writes raw output under gpurun_out/<tag>/, copy the
# summaries you want judged into profiles/.
#   pass 1: --kernel-trace --stats          (per-kernel time; average duration of rhs_kernel must agree with bench.py)
#   pass 2: --pmc FETCH_SIZE                (HBM read bytes; gfx950: double it for wide coalesced reads, MI355X_MICROARCH.md)
#   pass 3: --pmc WRITE_SIZE
set -u
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-plugin-figure --no-cpu-ref --no-fp64-leg --no-noop-count"
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/trace" -o trace -- $CMD > "$OUT/bench_trace.log" 2>&1
timeout 300 rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o fetch -- $CMD > "$OUT/bench_fetch.log" 2>&1
timeout 300 rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o write -- $CMD > "$OUT/bench_write.log" 2>&1
# the traced run's OWN bench line (work counters and HIP-event durations of exactly the launches in the trace): summarize.py
# computes the roofline fractions from it and the trace durations
grep "^{" "$OUT/bench_trace.log" | tail -1 > "$OUT/bench_under_rocprof.json"
find "$OUT" -name "*.db" -delete; find "$OUT" -name "*.csv" | head -20
ls -la "$OUT"/*
