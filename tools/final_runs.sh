#!/bin/bash
# tools/final_runs.sh <tag> -- the round's bench lines and rocprofv3 passes in one gpurun call (run from the repo root on the GPU box);
TAG=${1:-r06}
O=gpurun_out/$TAG; mkdir -p $O
timeout 400 python bench.py > $O/bench_line.json 2> $O/bench_line.err
bash profiles/run_rocprof.sh $TAG > $O/rocprof.log 2>&1
for c in c3 c4 c5 c1; do timeout 400 python bench.py --config $c > $O/bench_line_$c.json 2> $O/bench_line_$c.err; done
timeout 300 python bench.py --config c1 --solver chol > $O/bench_line_c1_chol.json 2> $O/bench_line_c1_chol.err
timeout 400 python bench.py --dtype f64 --no-plugin-figure > $O/bench_line_f64.json 2> $O/bench_line_f64.err
timeout 600 python bench.py --config c4full > $O/bench_line_c4full.json 2> $O/bench_line_c4full.err
ls -la $O | head -40
