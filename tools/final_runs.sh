#!/bin/bash
# final bench lines of the round (untraced runs)
python bench.py > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench_line.err
python bench.py --dtype f64 --no-plugin-figure > gpurun_out/r05_bench_line_f64.json 2>/dev/null
python bench.py --config c4 --no-plugin-figure > gpurun_out/r05_bench_line_c4.json 2>/dev/null
python bench.py --config c4 --dtype f64 --no-plugin-figure --no-cpu-baseline > gpurun_out/r05_bench_line_c4_f64.json 2>/dev/null
python bench.py --config c3 > gpurun_out/r05_bench_line_c3.json 2>/dev/null
python bench.py --config c5 > gpurun_out/r05_bench_line_c5.json 2>/dev/null
python tools/probe/nb_parity_probe.py 2>&1 | grep -v amdgpu > gpurun_out/r05_nb_parity_probe.txt
./tools/probe/lds_valu_probe > gpurun_out/r05_lds_valu_probe.txt 2>&1
