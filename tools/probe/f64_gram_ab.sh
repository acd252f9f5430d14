#!/bin/bash
# (needs the library build BEFORE the change under test as tools/probe/old_lib/RcppML_gpu_old.so -- a 20 MB binary, not committed; the record
#  of the one run is profiles/r05_f64_gram_ab.txt)
mkdir -p gpurun_out
RCPPML_GPU_LIB_PATH=$PWD/tools/probe/old_lib/RcppML_gpu_old.so python tools/probe/f64_gram_ab.py old 2>&1 | grep -v amdgpu
python tools/probe/f64_gram_ab.py new 2>&1 | grep -v amdgpu
python - <<'PY'
import numpy as np
a = np.load("gpurun_out/f64_gram_old.npz"); b = np.load("gpurun_out/f64_gram_new.npz")
for k in a.files:
    print(k, "bitwise equal:", bool(np.array_equal(a[k], b[k])), "symmetric:", bool(np.array_equal(b[k], b[k].T)))
PY
for rep in 1 2; do
RCPPML_GPU_LIB_PATH=$PWD/tools/probe/old_lib/RcppML_gpu_old.so python bench.py --dtype f64 --steps 20 --warmup 3 --no-cpu-baseline --no-plugin-figure --no-cpu-ref --no-noop-count 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old lib fp64 ms %.4f'%r['ms_per_step'], r['phases_ms_per_step'], 'loss %.12g'%r['final_loss'])"
python bench.py --dtype f64 --steps 20 --warmup 3 --no-cpu-baseline --no-plugin-figure --no-cpu-ref --no-noop-count 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new lib fp64 ms %.4f'%r['ms_per_step'], r['phases_ms_per_step'], 'loss %.12g'%r['final_loss'])"
done
