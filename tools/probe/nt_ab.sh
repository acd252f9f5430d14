#!/bin/bash
# A/B of the non-temporal hints on the window rhs (slot-stream LDS-DMA loads): RcppML_gpu.so (nt) against
# RcppML_gpu_np.so (make BUILD=build_np OUT=../lib/RcppML_gpu_np.so RWFLAGS=-DRW_SLOTS_DEFAULT_POLICY), alternating, on one box.
B="--no-cpu-baseline --no-cpu-ref --no-fp64-leg --no-plugin-figure --no-noop-count"
for rep in 1 2 3; do
  for lib in RcppML_gpu RcppML_gpu_np; do
    RCPPML_GPU_LIB_PATH=$PWD/rcppml_amd/lib/$lib.so python bench.py $B 2>/dev/null | python -c "
import json,sys;d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]);p=d['phases_ms_per_step'];print('$lib', 'ms_per_step %.4f' % d['ms_per_step'], 'rhs_H %.4f rhs_W %.4f' % (p['rhs_H'], p['rhs_W']))"
  done
done
for lib in RcppML_gpu RcppML_gpu_np; do
  echo "== $lib (rocprofv3 kernel trace of tools/rhs_tiled_bench.py)"
  RCPPML_GPU_LIB_PATH=$PWD/rcppml_amd/lib/$lib.so bash tools/rprof.sh ab_$lib python tools/rhs_tiled_bench.py 2>&1 | grep "rhs_win"
done
for lib in RcppML_gpu RcppML_gpu_np; do
  RCPPML_GPU_LIB_PATH=$PWD/rcppml_amd/lib/$lib.so python bench.py $B --config c4 2>/dev/null | python -c "
import json,sys;d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]);p=d['phases_ms_per_step'];print('c4 $lib', 'ms_per_step %.4f' % d['ms_per_step'], 'rhs_H %.4f rhs_W %.4f' % (p['rhs_H'], p['rhs_W']))"
  RCPPML_GPU_LIB_PATH=$PWD/rcppml_amd/lib/$lib.so python bench.py $B --dtype f64 2>/dev/null | python -c "
import json,sys;d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]);p=d['phases_ms_per_step'];print('f64 $lib', 'ms_per_step %.4f' % d['ms_per_step'], 'rhs_H %.4f rhs_W %.4f' % (p['rhs_H'], p['rhs_W']))"
done
