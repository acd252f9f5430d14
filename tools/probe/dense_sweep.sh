for s in 1 2 4 8; do echo "slices=$s"; RCPPML_GPU_DENSE_SLICES=$s timeout 200 python tools/dense_bench.py 2>&1 | tail -1 | grep -o '"gemm_pair_ms": [0-9.]*'; done
timeout 300 python -m pytest tests/test_gpu_dense.py -x -q -m gpu 2>&1 | tail -2; timeout 200 python tools/probe/dense_check.py 2>&1 | tail -3
