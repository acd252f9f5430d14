# tools/probe/rhs_shape_sweep.sh -- forced (waves, rounds) shapes of the row-tiled rhs on the C4 shard shape (needs make EXPERIMENTS=1); run on the GPU box from the repo root
export RCPPML_GPU_LIB_PATH=$PWD/rcppml_amd/lib/RcppML_gpu_exp.so
export M=30000 N=162500 K=128 DENS=0.03
echo "== planner"; python tools/rhs_tiled_bench.py 2>&1 | grep -E "side|tiled kernel"
for cfg in "12 4" "12 6" "12 8" "12 10" "16 2" "16 4" "16 6"; do set -- $cfg; echo "== NW=$1 NR=$2"; RCPPML_RT_NW=$1 RCPPML_RT_NR=$2 python tools/rhs_tiled_bench.py 2>&1 | grep -E "tiled kernel|side" | sed 's/slot_count.*tiled_columns/tc/' ; done
