#!/bin/bash
# profiles/pmc_rhs_tiled.sh [tool args] -- cache-path / issue counters of the rhs kernels (gather and LDS row-tiled) on the C2
# shape.  Separate --pmc passes with --kernel-trace only.  Output: gpurun_out/r02/pmc_tiled/summary.txt
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r02/pmc_tiled; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/rhs_tiled_bench.py $*"
run() { name=$1; shift; timeout 300 rocprofv3 --output-format csv --pmc "$@" --kernel-trace -d "$OUT/$name" -o $name -- $CMD > "$OUT/$name.log" 2>&1; }
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS
run ta TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
find "$OUT" -name "*.db" -delete
python - <<PY > "$OUT/summary.txt"
import csv, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'rhs_stage' in n or 'rhs_tiled_kernel' in n or 'rhs_tiled_reduce' in n:
            key = n.split('(')[0][-44:] + ' grid=' + r['Grid_Size'] + ' wg=' + r['Workgroup_Size']
            acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(k)
    for c, vals in v.items():
        vals = vals[len(vals) // 2:]          # the timed launches
        print('   %-32s n=%d avg=%.6g' % (c, len(vals), sum(vals) / len(vals)))
PY
cat "$OUT/summary.txt"
