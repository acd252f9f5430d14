#!/bin/bash
# profiles/pmc_cd.sh <tag> -- SQ counters for the CD solve kernels (own pass, --kernel-trace only)
TAG=${1:-cd}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d "$OUT/sq" -o sq -- python $REPO/tools/cd_sweep_histogram.py f32 > "$OUT/log.txt" 2>&1
find "$OUT" -name "*.db" -delete
python - <<PY
import csv, collections
rows=list(csv.DictReader(open("$OUT/sq/sq_counter_collection.csv")))
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if 'cd_' in r['Kernel_Name'] or 'rhs_kernel' in r['Kernel_Name']:
        acc[r['Kernel_Name'][:60]+' grid='+r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    print(k)
    for c,vals in v.items():
        print('   %-28s n=%d last=%.4g' % (c, len(vals), vals[-1]))
PY
