#!/bin/bash
# profiles/pmc_rhs.sh <tag> -- cache-path counters for the SpMM-like rhs kernel (separate --pmc passes, --kernel-trace only)
TAG=${1:-rhs}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace -d "$OUT/tcc" -o tcc -- python $REPO/tools/rhs_bench.py f32 > "$OUT/log1.txt" 2>&1
find "$OUT" -name "*.db" -delete
python - <<PY
import csv, collections, glob
for f in glob.glob("$OUT/*/*_counter_collection.csv"):
    rows=list(csv.DictReader(open(f)))
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        if 'rhs_kernel' in r['Kernel_Name']:
            acc['rhs grid='+r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items():
        print(k)
        for c,vals in v.items():
            print('   %-40s n=%d avg=%.5g' % (c, len(vals), sum(vals)/len(vals)))
PY
tail -2 "$OUT/log2.txt"
