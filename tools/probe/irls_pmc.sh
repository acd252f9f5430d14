#!/bin/bash
# SQ counters of the IRLS kernel: usage irls_pmc.sh <side> <cpw> <cd_maxit> <irls>
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_irls_$1_$2_$3_$4
rm -rf $out; mkdir -p $out
( cd $GRAFT_REPO_ROOT && rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS -d $out -o t -- python tools/probe/irls_one.py "$@" ) > $out/log.txt 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    kn = r["Kernel_Name"][:40]
    if "irls" not in kn: continue
    acc[kn][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[kn] += 1
for kn, d in acc.items():
    print(kn, "dispatches", cnt[kn])
    for c, v in sorted(d.items()): print("   %-28s %.4g per dispatch" % (c, v / max(cnt[kn], 1)))
PY
find $out -name "*.db" -delete
