#!/bin/bash
# A/B of the fused iteration tail (rcppml_hip_scale_order + rcppml_hip_gram_loss_mse: 8 launches between the solves of one
# iteration instead of 16) against the separate kernels (bench.py --no-fused-tail), alternating on one box.
# Output: gpurun_out/tail_ab.txt
mkdir -p gpurun_out
out=gpurun_out/tail_ab.txt
: > $out
for rep in 1 2 3; do
  for mode in fused separate; do
    flag=""; [ $mode = separate ] && flag="--no-fused-tail"
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-plugin-figure $flag > gpurun_out/tail_ab_$mode$rep.json 2> gpurun_out/tail_ab_$mode$rep.err
    python - "$mode$rep" gpurun_out/tail_ab_$mode$rep.json >> $out <<'PY'
import json, sys
r = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], "fused_tail=%s" % r.get("fused_tail"), "ms_per_step %.4f" % r["ms_per_step"], "eager %.4f" % (r.get("eager_ms_per_step") or 0),
      "final_loss %.9g" % r["final_loss"], "phases", r["phases_ms_per_step"])
PY
  done
done
cat $out
