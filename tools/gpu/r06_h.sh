#!/bin/bash
# round 6: same-box alternating step A/B of the round-tail split (UNIMP_GEMM_ROUND_TAIL) at b = 64
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_h; mkdir -p $O
for i in 1 2 3; do
  for v in 0 1; do
    UNIMP_GEMM_ROUND_TAIL=$v timeout 600 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-shape-legs --no-cfg5-leg --no-packed-leg > $O/b_${v}_${i}.json 2> $O/b_${v}_$i.err
    python - <<PY
import json
d = json.load(open("$O/b_${v}_${i}.json"))
print("round_tail=$v run $i:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["gemm_ms_per_step"], d["config"]["gemm_autotune"])
PY
  done
done
