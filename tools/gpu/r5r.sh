#!/bin/bash
# attention3 (version without the mid-tile barrier): tests incl. the many-items case, and the step with / without it on ONE box
O=gpurun_out/r5r; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "attention" > $O/pytest_attn.log 2>&1; echo "pytest_attn rc=$?" >> $O/rc.txt
tail -5 $O/pytest_attn.log
FAST="--no-cpu-baseline --no-parity --no-packed-leg --no-cfg5-leg --no-shape-legs --steps 8 --warmup 3"
for rep in 1 2; do
  UNIMP_DKV3=0 timeout 900 python bench.py $FAST > $O/bench_off_$rep.log 2> $O/bench_off_$rep.err; echo "bench_off_$rep rc=$?" >> $O/rc.txt
  UNIMP_DKV3=1 timeout 900 python bench.py $FAST > $O/bench_on_$rep.log 2> $O/bench_on_$rep.err; echo "bench_on_$rep rc=$?" >> $O/rc.txt
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5r/bench_o*.log")):
    l = [x for x in open(f) if x.startswith("{")]
    if l:
        d = json.loads(l[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
