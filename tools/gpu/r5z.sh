#!/bin/bash
# attention3 with its own lean epilogue (rotation form fixed at compile time): tests, kernel parts, the step with / without it on one box
O=gpurun_out/r5z; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "attention" > $O/pytest_attn.log 2>&1; echo "pytest_attn rc=$?" >> $O/rc.txt
tail -3 $O/pytest_attn.log
timeout 900 python tools/bench_attn3_parts.py > $O/parts.log 2>&1; echo "parts rc=$?" >> $O/rc.txt
cat $O/parts.log
FAST="--no-cpu-baseline --no-parity --no-packed-leg --no-cfg5-leg --no-shape-legs --steps 8 --warmup 3"
for rep in 1 2; do
  UNIMP_DKV3=0 timeout 900 python bench.py $FAST > $O/bench_off_$rep.log 2> $O/bench_off_$rep.err; echo "bench_off_$rep rc=$?" >> $O/rc.txt
  UNIMP_DKV3=1 timeout 900 python bench.py $FAST > $O/bench_on_$rep.log 2> $O/bench_on_$rep.err; echo "bench_on_$rep rc=$?" >> $O/rc.txt
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5z/bench_o*.log")):
    l = [x for x in open(f) if x.startswith("{")]
    if l:
        d = json.loads(l[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
