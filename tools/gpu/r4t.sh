#!/bin/bash
# deferred labeled-row count: tests + A/B at the reference's shape (b = 3, GA 2) and at b = 64, same box
O=gpurun_out/r4t; mkdir -p $O
timeout 1500 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "labeled_rows or accumulation or compact or train_steps or parity or scheduler or graphed" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
COMMON="--no-cpu-baseline --no-roofline --no-parity --no-packed-leg --no-cfg5-leg --no-shape-legs"
for i in 1 2; do
  for s in 1 0; do
    UNIMP_ROWS_SYNC=$s timeout 600 python bench.py --batch 3 --grad-accum 2 --steps 30 --warmup 8 $COMMON 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b3ga2 ROWS_SYNC=$s', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt
  done
done
for s in 1 0; do
  UNIMP_ROWS_SYNC=$s timeout 900 python bench.py --steps 10 --warmup 4 $COMMON 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b64 ROWS_SYNC=$s', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt
done
