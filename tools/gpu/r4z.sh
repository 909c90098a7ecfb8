#!/bin/bash
# MX fused output: tests + the cfg5 leg with and without it (same box)
O=gpurun_out/r4z; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "mx or layernorm" > $O/pytest_k.log 2>&1; echo "pytest kernels rc=$?"; tail -3 $O/pytest_k.log
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_widths_gpu.py -m gpu -q -x -k "fp8" > $O/pytest_m.log 2>&1; echo "pytest model rc=$?"; grep -E "passed|failed" $O/pytest_m.log | tail -3
C="--model 9b --fp8 --steps 10 --warmup 4 --no-cpu-baseline --no-parity --no-packed-leg --no-cfg5-leg --no-shape-legs"
for f in 1 0 1 0; do
  UNIMP_MX_FUSED_OUT=$f timeout 900 python bench.py $C 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused_out=$f', d['value'], d['ms_per_step'], d['roofline'].get('mx_gemms',{}).get('frac'), d['loss'] if 'loss' in d else d['config'].get('loss'))" | tee -a $O/ab.txt
done
