#!/bin/bash
# is the clean-built library as fast as the one final.sh measured?  GEMM A/B table (compare gpurun_out/r4q/ab.log) + the default bench line
O=gpurun_out/r4u; mkdir -p $O
timeout 900 python tools/bench_gemm_ab.py 5 pp256a,pp256 > $O/ab.log 2>&1
grep -v amdgpu $O/ab.log | cut -c1-150 | head -14
timeout 900 python bench.py --no-cpu-baseline --no-parity --no-packed-leg --no-cfg5-leg --no-shape-legs > $O/bench.json 2>$O/bench.err
python -c "import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
rocm-smi --showpower --showclocks 2>/dev/null | head -30 > $O/smi.txt
