#!/bin/bash
# where does the device idle at the reference's shape (b = 3, GA 2)?
O=gpurun_out/r4v; mkdir -p $O
C="--steps 12 --warmup 4 --batch 3 --grad-accum 2 --no-cpu-baseline --no-roofline --no-parity --no-packed-leg --no-cfg5-leg --no-shape-legs"
timeout 600 python bench.py $C 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no profiler:', d['value'], d['ms_per_step'])" | tee $O/plain.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 900 rocprofv3 --kernel-trace -d $O/st -o st --output-format csv -- python3 bench.py $C > $O/prof.log 2>&1
T="$(find $O/st -name '*kernel_trace.csv' | head -1)"
python tools/trace_window.py "$T" 12 $O/steps.csv gaps | tee $O/gaps.txt | cut -c1-220
rm -rf $O/st
