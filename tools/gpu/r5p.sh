#!/bin/bash
# attention3 as the default: attention tests (generation 2 forced onto it), batch sweep A/B kept as a profile, a b = 64 bench line
O=gpurun_out/r5p; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "attention" > $O/pytest_attn.log 2>&1; echo "pytest_attn rc=$?" >> $O/rc.txt
tail -3 $O/pytest_attn.log
timeout 900 python tools/bench_attn2.py lm3 lm6 lm8 lm16 lm32 lm64 lm1k lm2k > $O/attn_ab.log 2>&1; echo "attn_ab rc=$?" >> $O/rc.txt
cat $O/attn_ab.log | cut -c1-200
timeout 900 python tools/bench_attn3_parts.py > $O/parts.log 2>&1; echo "parts rc=$?" >> $O/rc.txt
cat $O/parts.log
timeout 1500 python bench.py --steps 6 --warmup 3 > $O/bench.log 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
tail -c 3000 $O/bench.log
