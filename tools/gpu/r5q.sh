#!/bin/bash
O=gpurun_out/r5q; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "attention_fwd_bwd" > $O/pytest_attn.log 2>&1; echo "pytest_attn rc=$?" >> $O/rc.txt
tail -3 $O/pytest_attn.log
timeout 900 python tools/bench_attn3_parts.py > $O/parts.log 2>&1; echo "parts rc=$?" >> $O/rc.txt
cat $O/parts.log
timeout 900 python tools/bench_attn2.py lm8 lm16 lm64 lm1k lm2k > $O/attn_ab.log 2>&1; echo "attn_ab rc=$?" >> $O/rc.txt
cat $O/attn_ab.log | cut -c1-175
