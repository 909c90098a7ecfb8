#!/bin/bash
O=gpurun_out/r5o; mkdir -p $O
timeout 900 python tools/bench_attn2.py lm3 lm6 lm8 lm16 lm32 lm64 lm1k lm2k > $O/attn_ab.log 2>&1; echo "attn_ab rc=$?" >> $O/rc.txt
cat $O/attn_ab.log
