#!/bin/bash
O=gpurun_out/r5j; mkdir -p $O
timeout 900 python tools/bench_attn3_parts.py > $O/parts.log 2>&1; echo "parts rc=$?" >> $O/rc.txt
cat $O/parts.log
