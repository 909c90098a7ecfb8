#!/bin/bash
O=gpurun_out/r5m; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "attention_fwd_bwd" > $O/pytest_attn.log 2>&1; echo "pytest_attn rc=$?" >> $O/rc.txt
tail -4 $O/pytest_attn.log
timeout 900 python tools/bench_attn3_parts.py > $O/parts.log 2>&1; echo "parts rc=$?" >> $O/rc.txt
cat $O/parts.log
timeout 600 python tools/bench_attn2.py lm64 lm2k > $O/attn_ab.log 2>&1; echo "attn_ab rc=$?" >> $O/rc.txt
cat $O/attn_ab.log
