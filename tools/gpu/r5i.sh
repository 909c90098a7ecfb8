#!/bin/bash
O=gpurun_out/r5i; mkdir -p $O
timeout 600 python tools/debug_attn3.py > $O/debug.log 2>&1; echo "debug rc=$?" >> $O/rc.txt
grep -v "^ \|per d\|worst" $O/debug.log | head -60
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "attention_fwd_bwd" > $O/pytest_attn.log 2>&1; echo "pytest_attn rc=$?" >> $O/rc.txt
tail -8 $O/pytest_attn.log
timeout 600 python tools/bench_attn2.py lm lm64 lm2k > $O/attn_ab.log 2>&1; echo "attn_ab rc=$?" >> $O/rc.txt
cat $O/attn_ab.log
timeout 300 python tools/stamp_attn3.py > $O/stamps.log 2>&1; echo "stamps rc=$?" >> $O/rc.txt
cat $O/stamps.log
