#!/bin/bash
O=gpurun_out/r5l; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "attention_fwd_bwd" > $O/pytest_attn.log 2>&1; echo "pytest_attn rc=$?" >> $O/rc.txt
tail -4 $O/pytest_attn.log
UNIMP_A3_NW=4 timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "attention_fwd_bwd and gen2" > $O/pytest_attn_nw4.log 2>&1; echo "pytest_attn_nw4 rc=$?" >> $O/rc.txt
tail -2 $O/pytest_attn_nw4.log
timeout 900 python tools/bench_attn3_parts.py > $O/parts.log 2>&1; echo "parts rc=$?" >> $O/rc.txt
cat $O/parts.log
UNIMP_A3_NW=4 timeout 900 python tools/bench_attn3_parts.py > $O/parts_nw4.log 2>&1; echo "parts_nw4 rc=$?" >> $O/rc.txt
cat $O/parts_nw4.log
timeout 600 python tools/bench_attn2.py lm64 lm2k > $O/attn_ab.log 2>&1; echo "attn_ab rc=$?" >> $O/rc.txt
cat $O/attn_ab.log
