#!/bin/bash
# round 5: first runs of the third-generation dK/dV kernel (attention3.hip): the attention tests on every generation, then the A/B
O=gpurun_out/r5h; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention_fwd_bwd" > $O/pytest_attn.log 2>&1; echo "pytest_attn rc=$?" >> $O/rc.txt
timeout 600 python tools/bench_attn2.py lm lm64 lm2k > $O/attn_ab.log 2>&1; echo "attn_ab rc=$?" >> $O/rc.txt
tail -5 $O/pytest_attn.log; cat $O/attn_ab.log
