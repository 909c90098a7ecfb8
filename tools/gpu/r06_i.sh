#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_i; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "decode or kv_reorder or skinny" > $O/pytest_kernels.log 2>&1; echo "pytest kernels rc=$?" >> $O/rc.txt; tail -3 $O/pytest_kernels.log
timeout 300 python tools/prof_decode.py 1 200 2>&1 | grep "decode K="
UNIMP_DECODE_MERGE_FUSED=0 timeout 300 python tools/prof_decode.py 1 200 2>&1 | grep "decode K="
timeout 300 python tools/prof_decode.py 10 50 2>&1 | grep "decode K="
UNIMP_DECODE_FUSED=0 UNIMP_SKINNY2=0 UNIMP_DECODE_MERGE_FUSED=0 timeout 300 python tools/prof_decode.py 10 50 2>&1 | grep "decode K="
UNIMP_DECODE_FUSED=0 UNIMP_SKINNY2=0 UNIMP_DECODE_MERGE_FUSED=0 timeout 300 python tools/prof_decode.py 1 200 2>&1 | grep "decode K="
timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -x -m gpu -k "cache or generate or beam" > $O/pytest_decode.log 2>&1; echo "pytest decode rc=$?" >> $O/rc.txt; tail -3 $O/pytest_decode.log
cat $O/rc.txt
