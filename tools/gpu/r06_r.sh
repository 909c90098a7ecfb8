#!/bin/bash
# one-launch decode attention: kernel tests, decode model tests, decode step timing
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${R06_TAG:-r06_r}; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py -q -x -k "decode" > $O/pytest.log 2>&1; echo "pytest kernels rc=$?" > $O/rc.txt
tail -15 $O/pytest.log
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -q -x -k "decode or generate or cached or beam" > $O/pytest_model.log 2>&1; echo "pytest model rc=$?" >> $O/rc.txt
tail -3 $O/pytest_model.log
timeout 600 python tools/prof_decode.py 1 200 2>&1 | grep "decode K" | tee $O/decode.txt
timeout 600 python tools/prof_decode.py 10 50 2>&1 | grep "decode K" | tee -a $O/decode.txt
UNIMP_DECODE_STEP_ATTN=0 timeout 600 python tools/prof_decode.py 1 200 2>&1 | grep "decode K" | tee -a $O/decode.txt
