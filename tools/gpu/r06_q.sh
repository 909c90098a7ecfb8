#!/bin/bash
# fused-LayerNorm decode GEMM (cooperative form): kernel tests, microbench, decode step timing
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${R06_TAG:-r06_q}; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py -q -x -k "skinny or decode" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
tail -3 $O/pytest.log
SKINNY_LN=1 timeout 300 python tools/bench_skinny.py 1 2>&1 | grep -v amdgpu | tee $O/skinny_m1.txt
SKINNY_LN=1 timeout 300 python tools/bench_skinny.py 10 2>&1 | grep -v amdgpu | tee $O/skinny_m10.txt
timeout 600 python tools/prof_decode.py 1 200 2>&1 | grep "decode K" | tee $O/decode.txt
timeout 600 python tools/prof_decode.py 10 50 2>&1 | grep "decode K" | tee -a $O/decode.txt
