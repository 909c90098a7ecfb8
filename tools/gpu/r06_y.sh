#!/bin/bash
# two-tile fused-LayerNorm decode GEMM: kernel tests, microbench with (default) and without (UNIMP_SKINNY2_NT=1), decode step timing
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${R06_TAG:-r06_y}; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py -q -x -k "skinny" > $O/pytest.log 2>&1; echo "pytest kernels rc=$?" > $O/rc.txt
tail -3 $O/pytest.log
for nt in 2 1; do
  echo "UNIMP_SKINNY2_NT=$nt"
  UNIMP_SKINNY2_NT=$nt SKINNY_LN=1 timeout 300 python tools/bench_skinny.py 1 2>&1 | grep "LN" | tee $O/skinny_m1_nt_$nt.txt
  UNIMP_SKINNY2_NT=$nt SKINNY_LN=1 timeout 300 python tools/bench_skinny.py 10 2>&1 | grep "LN" | tee $O/skinny_m10_nt_$nt.txt
done
for nt in 2 1 2 1; do
  UNIMP_SKINNY2_NT=$nt timeout 600 python tools/prof_decode.py 1 200 2>&1 | grep "decode K" | sed "s/^/NT=$nt /" | tee -a $O/decode.txt
  UNIMP_SKINNY2_NT=$nt timeout 600 python tools/prof_decode.py 10 50 2>&1 | grep "decode K" | sed "s/^/NT=$nt /" | tee -a $O/decode.txt
done
