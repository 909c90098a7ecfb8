#!/bin/bash
# rows-per-workgroup decode GEMM: kernel tests, microbench with (default) and without (UNIMP_SKINNY2_ROWS=16), decode step timing
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${R06_TAG:-r06_s}; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py -q -x -k "skinny" > $O/pytest.log 2>&1; echo "pytest kernels rc=$?" > $O/rc.txt
tail -3 $O/pytest.log
for rows in 0 16; do
  echo "UNIMP_SKINNY2_ROWS=$rows"
  UNIMP_SKINNY2_ROWS=$rows SKINNY_LN=1 timeout 300 python tools/bench_skinny.py 1 2>&1 | grep -v amdgpu | tee $O/skinny_m1_rows_$rows.txt
  UNIMP_SKINNY2_ROWS=$rows SKINNY_LN=1 timeout 300 python tools/bench_skinny.py 10 2>&1 | grep -v amdgpu | tee $O/skinny_m10_rows_$rows.txt
  UNIMP_SKINNY2_ROWS=$rows timeout 600 python tools/prof_decode.py 1 200 2>&1 | grep "decode K" | tee -a $O/decode.txt
  UNIMP_SKINNY2_ROWS=$rows timeout 600 python tools/prof_decode.py 10 50 2>&1 | grep "decode K" | tee -a $O/decode.txt
done
