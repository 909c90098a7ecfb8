#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_f; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "skinny" > $O/pytest_kernels.log 2>&1; echo "pytest kernels rc=$?" >> $O/rc.txt; tail -3 $O/pytest_kernels.log
for m in 1 10; do
  UNIMP_SKINNY_ROWS8=0 timeout 300 python tools/bench_skinny.py $m 2>&1 | grep -v amdgpu > $O/skinny_r16_m$m.txt
  UNIMP_SKINNY_ROWS8=1 timeout 300 python tools/bench_skinny.py $m 2>&1 | grep -v amdgpu | cut -c30- > $O/skinny_r8_m$m.txt
  echo "# M=$m: skinny2, 16 rows per workgroup | 8 rows per workgroup where K > 4096 and N <= 3200"; paste -d'|' $O/skinny_r16_m$m.txt $O/skinny_r8_m$m.txt
done
timeout 300 python tools/prof_decode.py 1 200 2>&1 | grep "decode K="
timeout 300 python tools/prof_decode.py 10 50 2>&1 | grep "decode K="
timeout 300 python tools/prof_decode.py 5 100 2>&1 | grep "decode K="
cat $O/rc.txt
