#!/bin/bash
# round 6, decode pass 3: skinny2 with the half-line operand map, default-policy vs nontemporal weight loads, long-K form
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_d; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "skinny" > $O/pytest_kernels.log 2>&1; echo "pytest kernels rc=$?" >> $O/rc.txt; tail -3 $O/pytest_kernels.log
for m in 1 10; do
  UNIMP_SKINNY2=0 timeout 300 python tools/bench_skinny.py $m 2>&1 | grep -v amdgpu > $O/skinny_old_m$m.txt
  UNIMP_SKINNY2=1 UNIMP_SKINNY_NT=0 timeout 300 python tools/bench_skinny.py $m 2>&1 | grep -v amdgpu | cut -c30- > $O/skinny_new_m$m.txt
  UNIMP_SKINNY2=1 UNIMP_SKINNY_NT=1 timeout 300 python tools/bench_skinny.py $m 2>&1 | grep -v amdgpu | cut -c30- > $O/skinny_newnt_m$m.txt
  echo "# M=$m: round-3 kernel | skinny2 default-policy loads | skinny2 nontemporal loads"; paste -d'|' $O/skinny_old_m$m.txt $O/skinny_new_m$m.txt $O/skinny_newnt_m$m.txt
done
timeout 300 python tools/prof_decode.py 1 200 2>&1 | grep "decode K="
UNIMP_SKINNY_NT=1 timeout 300 python tools/prof_decode.py 1 200 2>&1 | grep "decode K="
timeout 300 python tools/prof_decode.py 10 50 2>&1 | grep "decode K="
cat $O/rc.txt
