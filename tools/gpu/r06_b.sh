#!/bin/bash
# round 6, decode pass: the new decode-step kernels (skinny2 + fused LayerNorm, rope + append) -- tests, the skinny microbench old vs new, the decode
# step's kernel trace old vs new, the peeled MX kernel's tests and microbench
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_b; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "skinny or decode_rope or mxfp8 or mx_" > $O/pytest_kernels.log 2>&1; echo "pytest kernels rc=$?" >> $O/rc.txt; tail -4 $O/pytest_kernels.log
for m in 1 10; do
  UNIMP_SKINNY2=0 timeout 300 python tools/bench_skinny.py $m > $O/skinny_old_m$m.txt 2>&1; echo "skinny old m$m rc=$?" >> $O/rc.txt
  UNIMP_SKINNY2=1 timeout 300 python tools/bench_skinny.py $m > $O/skinny_new_m$m.txt 2>&1; echo "skinny new m$m rc=$?" >> $O/rc.txt
  paste -d'|' $O/skinny_old_m$m.txt $O/skinny_new_m$m.txt | grep -v amdgpu
done
for k in 1 10; do
  UNIMP_SKINNY2=0 UNIMP_DECODE_FUSED=0 timeout 600 rocprofv3 --kernel-trace -d $O/trace_old_k$k -o t --output-format csv -- python3 tools/prof_decode.py $k 48 > $O/prof_old_k$k.log 2>&1; echo "prof old k$k rc=$?" >> $O/rc.txt
  timeout 600 rocprofv3 --kernel-trace -d $O/trace_new_k$k -o t --output-format csv -- python3 tools/prof_decode.py $k 48 > $O/prof_new_k$k.log 2>&1; echo "prof new k$k rc=$?" >> $O/rc.txt
  for v in old new; do
    f=$(find $O/trace_${v}_k$k -name "*kernel_trace.csv" | head -1)
    grep "decode K=" $O/prof_${v}_k$k.log
    [ -n "$f" ] && python tools/trace_window.py $f 48 $O/decode_${v}_k$k.csv gaps > $O/decode_${v}_k$k.txt 2>&1
    head -12 $O/decode_${v}_k$k.txt
  done
done
timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -x -m gpu -k "cache or generate or beam" > $O/pytest_decode.log 2>&1; echo "pytest decode rc=$?" >> $O/rc.txt; tail -4 $O/pytest_decode.log
timeout 600 python tools/bench_mx.py > $O/mx.txt 2>&1; echo "mx rc=$?" >> $O/rc.txt; grep -v amdgpu $O/mx.txt
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +20M -delete; find $O -name "*agent_info.csv" -delete
cat $O/rc.txt
