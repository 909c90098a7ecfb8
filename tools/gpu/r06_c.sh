#!/bin/bash
# round 6, decode pass 2: skinny2 rewritten (all loads upfront, LayerNorm from the waves' own fragments), MLP decode path, hoisted media compare
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "skinny or decode_rope or mxfp8 or mx_" > $O/pytest_kernels.log 2>&1; echo "pytest kernels rc=$?" >> $O/rc.txt; tail -4 $O/pytest_kernels.log
for m in 1 10; do
  UNIMP_SKINNY2=0 timeout 300 python tools/bench_skinny.py $m > $O/skinny_old_m$m.txt 2>&1; echo "skinny old m$m rc=$?" >> $O/rc.txt
  UNIMP_SKINNY2=1 timeout 300 python tools/bench_skinny.py $m > $O/skinny_new_m$m.txt 2>&1; echo "skinny new m$m rc=$?" >> $O/rc.txt
  paste -d'|' $O/skinny_old_m$m.txt $O/skinny_new_m$m.txt | grep -v amdgpu
done
for k in 1 10; do
  timeout 600 rocprofv3 --kernel-trace -d $O/trace_new_k$k -o t --output-format csv -- python3 tools/prof_decode.py $k 48 > $O/prof_new_k$k.log 2>&1; echo "prof new k$k rc=$?" >> $O/rc.txt
  f=$(find $O/trace_new_k$k -name "*kernel_trace.csv" | head -1)
  grep "decode K=" $O/prof_new_k$k.log
  [ -n "$f" ] && python tools/trace_window.py $f 48 $O/decode_new_k$k.csv gaps > $O/decode_new_k$k.txt 2>&1
  head -8 $O/decode_new_k$k.txt; head -24 $O/decode_new_k$k.csv | cut -c1-150
done
timeout 300 python tools/prof_decode.py 1 200 > $O/decode_noprof_k1.log 2>&1; grep "decode K=" $O/decode_noprof_k1.log
timeout 300 python tools/prof_decode.py 10 50 > $O/decode_noprof_k10.log 2>&1; grep "decode K=" $O/decode_noprof_k10.log
timeout 1200 python -m pytest tests/test_fullsize_gpu.py tests/test_model_gpu.py -q -x -m gpu -k "cache or generate or beam or decode or mpt or opt" > $O/pytest_decode.log 2>&1; echo "pytest decode rc=$?" >> $O/rc.txt; tail -4 $O/pytest_decode.log
rocprofv3 -L > $O/counters.txt 2>&1; grep -i -c "" $O/counters.txt; grep -i -o "TCC_EA0_[A-Z_]*DRAM[A-Z_]*\|TCC_[A-Z0-9_]*MALL[A-Z_]*\|TCC_EA0_RDREQ[A-Z_0-9]*\|TCC_EA0_WRREQ[A-Z_0-9]*\|TCC_BUBBLE[A-Z_]*\|TCC_HIT[A-Z_]*\|TCC_MISS[A-Z_]*" $O/counters.txt | sort | uniq -c | head -40
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +20M -delete; find $O -name "*agent_info.csv" -delete
cat $O/rc.txt
