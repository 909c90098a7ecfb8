#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_e; mkdir -p $O
for k in 1 10; do
  timeout 600 rocprofv3 --kernel-trace -d $O/trace_new_k$k -o t --output-format csv -- python3 tools/prof_decode.py $k 48 > $O/prof_new_k$k.log 2>&1; echo "prof new k$k rc=$?" >> $O/rc.txt
  f=$(find $O/trace_new_k$k -name "*kernel_trace.csv" | head -1)
  grep "decode K=" $O/prof_new_k$k.log
  [ -n "$f" ] && python tools/trace_window.py $f 48 $O/decode_new_k$k.csv gaps > $O/decode_new_k$k.txt 2>&1
  head -8 $O/decode_new_k$k.txt | cut -c1-160; head -22 $O/decode_new_k$k.csv | cut -c1-150
done
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +20M -delete; find $O -name "*agent_info.csv" -delete
