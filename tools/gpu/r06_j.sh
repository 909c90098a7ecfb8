#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_j; mkdir -p $O
UNIMP_DECODE_MERGE_FUSED=0 timeout 300 python tools/prof_decode.py 10 50 2>&1 | grep "decode K="
timeout 600 rocprofv3 --kernel-trace -d $O/trace_k10 -o t --output-format csv -- python3 tools/prof_decode.py 10 48 > $O/prof_k10.log 2>&1
f=$(find $O/trace_k10 -name "*kernel_trace.csv" | head -1)
grep "decode K=" $O/prof_k10.log
python tools/trace_window.py $f 48 $O/decode_k10.csv gaps > $O/decode_k10.txt 2>&1
head -8 $O/decode_k10.txt | cut -c1-160; head -12 $O/decode_k10.csv | cut -c1-150
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
