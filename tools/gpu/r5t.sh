#!/bin/bash
# final measurement pass 2 of round 5 on the final tree: attention microbench, PMC passes (GEMM rows incl. gemm7, attention kernels), the GPU suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O/profiles
timeout 900 python tools/bench_attn2.py > $O/attn.log 2> $O/attn.err; rc=$?; echo "attn rc=$rc" >> $O/rc.txt
if [ $rc -eq 0 ] && ! grep -q Traceback $O/attn.log; then grep -v "^/opt/amdgpu" $O/attn.log > $O/profiles/r05_attention_microbench.txt; echo "keep: r05_attention_microbench.txt" >> $O/rc.txt; fi
ROUND=r05 bash tools/gpu/final.sh pmc pytest
tail -30 $O/rc.txt; tail -3 $O/pytest.log
