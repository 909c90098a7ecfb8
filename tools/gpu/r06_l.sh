#!/bin/bash
# round 6: gemm9 (dw / dwpk) at the reference's shape (b = 3 x GA 2 fused: 3072 text rows, 12 336 ViT rows)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_l; mkdir -p $O
AB_BATCH=6 timeout 1500 python tools/bench_gemm_ab.py 5 table,pp256a,pp128a,dw,dwpk > $O/gemm_ab_b6.log 2> $O/gemm_ab_b6.err; echo "gemm_ab rc=$?" >> $O/rc.txt
grep -v amdgpu $O/gemm_ab_b6.log | grep "\[" | cut -c1-230; tail -2 $O/gemm_ab_b6.err; cat $O/rc.txt
