#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_o; mkdir -p $O
timeout 900 python tools/bench_gemm_ab.py 5 pp256a,dwpk vit > $O/ab_vit.log 2>&1
AB_BATCH=64 timeout 900 python tools/bench_gemm_ab.py 5 pp256a,dwpk "lm " > $O/ab_lm.log 2>&1
grep "\[" $O/ab_vit.log $O/ab_lm.log | cut -c1-200
