#!/bin/bash
# round 6: gemm9 (variant dw: 128 x 256 tiles, two workgroups per CU) against the table's kernels -- bits and TFLOP/s per step shape
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_k; mkdir -p $O
timeout 1500 python tools/bench_gemm_ab.py 5 pp256a,dw,dwpk > $O/gemm_ab.log 2> $O/gemm_ab.err; echo "gemm_ab rc=$?" >> $O/rc.txt
grep -v amdgpu $O/gemm_ab.log | cut -c1-220; tail -3 $O/gemm_ab.err; cat $O/rc.txt
