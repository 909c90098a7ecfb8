#!/bin/bash
# the last commit's tree: the default bench line, the kernel trace of the step, the GPU suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O/profiles
UNIMP_BENCH_SHAPES=1 timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench_default rc=$?" >> $O/rc.txt
grep "^  gemm M=" $O/bench_default.err > $O/profiles/r05_gemm_shapes_b64.txt
ROUND=r05 bash tools/gpu/final.sh prof pytest
tail -12 $O/rc.txt; tail -3 $O/pytest.log
