# round 5, third GPU pass: peeled loops as the default + w4x (+ its L2-prefetch build) -- bits, A/B; a fresh autotune table; the default bench with it
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c; mkdir -p $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gemm" > $O/pytest_gemm.log 2>&1; echo "pytest_gemm rc=$?" >> $O/rc.txt
tail -4 $O/pytest_gemm.log
timeout 1200 python tools/bench_gemm_ab.py 3 pp256a,pp256b,pp256x,w4x,w4x_pf > $O/gemm_ab.log 2>&1; echo "gemm_ab rc=$?" >> $O/rc.txt
cat $O/gemm_ab.log
ROUND=r05 bash tools/gpu/final.sh tune > $O/tune.log 2>&1; echo "tune rc=$?" >> $O/rc.txt
tail -15 $O/tune.log
cp gpurun_out/final/gemm_autotune_gfx950.json $O/gemm_autotune_gfx950.json
( time UNIMP_BENCH_SHAPES=1 timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; echo "bench rc=$?" >> $O/rc.txt
grep "^  gemm M=" $O/bench.err > $O/gemm_shapes.txt
timeout 600 python -m pytest tests/test_dp_gpu.py -m gpu -q -x -k "sharded_optimizer_state_eight" > $O/pytest_a.log 2>&1; echo "pytest_a rc=$?" >> $O/rc.txt
tail -3 $O/pytest_a.log; cat $O/bench.time; cat $O/rc.txt
