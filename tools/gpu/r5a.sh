# round 5, first GPU pass: gemm7 (w4x) correctness + A/B against the ping-pong family, the tests touched this round, the default bench
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5a; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "w4x or gemm_layouts" > $O/pytest_g7.log 2>&1; echo "pytest_g7 rc=$?" >> $O/rc.txt
tail -15 $O/pytest_g7.log
timeout 900 python tools/bench_gemm_ab.py 5 pp256a,pp256x,w4x,w4x_nt,w4x_sc1,w4 > $O/gemm_ab.log 2>&1; echo "gemm_ab rc=$?" >> $O/rc.txt
cat $O/gemm_ab.log
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_dp_gpu.py tests/test_preprocess_gpu.py -m gpu -q -x -k "fused or flush or eight or two_rank or sharded or exp" > $O/pytest_a.log 2>&1; echo "pytest_a rc=$?" >> $O/rc.txt
timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -k "full_depth" -s > $O/pytest_b.log 2>&1; echo "pytest_b rc=$?" >> $O/rc.txt
( time timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; echo "bench rc=$?" >> $O/rc.txt
tail -5 $O/pytest_a.log; tail -5 $O/pytest_b.log; cat $O/bench.time; cat $O/rc.txt
