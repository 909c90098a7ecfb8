# round 5, fourth GPU pass: w4x for the dW form + deeper epilogues, pp256b = asm L phase + L2 prefetch -- bits, A/B; retune; bench (table completed live)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5d; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gemm" > $O/pytest_gemm.log 2>&1; echo "pytest_gemm rc=$?" >> $O/rc.txt
tail -4 $O/pytest_gemm.log
timeout 1200 python tools/bench_gemm_ab.py 3 pp256a,pp256b,w4x,w4x_pf > $O/gemm_ab.log 2>&1; echo "gemm_ab rc=$?" >> $O/rc.txt
cat $O/gemm_ab.log
ROUND=r05 bash tools/gpu/final.sh tune > $O/tune.log 2>&1; echo "tune rc=$?" >> $O/rc.txt
tail -4 $O/tune.log
( time UNIMP_GEMM_TUNE_WRITE=1 UNIMP_BENCH_SHAPES=1 timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; echo "bench rc=$?" >> $O/rc.txt
cp profiles/gemm_autotune_gfx950.json $O/gemm_autotune_gfx950.json
grep "^  gemm M=" $O/bench.err > $O/gemm_shapes.txt
cat $O/bench.time; cat $O/rc.txt
