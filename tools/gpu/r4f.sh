# round 4: the step with W^T frozen weights + one-set / fixed-kind kernels: re-tune the default bench's shapes from scratch, then the default bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4f; mkdir -p $O
T=$PWD/$O/gemm_autotune_gfx950.json; rm -f $T
UNIMP_GEMM_TUNE_FILE=$T UNIMP_GEMM_TUNE_WRITE=1 timeout 900 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > $O/tune.json 2> $O/tune.err; echo "tune rc=$?" > $O/rc.txt
UNIMP_GEMM_TUNE_FILE=$T UNIMP_BENCH_SHAPES=1 timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -3 $O/tune.err; cat $O/bench_default.json; grep "^  gemm M=" $O/bench_default.err | head -30
