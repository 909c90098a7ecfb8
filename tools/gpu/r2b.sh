cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2b; mkdir -p $O
timeout 1200 python -m pytest tests/test_dp_gpu.py tests/test_model_gpu.py tests/test_fullsize_gpu.py -x -q -s -k "rccl or two_rank or scheduler or checkpoint or beam or compact_head or forward_backward_parity or train_steps" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/rc.txt
# autotune table + per-shape GEMM timings on the current tree
export UNIMP_GEMM_TUNE_FILE=$PWD/$O/gemm_autotune_gfx950.json UNIMP_GEMM_TUNE_WRITE=1 UNIMP_BENCH_SHAPES=1
timeout 1500 python bench.py --steps 8 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
unset UNIMP_GEMM_TUNE_WRITE UNIMP_BENCH_SHAPES
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --batch 3 --grad-accum 2 > $O/bench_b3ga2.json 2> $O/bench_b3ga2.err; echo "bench b3 rc=$?" >> $O/rc.txt
timeout 900 rocprofv3 --kernel-trace --stats -d $O/stats -o st --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $O/prof.log 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +30M -delete
